#!/usr/bin/env python3
"""Where the general kernel (k4k_align_slow) spends its cycles on a repeat-rich genome.

Build the instrumented library first, in the container (it travels with the snapshot):
    make -C kit4b_amd/csrc EXTRA_HIPFLAGS=-DK4_SLOW_PROF OBJDIR=../_build_prof OUT=../libk4sfx_prof.so ../libk4sfx_prof.so
then on the GPU box:
    python tools/slow_prof.py [--reads 20000000] [--chrom-mbp 125] [--repeats 40000]
Prints one JSON line: cycles per section summed over waves (the sections of k4d_lcm_batched, see K4_SLOW_PROF in k4_general.hip),
their shares, and per-read / per-run averages.  A development tool: not part of the test suite, the bench or the product."""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import kit4b_amd as k4  # noqa: E402

_lib = [x.split("=", 1)[1] for x in sys.argv[1:] if x.startswith("--lib=")]
k4.LIB_PATH = os.path.join(ROOT, "kit4b_amd", _lib[0] if _lib else "libk4sfx_prof.so")

import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default="libk4sfx_prof.so", help="library under kit4b_amd/ (an uninstrumented one gives batch_ms only)")
    ap.add_argument("--reads", type=int, default=20_000_000)
    ap.add_argument("--chroms", type=int, default=8)
    ap.add_argument("--chrom-mbp", type=float, default=125.0)
    ap.add_argument("--repeats", type=int, default=40_000)
    ap.add_argument("--read-len", type=int, default=100)
    ap.add_argument("--max-subs", type=int, default=2)
    ap.add_argument("--ext", default="", help="as bench.py --ext: the EXT instantiation's profile")
    a = ap.parse_args()
    eng = bench.GpuEngine()
    dev = eng.device(0)
    chrom_len = int(a.chrom_mbp * 1e6)
    seq = bench.make_genome(dev, a.chroms, chrom_len)
    if a.repeats:
        bench.implant_repeats(seq, a.chroms, chrom_len, a.repeats, dev)
    eng.build_index(seq, a.chroms, chrom_len, 0, lambda *x: print("[prof]", *x, file=sys.stderr))
    ix = eng.ix
    reads, _ = bench.make_reads(seq, a.chroms, chrom_len, a.reads, a.read_len, 4321, dev)
    ext_kw = {}
    for tok in filter(None, a.ext.split(",")):
        ext_kw[{"c": "min_chimeric_len", "a": "micro_indel_len", "A": "max_splice_junct_len"}[tok[0]]] = int(tok[1:])
    eng.prepare(reads, a.reads, a.read_len, False, a.max_subs, ext_kw)
    L = k4.lib()
    L.k4i_debug_prof.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    buf = (C.c_uint64 * 32)()
    for rep in range(2):
        L.k4i_debug_prof(ix.h, buf)
        eng.timing_begin()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        (fast_ms, _g_ms), _, _ = eng.timing_end()
    L.k4i_debug_prof(ix.h, buf)
    v = [int(x) for x in buf]
    # slots of K4_SLOW_PROF in k4_general.hip (k4d_lcm_batched): cycles 0 lookup (k-mer table, deep-bucket searches, prefix sums),
    # 1 slots (suffix elements + windows: member? distance?), 2 replay, 4 whole reads, 5 read set-up; counts 6 (strand, core)
    # pairs, 7 slots evaluated, 8 groups, 9 steps, 10 in-bounds members, 11 reads, 12 members, 13..15 reads by phases done before
    names = ["lookup", "slots", "replay", "unused", "read_total", "read_setup"]
    cyc = dict(zip(names, v[:6]))
    n_pairs, n_slots, n_groups, n_steps, n_inb, n_reads, n_members = v[6], v[7], v[8], v[9], v[10], v[11], v[12]
    tot = max(cyc["read_total"], 1)
    ctr = ix.counters()
    slow_ms = _g_ms
    print(json.dumps({
        "step_kernels_ms": fast_ms, "general_kernel_ms": slow_ms,
        "lib": a.lib, "reads": a.reads, "batch_ms": dt * 1e3, "slow_reads": n_reads, "cycles": cyc,
        "share_of_read_total": {k: round(cyc[k] / tot, 3) for k in names if k not in ("read_total", "unused")},
        "per_slow_read": {"cycles": tot / max(n_reads, 1), "pairs": n_pairs / max(n_reads, 1), "groups": n_groups / max(n_reads, 1),
                          "steps": n_steps / max(n_reads, 1), "slots": n_slots / max(n_reads, 1), "members": n_members / max(n_reads, 1),
                          "in_bounds_members": n_inb / max(n_reads, 1)},
        "from_phase_hist": v[13:16],
        "per_group_cycles": {"lookup": cyc["lookup"] / max(n_groups, 1)},
        "per_step_cycles": {"slots": cyc["slots"] / max(n_steps, 1), "replay": cyc["replay"] / max(n_steps, 1)},
        "slots_per_s_G": n_slots / max(slow_ms, 1e-9) / 1e6,
        "replay_parts_per_step_cycles": {"open": v[16] / max(n_steps, 1), "filters_insert": v[17] / max(n_steps, 1), "limits_unclean": v[18] / max(n_steps, 1),
                                         "fold": v[19] / max(n_steps, 1)},
        "per_step": {"segments": v[21] / max(n_steps, 1), "fold_iterations": v[22] / max(n_steps, 1), "unclean_candidates": v[20] / max(n_steps, 1)},
        "ext_two_seg": {"calls": v[21], "steps": v[22], "candidates_explored": v[20], "cycles_lookup": v[16], "cycles_core_cmp": v[17],
                        "cycles_explorers": v[18], "cycles_both_calls": v[19]} if a.ext else None,
        "counters": {k: int(ctr[k]) for k in ("n_lookup", "n_probe", "n_cand", "n_slow")},
    }))


if __name__ == "__main__":
    main()
