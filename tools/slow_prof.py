#!/usr/bin/env python3
"""Where the general kernel (k4k_align_slow) spends its cycles on a repeat-rich genome.

Build the instrumented library first, in the container (it travels with the snapshot):
    make -C kit4b_amd/csrc EXTRA_HIPFLAGS=-DK4_SLOW_PROF OBJDIR=../_build_prof OUT=../libk4sfx_prof.so ../libk4sfx_prof.so
then on the GPU box:
    python tools/slow_prof.py [--reads 20000000] [--chrom-mbp 125] [--repeats 40000]
Prints one JSON line: cycles per section summed over waves (the sections of k4d_lcm_slow, see K4_SLOW_PROF in k4_align.hip),
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
    eng.prepare(reads, a.reads, a.read_len, False, a.max_subs)
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
    names = ["run_search", "walk", "hamming", "replay", "read_total", "read_setup"]
    cyc = dict(zip(names, v[:6]))
    n_runs, n_steps, n_lcm, n_reads, n_members = v[8], v[9], v[10], v[11], v[12]
    tot = max(cyc["read_total"], 1)
    ctr = ix.counters()
    # random touches of the general kernel: every pivot of a run search reads one suffix element and one window, every
    # in-bounds run member probes its wave's dedupe table (load + compare-and-swap) and fetches one window (two 16-byte
    # loads of one or two adjacent lines: counted once).  The members' suffix elements are consecutive: not random.
    slow_ms = dt * 1e3 - fast_ms
    touches = 2 * v[7] + 3 * v[10]
    print(json.dumps({
        "step_kernels_ms": fast_ms, "general_kernel_ms_est": slow_ms,
        "general_kernel_random_touches": touches, "general_kernel_Gtouches_s": touches / max(slow_ms, 1e-9) / 1e6,
        "calibration_Gtouches_s": 49.0,
        "lib": a.lib, "reads": a.reads, "batch_ms": dt * 1e3, "slow_reads": n_reads, "cycles": cyc,
        "share_of_read_total": {k: round(cyc[k] / tot, 3) for k in names if k != "read_total"},
        "per_slow_read": {"cycles": tot / max(n_reads, 1), "lookups": v[6] / max(n_reads, 1), "runs": n_runs / max(n_reads, 1),
                          "walk_steps": n_steps / max(n_reads, 1), "run_members": n_members / max(n_reads, 1)},
        "from_phase_hist": v[13:16], "search_cycles_per_lookup": cyc["run_search"] / max(v[6], 1),
        "per_run": { "members": n_members / max(n_runs, 1), "steps": n_steps / max(n_runs, 1)},
        "per_step_cycles": {"walk": cyc["walk"] / max(n_steps, 1), "hamming": cyc["hamming"] / max(n_steps, 1), "replay": cyc["replay"] / max(n_steps, 1)},
        "counters": {k: int(ctr[k]) for k in ("n_lookup", "n_probe", "n_cand", "n_slow")},
    }))


if __name__ == "__main__":
    main()
