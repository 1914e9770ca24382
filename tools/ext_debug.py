"""debug aid: the fresh-input comparison of tests/test_gpu_ext.py with every difference printed in full"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import kit4b_amd as k4
import synth
from oracle_bindings import Oracle
import test_gpu_ext as T

O = Oracle()
import pathlib
tmp = pathlib.Path("/tmp")
names, chroms, sites, h, ix = T._fresh(O, tmp, 1234)
np.set_printoptions(linewidth=250)
for rl, kw in ((100, dict(tot_mm=2, core_len=33, core_delta=33, max_slides=8, min_core_len=8)),
               (151, dict(tot_mm=5, core_len=25, core_delta=25, max_slides=12, min_core_len=9, mm_delta=2)),
               (64, dict(tot_mm=3, core_len=16, core_delta=16, max_slides=6, min_core_len=8)),
               (300, dict(tot_mm=6, core_len=42, core_delta=42, max_slides=24, min_core_len=8))):
    reads = T._reads(chroms, sites, rl, 7 * rl)
    for mh, ext in ((1, dict(min_chimeric_len=50)), (5, dict(min_chimeric_len=30, micro_indel_len=20)),
                    (1, dict(micro_indel_len=7, max_splice_junct_len=3500)), (2, dict(min_chimeric_len=65, max_splice_junct_len=600)),
                    (1, dict(strand=2, min_chimeric_len=45, micro_indel_len=12, max_splice_junct_len=2000))):
        a = ix.align_reads_ext_batch(reads, max_hits=mh, **kw, **ext)
        b = O.align_reads_ext_batch(h, reads, max_hits=mh, **kw, **ext)
        bad = np.zeros(len(reads), bool)
        for k in ("rslt", "inst", "low", "nxt", "hits", "seg2"):
            d = a[k] != b[k]
            if d.ndim > 1: d = d.any(axis=1)
            bad |= d
        print("rl", rl, "mh", mh, ext, "diffs", int(bad.sum()), flush=True)
        for i in np.nonzero(bad)[0][:4]:
            print(" read", i, "".join("ACGTN"[x] for x in reads[i]))
            for k in ("rslt", "inst", "low", "nxt"): print("   ", k, a[k][i], b[k][i])
            print("    gpu hits", a["hits"][i], a["seg2"][i]); print("    cpu hits", b["hits"][i], b["seg2"][i])
