"""Latency of a batch-of-one call through the host-pointer entry points (what the CSfxArray facade's AlignReads costs)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import kit4b_amd as k4  # noqa: E402
import synth  # noqa: E402

ix = k4.SfxIndex.open(os.path.join(ROOT, "tests", "golden", "g1.sfx"))
ix.set_max_iter(5000)
names, chroms = synth.golden_genome()
reads, _ = synth.make_reads(chroms, 2000, 100, seed=5)
for r in reads[:50]:
    ix.align_reads_batch([r], 2, 33, 33, 8)
for nb in (1, 16, 256):
    t0 = time.perf_counter()
    k = 0
    for i in range(0, 1000 if nb == 1 else 2000, nb):
        ix.align_reads_batch(reads[i:i + nb], 2, 33, 33, 8)
        k += 1
    dt = time.perf_counter() - t0
    print("batch of %d: %.1f us per call, %.1f us per read" % (nb, dt / k * 1e6, dt / (k * nb) * 1e6), flush=True)
ix.close()
