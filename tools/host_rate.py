"""PCIe-inclusive rate of the host-pointer entry point k4_kalign_batch (reads in pageable host memory in, records out)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import kit4b_amd as k4  # noqa: E402

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
dev = torch.device("cuda:0")
n_chrom, chrom_len, L = 24, 125_000_000, 100
seq = bench.make_genome(dev, n_chrom, chrom_len)
n = seq.numel()
sa = torch.empty(n * 4 + 16, dtype=torch.uint8, device=dev)
k4.build_sa_device(n, 4, seq.data_ptr(), sa.data_ptr())
ix = k4.SfxIndex.from_device(n, 4, seq.data_ptr(), sa.data_ptr(), k4.make_entries(["chr%d" % (i + 1) for i in range(n_chrom)], [chrom_len] * n_chrom),
                             keep=(sa,))
ix.set_max_iter(5000)
reads, truth = bench.make_reads(seq, n_chrom, chrom_len, n_reads, L, 77, dev)
cat = reads.cpu().numpy().reshape(-1)
offs = np.arange(n_reads, dtype=np.uint64) * L
lens = np.full(n_reads, L, dtype=np.uint32)
for rep in range(3):
    t0 = time.perf_counter()
    r = ix.kalign_batch((cat, offs, lens), max_subs=2)
    dt = time.perf_counter() - t0
    print("rep %d: %d reads in %.3f s = %.1f Mreads/s (host buffers, PCIe both ways, %d accepted)"
          % (rep, n_reads, dt, n_reads / dt / 1e6, int((r["out"]["nar"] == 1).sum())), flush=True)
ix.close()
