"""End-to-end stage timings of the device pipeline (text -> reads -> align -> SAM text) on a synthetic C2-shaped input.
    python tools/io_bench.py [n_reads=20000000] [chroms=24] [chrom_mbp=125]"""
import ctypes as C
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
n_chrom = int(sys.argv[2]) if len(sys.argv) > 2 else 24
chrom_len = int(float(sys.argv[3]) * 1e6) if len(sys.argv) > 3 else 125_000_000
L = 100
dev = torch.device("cuda:0")
seq = bench.make_genome(dev, n_chrom, chrom_len)
n = seq.numel()
el = 4 if n < 4_000_000_000 else 5
sa = torch.empty(n * el + 16, dtype=torch.uint8, device=dev)
k4.build_sa_device(n, el, seq.data_ptr(), sa.data_ptr())
names = ["chr%d" % (i + 1) for i in range(n_chrom)]
ix = k4.SfxIndex.from_device(n, el, seq.data_ptr(), sa.data_ptr(), k4.make_entries(names, [chrom_len] * n_chrom), keep=(sa,))
ix.set_max_iter(5000)
reads, truth = bench.make_reads(seq, n_chrom, chrom_len, n_reads, L, 1234, dev)
# FASTQ text, fixed-width records: "@r%09d\n" (12) + L + "\n+\n" (3) + L quals + "\n" (1)
W = 12 + L + 3 + L + 1
text = torch.empty((n_reads, W), dtype=torch.uint8, device=dev)
text[:, 0] = ord("@"); text[:, 1] = ord("r")
idx = torch.arange(n_reads, device=dev)
for d in range(9):
    text[:, 2 + d] = ((idx // (10 ** (8 - d))) % 10 + 48).to(torch.uint8)
text[:, 11] = 10
lut = torch.tensor([65, 67, 71, 84, 78], dtype=torch.uint8, device=dev)
text[:, 12:12 + L] = lut[reads.long()]
text[:, 12 + L] = 10; text[:, 13 + L] = ord("+"); text[:, 14 + L] = 10
text[:, 15 + L:15 + 2 * L] = ord("I")
text[:, W - 1] = 10
text = torch.cat([text.reshape(-1), torch.zeros(16, dtype=torch.uint8, device=dev)])
T = n_reads * W
del reads
torch.cuda.synchronize()
print("text %.2f GB, %d records" % (T / 1e9, n_reads), flush=True)

L_ = k4.lib()
st = torch.cuda.current_stream().cuda_stream
cap = n_reads + 8
d_reads = torch.empty(T + 64, dtype=torch.uint8, device=dev)
d_offs = torch.empty(cap, dtype=torch.int64, device=dev)
d_lens = torch.empty(cap, dtype=torch.int32, device=dev)
d_noff = torch.empty(cap, dtype=torch.int64, device=dev)
d_nlen = torch.empty(cap, dtype=torch.int32, device=dev)
o2 = torch.empty(cap, dtype=torch.int64, device=dev)
l2 = torch.empty(cap, dtype=torch.int32, device=dev)
rr = torch.empty((n_reads, 6), dtype=torch.int32, device=dev)
hits = torch.empty((n_reads, 4), dtype=torch.int32, device=dev)
kp = k4.KalignParams(2, 1, 1, 0, k4.STRAND_BOTH, 1, 0, 0, 0)
ix.reserve(n_reads, L, 1)
host_out = None
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pos, nrec, bases = 0, 0, 0
    while pos < T:
        ln = min(3 << 30, T - pos)
        info = k4.ParseInfo()
        ix._ck(L_.k4_parse_fastx_dev(ix.h, text.data_ptr() + pos, ln, pos, 1 if pos + ln == T else 0, 0, cap - nrec, d_reads.data_ptr(),
                                     bases, d_offs.data_ptr() + 8 * nrec, d_lens.data_ptr() + 4 * nrec, d_noff.data_ptr() + 8 * nrec,
                                     d_nlen.data_ptr() + 4 * nrec, C.byref(info), st))
        nrec += info.n_records; bases += info.n_bases; pos += info.consumed
    torch.cuda.synchronize(); t1 = time.perf_counter()
    u, o, ml = C.c_uint64(), C.c_uint64(), C.c_uint32()
    ix._ck(L_.k4_prepare_reads_dev(ix.h, 0, nrec, 50, 500, d_offs.data_ptr(), d_lens.data_ptr(), None, None, 0, o2.data_ptr(),
                                   l2.data_ptr(), C.byref(u), C.byref(o), C.byref(ml), st))
    torch.cuda.synchronize(); t2 = time.perf_counter()
    ix.kalign_batch_dev(kp, nrec, L, d_reads.data_ptr(), o2.data_ptr(), l2.data_ptr(), rr.data_ptr(), hits.data_ptr(), st)
    torch.cuda.synchronize(); t3 = time.perf_counter()
    nm = k4.SamNames()
    nm.d_text[0] = text.data_ptr(); nm.d_name_off[0] = d_noff.data_ptr(); nm.d_name_len[0] = d_nlen.data_ptr()
    d_sam, nb, stt = C.c_void_p(), C.c_uint64(), k4.SamStats()
    ix._ck(L_.k4_format_sam_dev(ix.h, 0, nrec, rr.data_ptr(), hits.data_ptr(), 1, None, d_reads.data_ptr(), o2.data_ptr(),
                                l2.data_ptr(), C.byref(nm), C.byref(d_sam), C.byref(nb), C.byref(stt), None, st))
    torch.cuda.synchronize(); t4 = time.perf_counter()
    if host_out is None:
        host_out = torch.empty(nb.value, dtype=torch.uint8).pin_memory()
    ix._ck(L_.k4_copy_to_host(ix.h, host_out.data_ptr(), d_sam, nb.value))
    t5 = time.perf_counter()
    L_.k4_free_device(d_sam)
    print("rep %d: records %d | parse %.1f ms (%.0f GB/s text) | filter %.1f ms | align %.1f ms | SAM %.1f ms (%.2f GB, %.0f GB/s, %d lines) | D2H %.1f ms (%.1f GB/s) | device total %.1f ms = %.1f Mreads/s"
          % (rep, nrec, (t1 - t0) * 1e3, T / (t1 - t0) / 1e9, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, nb.value / 1e9,
             nb.value / (t4 - t3) / 1e9, stt.n_lines, (t5 - t4) * 1e3, nb.value / (t5 - t4) / 1e9, (t4 - t0) * 1e3,
             nrec / (t4 - t0) / 1e6), flush=True)
first = bytes(host_out[:400].numpy().tobytes()).decode().split("\n")[:2]
print("\n".join(first))
ix.close()
