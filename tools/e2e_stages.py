#!/usr/bin/env python3
"""Where the host-text -> host-SAM pipeline (bench.py "e2e") spends its wall time: one run of k4_pipeline_* over C2-shaped
input with a clock read between the calls.  A development tool (GPU box): python tools/e2e_stages.py [--reads 20000000]"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import kit4b_amd as k4  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=20_000_000)
    ap.add_argument("--chroms", type=int, default=24)
    ap.add_argument("--chrom-mbp", type=float, default=125.0)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--chunk-mb", type=str, default="0", help="comma list of upload chunk sizes (MB; 0 = the default)")
    ap.add_argument("--pe", action="store_true", help="2 x 150 bp pairs (C3-shaped: -U2 -d200 -D600) instead of 100 bp single reads")
    a = ap.parse_args()
    eng = bench.GpuEngine()
    dev = eng.device(0)
    chrom_len = int(a.chrom_mbp * 1e6)
    seq = bench.make_genome(dev, a.chroms, chrom_len)
    eng.build_index(seq, a.chroms, chrom_len, 0, lambda *x: print("[e2e]", *x, file=sys.stderr))
    pe = a.pe
    L_ = 150 if pe else 100
    if pe:
        reads, _ = bench.make_pe_reads(seq, a.chroms, chrom_len, a.reads // 2, L_, 4321, dev)
    else:
        reads, _ = bench.make_reads(seq, a.chroms, chrom_len, a.reads, L_, 4321, dev)
    ends = 2 if pe else 1
    n = a.reads // ends
    W = 12 + L_ + 3 + L_ + 1
    T = n * W
    h_texts = []
    for e in range(ends):
        text = eng.fastq_text(reads[e:ends * n:ends] if pe else reads[:n], 0, n, L_, dev)
        h = torch.empty(T, dtype=torch.uint8, pin_memory=True)
        h.copy_(text.reshape(-1))
        h_texts.append(h)
        del text
    h_text = h_texts[0]
    h_sam = torch.empty(ends * n * (L_ + 90), dtype=torch.uint8, pin_memory=True)
    del reads
    torch.cuda.synchronize()
    lib = k4.lib()
    kp = k4.KalignParams(2, 1, 1, 0, k4.STRAND_BOTH, 10 if pe else 1, 1 if pe else 0, 0, 0)
    runs = []
    for chunk_mb in [int(x) for x in a.chunk_mb.split(',')]:
      for rep in range(a.reps):
          prm = k4.PipelineParams()
          prm.paired = 1 if pe else 0
          prm.kp = kp
          if pe:
              prm.pe = k4.PeParams(2, 200, 600, 0)
              prm.expect_text_bytes[1] = T
          prm.min_len, prm.max_len, prm.chunk_bytes = 50, 500, chunk_mb << 20
          prm.expect_text_bytes[0] = T
          pl = C.c_void_p()
          t = [time.perf_counter()]
          eng.ix._ck(lib.k4_pipeline_open(eng.ix.h, C.byref(prm), C.byref(pl))); t.append(time.perf_counter())
          for e in range(ends):
              eng.ix._ck(lib.k4_pipeline_submit_host(pl, e, h_texts[e].data_ptr(), T, 1))
          t.append(time.perf_counter())
          view = k4.PipelineView()
          eng.ix._ck(lib.k4_pipeline_wait_aligned(pl, C.byref(view))); t.append(time.perf_counter())
          stats, nbytes = k4.SamStats(), C.c_uint64()
          eng.ix._ck(lib.k4_pipeline_format(pl, C.byref(stats), None, C.byref(nbytes))); t.append(time.perf_counter())
          got = C.c_uint64()
          eng.ix._ck(lib.k4_pipeline_read_sam(pl, h_sam.data_ptr(), h_sam.numel(), C.byref(got))); t.append(time.perf_counter())
          lib.k4_pipeline_close(pl); t.append(time.perf_counter())
          names = ["open", "submit_host", "wait_aligned", "format", "read_sam", "close"]
          r = {k: round((t[i + 1] - t[i]) * 1e3, 2) for i, k in enumerate(names)}
          r["total_ms_without_close"] = round((t[5] - t[0]) * 1e3, 2)
          r["sam_GB"] = got.value / 1e9
          r["chunk_mb"] = chunk_mb
          runs.append(r)
    d_buf = torch.empty(T, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); d_buf.copy_(h_text, non_blocking=True); torch.cuda.synchronize(); t_up = time.perf_counter() - t0
    print(json.dumps({"reads": n, "text_GB": T / 1e9, "h2d_alone_ms": round(t_up * 1e3, 2), "runs": runs}))


if __name__ == "__main__":
    main()
