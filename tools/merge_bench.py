#!/usr/bin/env python3
"""Throughput of the parallel shard merge (k4_merge.h, `k4merge` / the parent of `k4align -G`): N coordinate-sorted SAM shards of
a C2-shaped run (100 bp reads, ~145 bytes a record) in tmpfs -> one coordinate-sorted SAM in tmpfs.

    python tools/merge_bench.py [--records 46000000] [--shards 8] [--threads 0] [--dir /dev/shm/k4merge_bench]

Prints one JSON line (bytes, seconds, GB/s per thread count).  Host-only; the shards are synthetic (sorted random loci)."""
import argparse
import json
import os
import shutil
import subprocess
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def digits(col, v, width):
    for d in range(width):
        col[:, d] = (v // 10 ** (width - 1 - d)) % 10 + 48


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--records", type=int, default=46_000_000)
    ap.add_argument("--shards", type=int, default=8)
    ap.add_argument("--threads", default="1,4,8,16")
    ap.add_argument("--dir", default="/dev/shm/k4merge_bench")
    a = ap.parse_args()
    os.makedirs(a.dir, exist_ok=True)
    rng = np.random.default_rng(7)
    n_chrom, chrom_len, L = 24, 125_000_000, 100
    hdr = "@HD\tVN:1.4\tSO:coordinate\n" + "".join("@SQ\tAS:syn3g\tSN:chr%02d\tLN:%d\n" % (c + 1, chrom_len) for c in range(n_chrom)) + "@PG\tID:ngskit4b\tVN:2.0.2\n"
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    paths, total = [], 0
    t0 = time.time()
    for k in range(a.shards):
        n = a.records // a.shards
        g = np.sort(rng.integers(0, n_chrom * (chrom_len - L), n, dtype=np.int64))
        chrom, pos = g // (chrom_len - L), g % (chrom_len - L) + 1
        tmpl = b"r000000000\t00\tchr00\t000000000\t254\t100M\t*\t0\t0\t" + b"A" * L + b"\t*\n"
        rec = np.tile(np.frombuffer(tmpl, dtype=np.uint8), (n, 1))
        digits(rec[:, 1:10], np.arange(n, dtype=np.int64) + k * n, 9)
        strand = rng.integers(0, 2, n)
        rec[:, 11] = 48 + strand            # FLAG 00 / 16
        rec[:, 12] = np.where(strand == 1, 54, 48)
        digits(rec[:, 17:19], chrom + 1, 2)
        digits(rec[:, 20:29], pos, 9)
        s0 = tmpl.index(b"A" * L)
        rec[:, s0:s0 + L] = lut[rng.integers(0, 4, (n, L), dtype=np.uint8)]
        p = os.path.join(a.dir, "shard%d.sam" % k)
        with open(p, "wb") as f:
            f.write(hdr.encode())
            f.write(rec.tobytes())
        total += os.path.getsize(p)
        paths.append(p)
        del rec
    gen_s = time.time() - t0
    out = os.path.join(a.dir, "merged.sam")
    res = {}
    for t in [int(x) for x in a.threads.split(",")]:
        if os.path.exists(out):
            os.remove(out)
        t0 = time.time()
        r = subprocess.run([os.path.join(ROOT, "kit4b_amd", "k4merge"), "-t", str(t), out] + paths, capture_output=True, text=True)
        dt = time.time() - t0
        assert r.returncode == 0, r.stderr
        res["threads_%d" % t] = {"seconds": round(dt, 3), "GB_per_s": round(total / dt / 1e9, 3)}
    # sortedness of the result (keys as the merge sees them), on a sample of the file
    sz = os.path.getsize(out)
    with open(out, "rb") as f:
        f.seek(len(hdr))
        head = f.read(50_000_000).split(b"\n")[:-1]
    keys = [(l.split(b"\t")[2], int(l.split(b"\t")[3])) for l in head]
    print(json.dumps({"shards": a.shards, "records": (a.records // a.shards) * a.shards, "input_bytes": total, "output_bytes": sz,
                      "generate_s": round(gen_s, 1), "cores": len(os.sched_getaffinity(0)), "sorted_sample_ok": keys == sorted(keys), "merge": res}))
    shutil.rmtree(a.dir, ignore_errors=True)


if __name__ == "__main__":
    main()
