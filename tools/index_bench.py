"""`k4index` (and optionally `ngskit4b index`) on a synthetic genome FASTA:  python tools/index_bench.py [chroms=8] [chrom_mbp=125] [ref=0|1]"""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

n_chrom = int(sys.argv[1]) if len(sys.argv) > 1 else 8
chrom_len = int(float(sys.argv[2]) * 1e6) if len(sys.argv) > 2 else 125_000_000
run_ref = len(sys.argv) > 3 and sys.argv[3] == "1"
dev = torch.device("cuda:0")
tmp = tempfile.mkdtemp(prefix="k4idx_")
fa = os.path.join(tmp, "g.fa")
lut = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device=dev)
g = torch.Generator(device=dev)
g.manual_seed(5)
with open(fa, "wb") as f:
    for c in range(n_chrom):
        W = 80
        rows = chrom_len // W
        s = lut[torch.randint(0, 4, (rows, W), device=dev, generator=g)]
        t = torch.cat([s, torch.full((rows, 1), 10, dtype=torch.uint8, device=dev)], dim=1)
        f.write(b">chr%d\n" % (c + 1))
        f.write(t.cpu().numpy().tobytes())
del s, t
torch.cuda.empty_cache()
print("FASTA %.2f GB" % (os.path.getsize(fa) / 1e9), flush=True)
t0 = time.time()
p = subprocess.run([os.path.join(ROOT, "kit4b_amd", "k4index"), "-i", fa, "-o", os.path.join(tmp, "k4.sfx"), "-r", "syn"], capture_output=True, text=True)
print("k4index rc", p.returncode, "wall %.1fs" % (time.time() - t0), p.stderr.strip().splitlines()[-1], flush=True)
if run_ref:
    t0 = time.time()
    r = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "ngskit4b"), "index", "-i", fa, "-o", os.path.join(tmp, "ref.sfx"), "-r", "syn", "-T", "16",
                        "-F", os.path.join(tmp, "log")], capture_output=True, text=True)
    print("ngskit4b index rc", r.returncode, "wall %.1fs" % (time.time() - t0), flush=True)
    a, b = os.path.getsize(os.path.join(tmp, "ref.sfx")), os.path.getsize(os.path.join(tmp, "k4.sfx"))
    print("sizes", a, b, flush=True)
import shutil
shutil.rmtree(tmp, ignore_errors=True)
