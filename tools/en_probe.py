"""Why do `bench.py --workload rep`'s device tallies and the reference's differ in 15 reads with Ns?  The first 8 M reads of that
workload through both routes on the device: as codes (what bench.py times) and as FASTA text through k4_parse_fastx_dev (what
k4align and the reference see); reads whose NAR differs are printed.   python tools/en_probe.py [n_reads=8000000]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import kit4b_amd as k4  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
eng = bench.GpuEngine()
dev = eng.device(0)
n_chrom, chrom_len, L = 8, 125_000_000, 100
seq = bench.make_genome(dev, n_chrom, chrom_len)
bench.implant_repeats(seq, n_chrom, chrom_len, 40_000, dev)
eng.build_index(seq, n_chrom, chrom_len, 0, lambda *a: None)
reads, _ = bench.make_reads(seq, n_chrom, chrom_len, n, L, bench.shard_seed(False, 0) if hasattr(bench, "shard_seed") else bench.READS_SEED, dev)
eng.prepare(reads, n, L, False, 2)
eng.step()
torch.cuda.synchronize()
nar_raw = eng.out[:, 4].cpu().numpy().copy()
tmp = "/tmp/en_probe.fa"
bench.write_fasta(reads, tmp, dev)
text = open(tmp, "rb").read()
p = eng.ix.parse_fastx(text)
assert p["n"] == n, (p["n"], n)
lens = p["lens"].cpu().numpy()
offs = p["offs"].cpu().numpy()
rd = p["reads"].cpu().numpy()
raw = reads.cpu().numpy()
bad_len = np.nonzero(lens != L)[0]
print("reads whose parsed length differs:", len(bad_len), bad_len[:10], lens[bad_len[:10]])
same = 0
diff = []
for i in range(n):
    if lens[i] != L or not np.array_equal(rd[offs[i]:offs[i] + L], np.minimum(raw[i], 4)):
        diff.append(i)
print("reads whose parsed codes differ from the codes bench.py aligns:", len(diff))
for i in diff[:20]:
    print(i, "raw ", "".join("ACGTN567"[c] for c in raw[i]))
    print(i, "text", text[i * (10 + L + 1) + 10:i * (10 + L + 1) + 10 + L].decode())
    print(i, "pars", "".join("ACGTN567"[c] for c in rd[offs[i]:offs[i] + lens[i]]), "nar raw", nar_raw[i])
# the reference on the same FASTA, every read reported (-M1: unaligned reads carry YU:Z:<NAR>)
import subprocess
sfx = "/tmp/en_probe.sfx"
eng.ix.write_sfx(sfx)
ngs = os.path.join(ROOT, "oracle", "_ref", "ngskit4b")
r = subprocess.run([ngs, "kalign", "-I", sfx, "-o", "/tmp/en_probe.sam", "-T", "16", "-F", "/tmp/en_probe.log", "-s2", "-M1", "-i", tmp], capture_output=True)
print("reference rc", r.returncode, flush=True)
ref_en = set()
ref_state = {}
with open("/tmp/en_probe.sam") as f:
    for ln in f:
        if ln[0] == "@":
            continue
        t = ln.split("\t", 2)
        i = int(t[0][1:])
        tag = ln.rstrip().rsplit("\t", 1)[-1]
        ref_state[i] = tag if tag.startswith("YU:Z:") else "AA"
gpu_en = set(np.nonzero(nar_raw == 2)[0].tolist())
ref_en = set(i for i, v in ref_state.items() if v == "YU:Z:EN")
print("EN: device", len(gpu_en), "reference", len(ref_en), "device only", len(gpu_en - ref_en), "reference only", len(ref_en - gpu_en), flush=True)
print("states in the reference's SAM:", {k: sum(1 for v in ref_state.values() if v == k) for k in set(ref_state.values())}, len(ref_state), flush=True)
print("codes above N in the reads:", int((raw > 4).sum()), np.unique(raw[raw > 4])[:10], flush=True)
for i in sorted(gpu_en - ref_en)[:12]:
    print(i, "reference says", ref_state.get(i), "Ns", int((raw[i] == 4).sum()), "".join("ACGTN???"[min(int(c), 7)] for c in raw[i]), flush=True)
for i in sorted(ref_en - gpu_en)[:6]:
    print(i, "device nar", int(nar_raw[i]), "Ns", int((raw[i] == 4).sum()), "".join("ACGTN???"[min(int(c), 7)] for c in raw[i]), flush=True)
for f in (tmp, sfx, "/tmp/en_probe.sam", "/tmp/en_probe.log"):
    if os.path.exists(f):
        os.remove(f)
