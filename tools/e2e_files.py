"""File-to-file timing of `k4align` on the C2 workload: a 3 Gbp index (.sfx, 15 GB) and 50 M x 100 bp FASTQ reads (10.8 GB) in
tmpfs -> coordinate-sorted SAM in tmpfs; the overlapped pipeline (default) next to the serial whole-input path of round 1
(-Z), and BAM (+ .bai) output at deflate levels 6 and 1.    python tools/e2e_files.py [n_reads=50000000] [out.json]"""
import json
import os
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import kit4b_amd as k4  # noqa: E402

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
out_json = sys.argv[2] if len(sys.argv) > 2 else None
tmp = "/dev/shm/k4e2e"
os.makedirs(tmp, exist_ok=True)
dev = torch.device("cuda:0")
n_chrom, chrom_len, L = 24, 125_000_000, 100
t0 = time.time()
seq = bench.make_genome(dev, n_chrom, chrom_len)
n = seq.numel()
sa = torch.empty(n * 4 + 16, dtype=torch.uint8, device=dev)
k4.build_sa_device(n, 4, seq.data_ptr(), sa.data_ptr())
ix = k4.SfxIndex.from_device(n, 4, seq.data_ptr(), sa.data_ptr(), k4.make_entries(["chr%d" % (i + 1) for i in range(n_chrom)], [chrom_len] * n_chrom),
                             dataset="syn3g", keep=(sa,))
sfx = os.path.join(tmp, "g.sfx")
ix.write_sfx(sfx)
print("index built and written (%.1f GB) in %.1fs" % (os.path.getsize(sfx) / 1e9, time.time() - t0), flush=True)
reads, truth = bench.make_reads(seq, n_chrom, chrom_len, n_reads, L, bench.READS_SEED, dev)
ix.close()
del seq, sa
W = 12 + L + 3 + L + 1
fq = os.path.join(tmp, "r.fq")
with open(fq, "wb") as f:
    step = 5_000_000
    lut = torch.tensor([65, 67, 71, 84, 78], dtype=torch.uint8, device=dev)
    for b in range(0, n_reads, step):
        m = min(step, n_reads - b)
        text = torch.empty((m, W), dtype=torch.uint8, device=dev)
        text[:, 0] = ord("@"); text[:, 1] = ord("r")
        idx = torch.arange(b, b + m, device=dev)
        for d in range(9):
            text[:, 2 + d] = ((idx // (10 ** (8 - d))) % 10 + 48).to(torch.uint8)
        text[:, 11] = 10
        text[:, 12:12 + L] = lut[reads[b:b + m].long()]
        text[:, 12 + L] = 10; text[:, 13 + L] = ord("+"); text[:, 14 + L] = 10
        text[:, 15 + L:15 + 2 * L] = ord("I")
        text[:, W - 1] = 10
        text.cpu().numpy().tofile(f)
del reads, truth
torch.cuda.empty_cache()
print("reads written (%.1f GB) in %.1fs" % (os.path.getsize(fq) / 1e9, time.time() - t0), flush=True)
res = {"workload": "C2: %d x 100 bp FASTQ reads (%.1f GB) vs 3 Gbp .sfx (%.1f GB), files in tmpfs, kalign -s2" % (n_reads, os.path.getsize(fq) / 1e9, os.path.getsize(sfx) / 1e9)}
exe = os.path.join(ROOT, "kit4b_amd", "k4align")
sams = {}
only = os.environ.get("K4_E2E_TAGS", "").split(",") if os.environ.get("K4_E2E_TAGS") else None  # a subset of the runs below
for tag, extra in (("pipelined", []), ("pipelined_t8", ["-t", "8"]), ("pipelined_t16", ["-t", "16"]), ("serial_r01", ["-Z"]), ("bam_z6_t16", ["-t", "16"]), ("bam_z1_t16", ["-t", "16", "-z", "1"]), ("snp_p5", ["-p", "5"]),
                   ("rank1_sam", ["-G", "0", "-t", "16"]), ("bam_rank1_z6_t16", ["-G", "0", "-t", "16"])):  # the -G path with one rank: shard + merge
    if only and tag not in only:
        continue
    sam = os.path.join(tmp, tag + (".bam" if tag.startswith("bam") else ".sam"))
    t0 = time.time()
    p = subprocess.run([exe, "-I", sfx, "-i", fq, "-o", sam, "-s2"] + extra, capture_output=True, text=True,
                       env=dict(os.environ, K4_TRACE="1") if "-G" in extra else None)
    wall = time.time() - t0
    last = [l for l in p.stderr.splitlines() if "alignments written" in l or "alignments reported to" in l or "GPUs written to" in l]
    res[tag] = {"rc": p.returncode, "wall_s": wall, "report": last[-1] if last else p.stderr[-500:], "sam_GB": os.path.getsize(sam) / 1e9 if os.path.exists(sam) else None}
    if "-G" in extra:  # what the rank itself reported, with the stage trace
        res[tag]["rank_stderr"] = [l for l in p.stderr.splitlines() if l.startswith("[k4 trace]") or "rank 0" in l or "index '" in l][-24:]
    import re
    m = re.search(r"index ([\d.]+)s", res[tag]["report"])
    if m:
        res[tag]["Mreads_s_excl_index_load"] = n_reads / (wall - float(m.group(1))) / 1e6
        res[tag]["Mreads_s_wall"] = n_reads / wall / 1e6
    print(tag, json.dumps(res[tag]), flush=True)
    if tag.startswith("snp"):
        snp_line = [l for l in p.stderr.splitlines() if "putative SNPs" in l]
        res[tag]["snp_report"] = snp_line[-1] if snp_line else None
        side = [sam + ext for ext in (".covsegs.wig", ".disnp.csv", ".trisnp.csv")]
        res[tag]["side_files_GB"] = {os.path.basename(f): os.path.getsize(f) / 1e9 for f in side if os.path.exists(f)}
        for f in [sam, sam + ".snp"] + side:
            if os.path.exists(f):
                os.remove(f)
    elif tag.startswith("bam"):
        res[tag]["bai_MB"] = os.path.getsize(sam + ".bai") / 1e6 if os.path.exists(sam + ".bai") else None
        for f in (sam, sam + ".bai"):
            if os.path.exists(f):
                os.remove(f)
    else:
        sams[tag] = sam
if all(os.path.exists(s) for s in sams.values()):
    import hashlib

    def digest(path):
        h = hashlib.sha256()
        with open(path, "rb") as f:
            while True:
                b = f.read(1 << 26)
                if not b:
                    break
                h.update(b)
        return h.hexdigest()

    d = {k: digest(v) for k, v in sams.items()}
    res["sam_identical"] = len(set(d.values())) == 1
    print("sam identical:", res["sam_identical"], flush=True)
for f in list(sams.values()) + [sfx, fq]:
    if os.path.exists(f):
        os.remove(f)
if out_json:
    json.dump(res, open(out_json, "w"), indent=1)
