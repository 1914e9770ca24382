"""Full-scale check against the REAL reference on the GPU box's host cores (run through gpurun; needs oracle/_ref/ngskit4b,
which travels with the snapshot):  3 Gbp index built on the GPU -> .sfx file -> `ngskit4b kalign -s2 -T<cores>` and
`k4align` on the same FASTA -> SAM records compared, both timed.
    python tools/ref_fullscale.py [n_reads=2000000] [chroms=24] [chrom_mbp=125] [threads=16] [pe_mode=0] [read_len=100] [repeats=0] [extra="-r5 -R8"]
extra: further options handed to both programs (e.g. the report-all multi-loci mode).
repeats > 0: that many segment copies (high-copy families and pairs, 0-3 % diverged) and N runs are implanted first.
pe_mode 1..4: n_reads pairs of 2 x read_len, `-U<pe_mode> -d200 -D600` on both programs."""
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import kit4b_amd as k4  # noqa: E402

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
n_chrom = int(sys.argv[2]) if len(sys.argv) > 2 else 24
chrom_len = int(float(sys.argv[3]) * 1e6) if len(sys.argv) > 3 else 125_000_000
threads = int(sys.argv[4]) if len(sys.argv) > 4 else 16
pe_mode = int(sys.argv[5]) if len(sys.argv) > 5 else 0
NGS = os.path.join(ROOT, "oracle", "_ref", "ngskit4b")
K4ALIGN = os.path.join(ROOT, "kit4b_amd", "k4align")
assert os.path.exists(NGS), "oracle/_ref/ngskit4b missing (make -C oracle ngskit4b where /root/reference exists)"
L = int(sys.argv[6]) if len(sys.argv) > 6 else 100
tmp = tempfile.mkdtemp(prefix="k4ref_")
print("scratch", tmp, "free GB", shutil.disk_usage(tmp).free / 1e9, flush=True)
dev = torch.device("cuda:0")
seq = bench.make_genome(dev, n_chrom, chrom_len)
n = seq.numel()
n_rep = int(sys.argv[7]) if len(sys.argv) > 7 else 0
extra = sys.argv[8].split() if len(sys.argv) > 8 else []
subs = [] if any(a.startswith("-s") for a in (sys.argv[8].split() if len(sys.argv) > 8 else [])) else ["-s2"]
k4_extra = sys.argv[10].split() if len(sys.argv) > 10 else []  # options for k4align only (e.g. "-b 100")
dropin_threads = int(sys.argv[9]) if len(sys.argv) > 9 else 0  # > 0: also run oracle/_ref/ngskit4b_k4 (the reference's
# own front end on libk4sfx.so through the facade) with this many threads
if n_rep > 0:
    bench.implant_repeats(seq, n_chrom, chrom_len, n_rep, dev)
elif n_rep < 0:  # one element copied -n_rep times + low-complexity stretches: MaxIter and the node limit are reached
    bench.implant_stress(seq, n_chrom, chrom_len, -n_rep, dev)
    print("implanted %d repeat copies" % n_rep, flush=True)
el = 4 if n < 4_000_000_000 else 5
sa = torch.empty(n * el + 16, dtype=torch.uint8, device=dev)
k4.build_sa_device(n, el, seq.data_ptr(), sa.data_ptr())
names = ["chr%d" % (i + 1) for i in range(n_chrom)]
ix = k4.SfxIndex.from_device(n, el, seq.data_ptr(), sa.data_ptr(), k4.make_entries(names, [chrom_len] * n_chrom), dataset="syn3g",
                             keep=(sa,))
sfx = os.path.join(tmp, "g.sfx")
t0 = time.time()
ix.write_sfx(sfx)
print("wrote %s (%.1f GB) in %.1fs" % (sfx, os.path.getsize(sfx) / 1e9, time.time() - t0), flush=True)
if pe_mode:
    reads, truth = bench.make_pe_reads(seq, n_chrom, chrom_len, n_reads, L, bench.READS_SEED + 2, dev)
    fa, fa2 = os.path.join(tmp, "r1.fa"), os.path.join(tmp, "r2.fa")
    bench.write_fasta(reads[0::2], fa, dev)
    bench.write_fasta(reads[1::2], fa2, dev)
    in_args = ["-i", fa, "-u", fa2, "-U%d" % pe_mode, "-d200", "-D600"]
elif os.environ.get("K4_REF_HAP") == "1":  # three haplotypes (the genome, one with a substitution every ~80 bases, one with those and
    # half as many again), a third of the reads from each: called SNP loci close enough for the DiSNP / TriSNP files to fill
    def haplotype(src, seed, every):
        g = torch.Generator(device=dev)
        g.manual_seed(seed)
        h = src.clone()
        pos = torch.randint(0, n, (n // every,), device=dev, generator=g)
        pos = pos[h[pos] < 4]
        h[pos] = ((h[pos].long() + 1 + torch.randint(0, 3, (pos.numel(),), device=dev, generator=g)) % 4).to(torch.uint8)
        return h
    h1 = haplotype(seq, 9001, 80)
    h2 = haplotype(h1, 9002, 160)
    parts = [bench.make_reads(g_, n_chrom, chrom_len, n_reads // 3, L, bench.READS_SEED + k, dev)[0] for k, g_ in enumerate((seq, h1, h2))]
    reads = torch.cat(parts)
    reads = reads[torch.randperm(reads.shape[0], device=dev, generator=torch.Generator(device=dev).manual_seed(77))]
    n_reads = reads.shape[0]
    del h1, h2, parts
    fa = os.path.join(tmp, "reads.fa")
    bench.write_fasta(reads, fa, dev)
    in_args = ["-i", fa]
else:
    reads, truth = bench.make_reads(seq, n_chrom, chrom_len, n_reads, L, bench.READS_SEED, dev)
    fa = os.path.join(tmp, "reads.fa")
    bench.write_fasta(reads, fa, dev)
    in_args = ["-i", fa]
ix.close()
del seq, sa, reads
torch.cuda.empty_cache()
if os.environ.get("K4_REF_FASTQ") == "1":  # the same reads as FASTQ (Illumina-like names with a comment, qualities), gzipped with K4_REF_GZ=1
    def to_fastq(path, mate):
        out = path[:-3] + ".fq"
        with open(path) as f, open(out, "w") as g:
            while True:
                h = f.readline()
                if not h:
                    break
                sq = f.readline().rstrip("\n")
                g.write("@%s:7:1101:%d:%d %d:N:0:ACGT\n%s\n+\n%s\n" % (h[1:].rstrip("\n"), len(sq), mate, mate, sq, "F" * len(sq)))
        os.remove(path)
        if os.environ.get("K4_REF_GZ") == "1":
            subprocess.run(["gzip", "-1", out], check=True)
            out += ".gz"
        return out
    in_args = [to_fastq(a, 1 + (k > 1)) if a.endswith(".fa") else a for k, a in enumerate(in_args)]

SNP = os.environ.get("K4_REF_SNP") == "1"  # both programs also call SNPs (-S): the CSV files are compared
BAM = os.environ.get("K4_REF_BAM") == "1"  # both programs write BAM (+ .bai); the decoded records are compared
ref_sam, ref_log = os.path.join(tmp, "ref.bam" if BAM else "ref.sam"), os.path.join(tmp, "ref.log")
t0 = time.time()
ref_snp, gpu_snp = os.path.join(tmp, "ref.snp.csv"), os.path.join(tmp, "gpu.snp.csv")
r = subprocess.run([NGS, "kalign", "-I", sfx, "-o", ref_sam, "-T", str(threads), "-F", ref_log] + subs + extra + (["-S", ref_snp] if SNP else []) + in_args, capture_output=True)
t_ref = time.time() - t0
print("reference rc", r.returncode, "wall %.1fs" % t_ref, flush=True)
log = open(ref_log, errors="replace").read() if os.path.exists(ref_log) else ""
keep = [ln for ln in log.splitlines() if re.search(r"align|Align|loaded|Loading|core|sort|Sort|SAM|Completed|completed", ln)]
print("\n".join(keep[-40:]), flush=True)

gpu_sam = os.path.join(tmp, "gpu.bam" if BAM else "gpu.sam")
t0 = time.time()
g = subprocess.run([K4ALIGN, "-I", sfx, "-o", gpu_sam] + subs + extra + k4_extra + (["-S", gpu_snp] if SNP else []) + in_args, capture_output=True, text=True)
t_gpu = time.time() - t0
print("k4align rc", g.returncode, "wall %.1fs" % t_gpu)
print(g.stderr[-1500:], flush=True)


def body(path):
    hdr, recs = [], []
    with open(path, "rb") as f:
        for ln in f:
            (hdr if ln[:1] == b"@" else recs).append(ln)
    return [h for h in hdr if not h.startswith(b"@PG")], recs


dropin = None
if dropin_threads > 0 and os.path.exists(os.path.join(ROOT, "oracle", "_ref", "ngskit4b_k4")):
    di_sam, di_log = os.path.join(tmp, "dropin.sam"), os.path.join(tmp, "dropin.log")
    t0 = time.time()
    d = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "ngskit4b_k4"), "kalign", "-I", sfx, "-o", di_sam, "-T", str(dropin_threads),
                        "-F", di_log] + subs + extra + in_args, capture_output=True)
    t_di = time.time() - t0
    lg = open(di_log, errors="replace").read() if os.path.exists(di_log) else ""
    al = [ln for ln in lg.splitlines() if "Now aligning" in ln or "Alignment of" in ln]
    print("drop-in rc", d.returncode, "wall %.1fs" % t_di, "\n".join(al), flush=True)
    dropin = (di_sam, t_di, al)

if BAM:
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import samutil

    def body(path):  # noqa: F811  (header lines but @PG + the dictionary; one tuple per record, every field)
        text, refs, recs = samutil.read_bam(path)
        hdr = [h for h in text.splitlines() if not h.startswith("@PG")] + ["%s:%d" % r for r in refs]
        return hdr, [(r["ref"], r["pos"], r["bin"], r["mapq"], r["flag"], r["next_ref"], r["next_pos"], r["tlen"], r["name"],
                      tuple(r["cigar"]), r["seq"], bytes(r["qual"]), bytes(r["aux"])) for r in recs]
hr, rr = body(ref_sam)
hg, rg = body(gpu_sam)
same_hdr = hr == hg
same_order = rr == rg
same_set = sorted(rr) == sorted(rg)
out = {"format": "BAM" if BAM else "SAM", "extra_args": extra, "k4align_extra_args": k4_extra, "repeat_copies": n_rep, "reads": n_reads * (2 if pe_mode else 1), "pe_mode": pe_mode, "read_len": L, "genome_bp": n_chrom * chrom_len, "threads": threads, "reference_wall_s": t_ref, "k4align_wall_s": t_gpu,
       "reference_sam_records": len(rr), "k4align_sam_records": len(rg), "headers_equal": same_hdr,
       "records_equal_as_multiset": same_set, "records_equal_in_order": same_order}
if SNP:
    a, b = open(ref_snp).read().splitlines(), open(gpu_snp).read().splitlines()
    strip = lambda ln: ",".join(f for k, f in enumerate(ln.split(",")) if k != 8)  # noqa: E731  (column 8 = Rank)
    out["snp_lines_reference"] = len(a) - 1
    out["snp_lines_k4align"] = len(b) - 1
    out["snp_files_identical"] = a == b
    out["snp_files_identical_but_rank"] = [strip(x) for x in a] == [strip(x) for x in b]
    wr, wg = ref_snp[:-4] + ".covsegs.wig", gpu_snp[:-4] + ".covsegs.wig"
    if os.path.exists(wr) and os.path.exists(wg):
        out["coverage_wig_bytes"] = os.path.getsize(wr)
        out["coverage_wig_identical"] = open(wr, "rb").read() == open(wg, "rb").read()
    for key, ext in (("disnp", ".disnp.csv"), ("trisnp", ".trisnp.csv")):  # the haplotype files beside the SNP file
        hr_, hg_ = ref_snp[:-4] + ext, gpu_snp[:-4] + ext
        if os.path.exists(hr_) and os.path.exists(hg_):
            out[key + "_lines_reference"] = open(hr_).read().count("\n") - 1
            out[key + "_identical"] = open(hr_, "rb").read() == open(hg_, "rb").read()
if dropin:
    hd, rd = body(dropin[0])
    out["dropin_threads"] = dropin_threads
    out["dropin_wall_s"] = dropin[1]
    out["dropin_records_equal_as_multiset"] = sorted(rd) == sorted(rr) and hd == hr
    out["dropin_align_log"] = dropin[2]
print(json.dumps(out), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump({"summary": out, "reference_log_tail": keep[-40:], "k4align_stderr": g.stderr[-3000:]},
          open(os.path.join(ROOT, "gpurun_out", "ref_fullscale%s%s.json" % ("_pe%d" % pe_mode if pe_mode else "", ("_rep" if n_rep > 0 else "_stress" if n_rep < 0 else "") + ("_" + "".join(extra).replace("-", "") if extra else "") + ("_bam" if BAM else "") + ("_snp" if SNP else "") + ("_hap" if os.environ.get("K4_REF_HAP") == "1" else ""))), "w"), indent=1)
shutil.rmtree(tmp, ignore_errors=True)
