"""FASTQ qualities in the output (kalign -g0..2): golden SAM / BAM files from the REAL reference front end (`oracle/_ref/ngskit4b`).

    python tests/golden/make_golden_qual.py

For each case: the reads as FASTQ (xz) with quality lines that exercise the scaling (whole range of the encoding, characters
outside it, reads whose scores all scale to zero), the SAM the reference wrote with -g<n> (xz), one BAM.  Data only; the index is
tests/golden/g1.sfx.  Cases and their arguments: tests/golden/qual_cases.json."""
import json
import lzma
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(HERE))
import synth  # noqa: E402

NGS = os.path.join(ROOT, "oracle", "_ref", "ngskit4b")


def fastq(reads, rng, lo, hi, tag, zero_every=17):
    """records '@<tag>%06d' with qualities uniform in [lo, hi]; every zero_every-th read gets the encoding's lowest character
    throughout (its scores all scale to 0: QUAL `*`), a few characters lie outside the encoding's range"""
    out = []
    for i, r in enumerate(reads):
        L = len(r)
        q = rng.integers(lo, hi + 1, L)
        if i % zero_every == 0:
            q[:] = lo
        if i % 29 == 3:
            q[rng.integers(0, L)] = 126  # '~': above every encoding's range
        if i % 31 == 5:
            q[rng.integers(0, L)] = 35   # '#': below the Illumina / Solexa ranges
        out.append("@%s%06d\n%s\n+\n%s\n" % (tag, i, "".join("ACGTN"[min(int(b), 4)] for b in r), "".join(chr(int(c)) for c in q)))
    return "".join(out).encode()


def run(args, files, out, threads="4"):
    log = out + ".log"
    subprocess.run([NGS, "kalign", "-I", os.path.join(HERE, "g1.sfx"), "-o", out, "-T", threads, "-F", log] + args + files, check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


def main():
    names, chroms = synth.golden_genome()
    rng = np.random.default_rng(20261005)
    cases = {}
    with tempfile.TemporaryDirectory() as tmp:
        def put(name, data):
            p = os.path.join(tmp, name)
            open(p, "wb").write(data)
            lzma.open(os.path.join(HERE, name + ".xz"), "wb").write(data)
            return p

        # SE, Sanger / Illumina 1.8+ (-g0)
        r = synth.make_reads(chroms, 1200, 100, seed=6101, n_prob=0.03, edge_frac=0.05, random_frac=0.04)[0]
        f = put("qual_se_g0.fq", fastq(r, rng, 33, 74, "s"))
        run(["-s2", "-g0"], ["-i", f], os.path.join(tmp, "o.sam"))
        lzma.open(os.path.join(HERE, "qual_se_g0.sam.xz"), "wb").write(open(os.path.join(tmp, "o.sam"), "rb").read())
        cases["se_g0"] = {"args": ["-s2", "-g0"], "reads": ["qual_se_g0.fq.xz"], "sam": "qual_se_g0.sam.xz"}
        run(["-s2", "-g0"], ["-i", f], os.path.join(tmp, "o.bam"))
        open(os.path.join(HERE, "qual_se_g0.bam"), "wb").write(open(os.path.join(tmp, "o.bam"), "rb").read())
        cases["se_g0"]["bam"] = "qual_se_g0.bam"
        # SE, every loaded read reported (-M1): unaligned records carry their scores too; Solexa scaling (-g2)
        r = synth.make_reads(chroms, 900, 120, seed=6102, n_prob=0.05, edge_frac=0.05, random_frac=0.10, sub_lambda=1.5)[0]
        f = put("qual_se_g2_M1.fq", fastq(r, rng, 59, 104, "x"))
        run(["-s2", "-g2", "-M1"], ["-i", f], os.path.join(tmp, "o2.sam"))
        lzma.open(os.path.join(HERE, "qual_se_g2_M1.sam.xz"), "wb").write(open(os.path.join(tmp, "o2.sam"), "rb").read())
        cases["se_g2_M1"] = {"args": ["-s2", "-g2", "-M1"], "reads": ["qual_se_g2_M1.fq.xz"], "sam": "qual_se_g2_M1.sam.xz"}
        # PE, Illumina 1.3+ (-g1), orphan rescue on (-U1)
        p1, p2, _ = synth.make_pe_reads(chroms, 700, 100, seed=6103)
        f1 = put("qual_pe_g1_1.fq", fastq(p1, rng, 64, 104, "p"))
        f2 = put("qual_pe_g1_2.fq", fastq(p2, rng, 64, 104, "p", zero_every=13))
        run(["-s2", "-g1", "-U1", "-d100", "-D600"], ["-i", f1, "-u", f2], os.path.join(tmp, "o3.sam"))
        lzma.open(os.path.join(HERE, "qual_pe_g1.sam.xz"), "wb").write(open(os.path.join(tmp, "o3.sam"), "rb").read())
        cases["pe_g1"] = {"args": ["-s2", "-g1", "-U1", "-d100", "-D600"], "reads": ["qual_pe_g1_1.fq.xz", "qual_pe_g1_2.fq.xz"], "sam": "qual_pe_g1.sam.xz"}
    json.dump(cases, open(os.path.join(HERE, "qual_cases.json"), "w"), indent=1)
    for k, v in cases.items():
        n = sum(1 for l in lzma.open(os.path.join(HERE, v["sam"])).read().decode().splitlines() if not l.startswith("@"))
        print(k, v["args"], n, "records")


if __name__ == "__main__":
    main()
