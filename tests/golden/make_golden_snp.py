"""Golden SNP calls from the REAL reference front end (`oracle/_ref/ngskit4b kalign -p<n> -P<q> -S x.csv`).

    python tests/golden/make_golden_snp.py

Reads are drawn from a copy of the golden genome g1 with a substitution every ~700 bases (so that alignments pile up mismatches
at fixed loci), ~10-fold coverage, plus sequencing errors; the reference aligns them to g1 and calls SNPs.  Kept per case: the reads
(FASTA, xz), the command line (snp_cases.json), the SAM (xz) and the SNP CSV the reference wrote.  Data only."""
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
CASES = {
    "snp_se": dict(args=["-s3", "-p5", "-P0.05"], n=12000, L=100, seed=77, pe=False),
    "snp_se_c50_p8": dict(args=["-s3", "-c50", "-p8", "-P0.2", "-110.0"], n=16000, L=120, seed=78, pe=False),
    "snp_pe_u1": dict(args=["-s3", "-U1", "-d200", "-D600", "-p6", "-P0.05"], n=5000, L=125, seed=79, pe=True),
    # three haplotypes (the genome itself, one with a substitution every ~60 bases, one with those and more): SNP loci close enough
    # for the DiSNP / TriSNP files to fill
    # end trims and one strand only in front of the SNP stage (the pile-up starts from the trimmed reads)
    "snp_se_y5_Y7_Q1": dict(args=["-s3", "-y5", "-Y7", "-Q1", "-p5", "-P0.05"], n=12000, L=100, seed=82, pe=False),
    "snp_se_hap": dict(args=["-s6", "-p5", "-P0.05"], n=9000, L=100, seed=80, pe=False, hap=True),
    "snp_pe_hap_c60": dict(args=["-s6", "-c60", "-U1", "-d200", "-D600", "-p5", "-P0.05"], n=3500, L=110, seed=81, pe=True, hap=True),
}


def mutated_genome(chroms, seed):
    rng = np.random.default_rng(seed)
    mut = [c.copy() for c in chroms]
    for c in mut[:3]:
        pos = rng.choice(len(c) - 200, size=len(c) // 700, replace=False) + 100
        for p in pos:
            c[p] = (c[p] + 1 + rng.integers(0, 3)) % 4
    return mut


def haplotypes(chroms, seed):
    def sub(src, seed, every):
        rng = np.random.default_rng(seed)
        mut = [c.copy() for c in src]
        for c in mut[:3]:
            pos = rng.choice(len(c) - 200, size=len(c) // every, replace=False) + 100
            for p in pos:
                c[p] = (c[p] + 1 + rng.integers(0, 3)) % 4
        return mut
    h1 = sub(chroms, seed, 60)
    return [chroms, h1, sub(h1, seed + 1000, 120)]


def main():
    names, chroms = synth.golden_genome()
    meta = {}
    with tempfile.TemporaryDirectory() as tmp:
        for name, c in CASES.items():
            mut = mutated_genome(chroms, c["seed"])
            haps = haplotypes(chroms, c["seed"]) if c.get("hap") else [mut]
            files = []
            if c["pe"]:
                pe1, pe2 = [], []
                for k, g in enumerate(haps):
                    a, b, _ = synth.make_pe_reads(g, c["n"], c["L"], seed=c["seed"] + 1 + k, sub_lambda=0.8 if len(haps) == 1 else 0.4, n_prob=0.01)
                    pe1 += a
                    pe2 += b
                for k, rd, flag in (("1", pe1, "-i"), ("2", pe2, "-u")):
                    fa = os.path.join(tmp, "%s_%s.fa" % (name, k))
                    synth.write_fasta(fa, rd)
                    files += [flag, fa]
                    with open(fa, "rb") as f, lzma.open(os.path.join(HERE, "%s_%s.fa.xz" % (name, k)), "wb", preset=9) as g:
                        g.write(f.read())
            else:
                reads = []
                for k, g in enumerate(haps):
                    reads += synth.make_reads(g, c["n"], c["L"], seed=c["seed"] + 1 + k, sub_lambda=0.8 if len(haps) == 1 else 0.4, n_prob=0.01)[0]
                fa = os.path.join(tmp, name + ".fa")
                synth.write_fasta(fa, reads)
                files = ["-i", fa]
                with open(fa, "rb") as f, lzma.open(os.path.join(HERE, name + ".fa.xz"), "wb", preset=9) as g:
                    g.write(f.read())
            sam, csv = os.path.join(tmp, name + ".sam"), os.path.join(tmp, name + ".csv")
            subprocess.run([NGS, "kalign", "-I", os.path.join(HERE, "g1.sfx"), "-o", sam, "-T", "4", "-F", os.path.join(tmp, name + ".log"),
                            "-S", csv] + c["args"] + files, check=True, capture_output=True, timeout=600)
            with open(sam, "rb") as f, lzma.open(os.path.join(HERE, name + ".sam.xz"), "wb", preset=9) as g:
                g.write(f.read())
            text = open(csv).read()
            open(os.path.join(HERE, name + ".csv"), "w").write(text)
            with open(os.path.join(tmp, name + ".covsegs.wig"), "rb") as f, lzma.open(os.path.join(HERE, name + ".covsegs.wig.xz"), "wb", preset=9) as g:
                g.write(f.read())  # the coverage WIG kalign writes beside the SNP file
            for ext in (".disnp.csv", ".trisnp.csv"):  # the haplotype files kalign writes beside the SNP file
                open(os.path.join(HERE, name + ext), "w").write(open(os.path.join(tmp, name + ext)).read())
            if name not in ("snp_se", "snp_se_hap", "snp_se_y5_Y7_Q1"):  # the same calls as VCF (a SNP file name ending in .vcf)
                vcf = os.path.join(tmp, name + ".vcf")
                subprocess.run([NGS, "kalign", "-I", os.path.join(HERE, "g1.sfx"), "-o", sam, "-T", "4", "-F", os.path.join(tmp, name + ".log"),
                                "-S", vcf] + c["args"] + files, check=True, capture_output=True, timeout=600)
                open(os.path.join(HERE, name + ".vcf"), "w").write(open(vcf).read())
            meta[name] = dict(args=c["args"], snps=len(text.splitlines()) - 1)
            print(name, meta[name])
    json.dump(meta, open(os.path.join(HERE, "snp_cases.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
