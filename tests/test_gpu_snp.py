"""kalign's SNP calling on the device (k4_snp_csv_dev: pile-up, window sums and per-locus tests as kernels; p-values, ranks and text
on the host) against the SNP files `ngskit4b kalign -p -P -S` wrote (tests/golden/snp_*.csv) -- through the API on the device's own
alignments, and through the k4align program.  CKAligner::ProcessSNPs / OutputSNPs, ngskit4b/KAligner.cpp:8168-8590, 7098-7760."""
import json
import lzma
import os
import subprocess

import numpy as np
import pytest

import samutil
from test_oracle_snp import SNP_CASES, snp_args, trim_reads
from test_oracle_sam_golden import kalign_args

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def k4():
    import kit4b_amd

    kit4b_amd.lib()  # raises if the HIP extension is missing: no fallback
    return kit4b_amd


@pytest.mark.parametrize("case", sorted(SNP_CASES))
def test_snp_csv_through_the_api(k4, golden_dir, case):
    args = [a for a in SNP_CASES[case]["args"] if a[:2] not in ("-p", "-P", "-1")]
    kw, pe = kalign_args(args)
    ix = k4.SfxIndex.open(os.path.join(golden_dir, "g1.sfx"))
    ix.set_max_iter(5000)
    if case.startswith("snp_pe"):
        _, r1 = samutil.read_fasta_xz(os.path.join(golden_dir, case + "_1.fa.xz"))
        _, r2 = samutil.read_fasta_xz(os.path.join(golden_dir, case + "_2.fa.xz"))
        out = ix.kalign_pe_batch(r1, r2, **pe, **kw)
        reads = [x for p in zip(r1, r2) for x in p]
        text, n = ix.snp_csv(reads, pe_recs=out, **snp_args(SNP_CASES[case]["args"]))
        files = ix.snp_files(reads, pe_recs=out, **snp_args(SNP_CASES[case]["args"]))
    else:
        _, reads = samutil.read_fasta_xz(os.path.join(golden_dir, case + ".fa.xz"))
        reads = trim_reads(args, reads)
        r = ix.kalign_ext_batch(reads, **kw) if "min_chimeric_len" in kw else ix.kalign_batch(reads, **kw)
        text, n = ix.snp_csv(reads, out=r["out"], hits=r["hits"], **snp_args(SNP_CASES[case]["args"]))
        files = ix.snp_files(reads, out=r["out"], hits=r["hits"], **snp_args(SNP_CASES[case]["args"]))
    want = open(os.path.join(golden_dir, case + ".csv")).read()
    assert n == SNP_CASES[case]["snps"]
    assert text == want
    # every file of the run through k4_snp_run_dev: the same CSV, the coverage WIG and the DiSNP / TriSNP haplotype files
    assert files["snp"] == want and files["n_snps"] == n
    assert files["wig"] == lzma.open(os.path.join(golden_dir, case + ".covsegs.wig.xz")).read().decode()
    assert files["disnp"] == open(os.path.join(golden_dir, case + ".disnp.csv")).read()
    assert files["trisnp"] == open(os.path.join(golden_dir, case + ".trisnp.csv")).read()
    ix.close()


@pytest.mark.parametrize("case", sorted(SNP_CASES))
def test_k4align_writes_the_reference_snp_file(golden_dir, tmp_path, case):
    def unxz(name):
        dst = str(tmp_path / name[:-3])
        open(dst, "wb").write(lzma.open(os.path.join(golden_dir, name)).read())
        return dst

    files = ["-i", unxz(case + "_1.fa.xz"), "-u", unxz(case + "_2.fa.xz")] if case.startswith("snp_pe") else ["-i", unxz(case + ".fa.xz")]
    out, snp = str(tmp_path / "o.sam"), str(tmp_path / "o.csv")
    p = subprocess.run([os.path.join(ROOT, "kit4b_amd", "k4align"), "-I", os.path.join(golden_dir, "g1.sfx"), "-o", out, "-S", snp]
                       + SNP_CASES[case]["args"] + files, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    assert open(snp).read() == open(os.path.join(golden_dir, case + ".csv")).read()
    assert ("with %d putative SNPs discovered" % SNP_CASES[case]["snps"]) in p.stderr
    # the coverage WIG beside it (<snp file minus extension>.covsegs.wig), spans walked on host threads
    assert open(str(tmp_path / "o.covsegs.wig")).read() == lzma.open(os.path.join(golden_dir, case + ".covsegs.wig.xz")).read().decode()
    # and the haplotype files (<snp file minus extension>.disnp.csv / .trisnp.csv)
    for ext in (".disnp.csv", ".trisnp.csv"):
        assert open(str(tmp_path / ("o" + ext))).read() == open(os.path.join(golden_dir, case + ext)).read()
    got = [l for l in open(out).read().splitlines() if not l.startswith("@")]
    want = [l for l in lzma.open(os.path.join(golden_dir, case + ".sam.xz")).read().decode().splitlines() if not l.startswith("@")]
    assert sorted(got) == sorted(want)


def test_snp_option_rules(golden_dir, tmp_path):
    exe = os.path.join(ROOT, "kit4b_amd", "k4align")
    fa = tmp_path / "r.fa"
    fa.write_text(">r1\n" + "ACGT" * 25 + "\n")
    base = [exe, "-I", os.path.join(golden_dir, "g1.sfx"), "-i", str(fa), "-o", str(tmp_path / "o.sam")]
    p = subprocess.run(base + ["-p500"], capture_output=True, text=True, timeout=60)
    assert p.returncode == 1 and "range 1..100" in p.stderr
    p = subprocess.run(base + ["-p5", "-P0.5"], capture_output=True, text=True, timeout=60)
    assert p.returncode == 1 and "QValue" in p.stderr
    p = subprocess.run(base + ["-p5", "-r5", "-R4"], capture_output=True, text=True, timeout=60)
    assert p.returncode == 1 and "multiloci" in p.stderr
    p = subprocess.run(base + ["-p5", "-b", "1"], capture_output=True, text=True, timeout=60)
    assert p.returncode == 3
    p = subprocess.run(base + ["-p5"], capture_output=True, text=True, timeout=120)  # no alignment, no SNP: header only, default name
    assert p.returncode == 0, p.stderr
    assert open(str(tmp_path / "o.sam") + ".snp").read().count("\n") == 1


@pytest.mark.parametrize("case", ["snp_se_c50_p8", "snp_pe_u1", "snp_pe_hap_c60"])
def test_k4align_writes_the_reference_vcf(golden_dir, tmp_path, case):
    """`-S x.vcf`: the VCF form -- records identical to the reference's, header lines but ##source / ##reference too"""
    def unxz(name):
        dst = str(tmp_path / name[:-3])
        open(dst, "wb").write(lzma.open(os.path.join(golden_dir, name)).read())
        return dst

    files = ["-i", unxz(case + "_1.fa.xz"), "-u", unxz(case + "_2.fa.xz")] if case.startswith("snp_pe") else ["-i", unxz(case + ".fa.xz")]
    vcf = str(tmp_path / "o.VCF")
    p = subprocess.run([os.path.join(ROOT, "kit4b_amd", "k4align"), "-I", os.path.join(golden_dir, "g1.sfx"), "-o", str(tmp_path / "o.sam"), "-S", vcf]
                       + SNP_CASES[case]["args"] + files, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    keep = lambda t: [l for l in t.splitlines() if not l.startswith(("##source", "##reference"))]  # noqa: E731
    assert keep(open(vcf).read()) == keep(open(os.path.join(golden_dir, case + ".vcf")).read())
