"""kalign's SNP calling (CKAligner::ProcessSNPs / OutputSNPs, ngskit4b/KAligner.cpp:8168-8590, 7098-7760), main CSV: the CPU
oracle against what `ngskit4b kalign -p -P -S` wrote (tests/golden/snp_*.csv, make_golden_snp.py).  No GPU."""
import json
import os

import numpy as np
import pytest

import samutil
from oracle_bindings import oracle_kalign_pe
from test_oracle_sam_golden import kalign_args

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SNP_CASES = json.load(open(os.path.join(GOLDEN, "snp_cases.json")))


def snp_args(args):
    o = dict(min_snp_reads=5, qvalue=0.05, snp_nonref_pcnt=25.0)
    for a in args:
        if a.startswith("-p"): o["min_snp_reads"] = int(a[2:])
        elif a.startswith("-P"): o["qvalue"] = float(a[2:])
        elif a.startswith("-1"): o["snp_nonref_pcnt"] = float(a[2:])
    return o


def trim_reads(args, reads):
    """-y / -Y: bases taken off the reads' ends when loading (KAligner.cpp:12254-12260)"""
    y = sum(int(a[2:]) for a in args if a.startswith("-y"))
    Y = sum(int(a[2:]) for a in args if a.startswith("-Y"))
    return [r[y:len(r) - Y] for r in reads] if y or Y else reads


def rows(text):
    """CSV rows as tuples with the Rank column (index 8) split off: equal p-values have no defined order in the reference's sort"""
    out, ranks = [], []
    for ln in text.splitlines()[1:]:
        f = ln.split(",")
        ranks.append((f[9], int(f[8])))
        out.append(tuple(f[:8] + f[9:]))
    return out, ranks


def aligned_inputs(oracle, h, case):
    """(reads, nar, hits) of the case as CKAligner leaves them for the SNP pass: the oracle's own alignment (pinned elsewhere by
    the reference's SAM of the same run, which is checked here too)"""
    args = [a for a in SNP_CASES[case]["args"] if a[:2] not in ("-p", "-P", "-1")]
    kw, pe = kalign_args(args)
    if case.startswith("snp_pe"):
        n1, r1 = samutil.read_fasta_xz(os.path.join(GOLDEN, case + "_1.fa.xz"))
        n2, r2 = samutil.read_fasta_xz(os.path.join(GOLDEN, case + "_2.fa.xz"))
        out = oracle_kalign_pe(oracle, h, r1, r2, threads=4, **pe, **kw)
        names = [x for p in zip(n1, n2) for x in p]
        reads = [x for p in zip(r1, r2) for x in p]
        res = [dict(nar=int(o["nar"]), hit=o["hit"], pe_aligned=int(o["pe_aligned"])) for o in out]
        got = samutil.sam_records(names, reads, res, ["chr1", "chr2", "chr3", "chr4", "chr5"], paired=True)
        nar, hits = out["nar"].copy(), out["hit"].copy()
    else:
        names, reads = samutil.read_fasta_xz(os.path.join(GOLDEN, case + ".fa.xz"))
        reads = trim_reads(args, reads)
        if "min_chimeric_len" in kw:
            r = oracle.kalign_ext_batch(h, reads, **kw)
        else:
            r = oracle.kalign_batch(h, reads, **kw)
        res = [dict(nar=int(o["nar"]), hit=hh[0]) for o, hh in zip(r["out"], r["hits"])]
        got = samutil.sam_records(names, reads, res, ["chr1", "chr2", "chr3", "chr4", "chr5"])
        nar, hits = r["out"]["nar"].copy(), r["hits"][:, 0].copy()
    _, recs = samutil.read_sam_xz(os.path.join(GOLDEN, case + ".sam.xz"))
    assert sorted(got) == sorted(recs)  # the alignments the SNP pass starts from are the reference's
    return reads, nar, hits


@pytest.mark.parametrize("case", sorted(SNP_CASES))
def test_snp_csv_matches_the_reference(oracle, golden_dir, case):
    h = oracle.open(os.path.join(golden_dir, "g1.sfx"))
    oracle.set_max_iter(h, 5000)
    reads, nar, hits = aligned_inputs(oracle, h, case)
    text, n = oracle.snp_csv(h, reads, nar, hits, **snp_args(SNP_CASES[case]["args"]))
    want = open(os.path.join(golden_dir, case + ".csv")).read()
    assert n == SNP_CASES[case]["snps"] == len(want.splitlines()) - 1
    assert text.splitlines()[0] == want.splitlines()[0]
    got_rows, got_ranks = rows(text)
    want_rows, want_ranks = rows(want)
    assert got_rows == want_rows
    # ranks: identical where p-values are distinct; within a group of equal printed p-values the same multiset
    assert sorted(got_ranks) == sorted(want_ranks)
    oracle.close(h)


@pytest.mark.parametrize("case", ["snp_se_c50_p8", "snp_pe_u1", "snp_pe_hap_c60"])
def test_snp_vcf_records_match_the_reference(oracle, golden_dir, case):
    """the same calls in kalign's VCF form (`-S x.vcf`): records identical; the header lines name the writer's version and the index"""
    h = oracle.open(os.path.join(golden_dir, "g1.sfx"))
    oracle.set_max_iter(h, 5000)
    reads, nar, hits = aligned_inputs(oracle, h, case)
    text, n = oracle.snp_csv(h, reads, nar, hits, vcf=True, **snp_args(SNP_CASES[case]["args"]))
    want = [l for l in open(os.path.join(golden_dir, case + ".vcf")).read().splitlines() if not l.startswith("#")]
    assert text.splitlines() == want and n == len(want) == SNP_CASES[case]["snps"]
    oracle.close(h)


@pytest.mark.parametrize("case", sorted(SNP_CASES))
def test_coverage_wig_matches_the_reference(oracle, golden_dir, case):
    """<snp file>.covsegs.wig: variableStep spans of roughly equal coverage (AccumWIGCnts / CompleteWIGSpan, KAligner.cpp:6993-7085),
    incl. what the reference loses: a span that starts at locus 0, and the last span of a chromosome without a candidate locus"""
    import lzma

    h = oracle.open(os.path.join(golden_dir, "g1.sfx"))
    oracle.set_max_iter(h, 5000)
    reads, nar, hits = aligned_inputs(oracle, h, case)
    text = oracle.snp_wig(h, reads, nar, hits, **snp_args(SNP_CASES[case]["args"]))
    want = lzma.open(os.path.join(golden_dir, case + ".covsegs.wig.xz")).read().decode()
    assert text == want
    oracle.close(h)


@pytest.mark.parametrize("case", sorted(SNP_CASES))
@pytest.mark.parametrize("n_loci", [2, 3])
def test_haplotype_files_match_the_reference(oracle, golden_dir, case, n_loci):
    """<snp file>.disnp.csv / .trisnp.csv (KAligner.cpp:7767-8101): two / three called SNP loci within min(300, mean aligned length)
    bases, the reads covering all of them and the count of each base combination -- byte for byte (the *_hap cases fill them; the
    others hold the header alone)"""
    h = oracle.open(os.path.join(golden_dir, "g1.sfx"))
    oracle.set_max_iter(h, 5000)
    reads, nar, hits = aligned_inputs(oracle, h, case)
    text = oracle.snp_haplotypes(h, n_loci, reads, nar, hits, **snp_args(SNP_CASES[case]["args"]))
    want = open(os.path.join(golden_dir, case + (".disnp.csv" if n_loci == 2 else ".trisnp.csv"))).read()
    assert text == want
    if case.endswith("_hap") or "_hap_" in case:
        assert len(want.splitlines()) > 50
    oracle.close(h)
