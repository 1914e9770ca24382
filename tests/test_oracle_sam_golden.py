"""Pins the oracle's CKAligner-level restatement (AlignRead classification, ProcCoredApprox multi-hit pairing,
ProcessPairedEnds incl. mate rescue) to what the real `ngskit4b kalign` wrote (tests/golden/sam_*.xz)."""
import json
import os

import numpy as np
import pytest

import samutil
from oracle_bindings import oracle_kalign_pe

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = json.load(open(os.path.join(GOLDEN, "sam_cases.json")))
NAR_CODE = {"AA": 1, "EN": 2, "NL": 3, "MH": 4, "ML": 5, "UI": 13, "OI": 14, "UP": 15, "IS": 16, "IT": 17, "NP": 18}
CHROMS = ["chr1", "chr2", "chr3", "chr4", "chr5"]


def kalign_args(args):
    kw = dict(max_subs=5, min_edit_dist=1, pmode=0)
    pe = dict(pe_mode=0, pair_min_len=100, pair_max_len=1000)
    for a in args:
        if a.startswith("-s"): kw["max_subs"] = int(a[2:])
        elif a.startswith("-e"): kw["min_edit_dist"] = int(a[2:])
        elif a.startswith("-m"): kw["pmode"] = int(a[2:])
        elif a.startswith("-U"): pe["pe_mode"] = int(a[2:])
        elif a.startswith("-d"): pe["pair_min_len"] = int(a[2:])
        elif a.startswith("-D"): pe["pair_max_len"] = int(a[2:])
        elif a.startswith("-R"): kw["max_ml"] = int(a[2:])
        elif a == "-r5": kw["pe_mode"] = max(kw.get("pe_mode", 0), 2)  # MLMode eMLall: every instance reported
        elif a == "-r2": kw["pe_mode"] = 2                              # MLMode eMLrand: ... then one of them picked
        elif a in ("-r3", "-r4"): kw["pe_mode"] = 1                     # eMLuniq / eMLmulti: multi-aligned reads keep their loci
        elif a == "-X": kw["pe_mode"] = max(kw.get("pe_mode", 0), 3)   # ... reads over the -R limit clamped to it
        elif a == "-N": kw["pe_mode"] = 4                               # ... through LocateBestMatches
    return kw, pe


def pick_rand_hits(out, hits):
    """MLMode eMLrand (`-r2`, KAligner.cpp:9945-9962): rand() of the C library (never seeded by kalign) once per read within
    the instance limit, in load order -- what the reference does when it runs one thread."""
    import ctypes

    libc = ctypes.CDLL(None)
    libc.srand(1)
    for o, hh in zip(out, hits):
        if int(o["nar"]) == 1 and int(o["num_hits"]) >= 1:
            hh[0] = hh[libc.rand() % int(o["num_hits"])]
            o["num_hits"] = 1


def expand_all_hits(names, reads, out, hits):
    """one entry per reported instance (NumHits of an accepted read), as CKAligner::WriteHitLoci duplicates the read"""
    n2, r2, res = [], [], []
    for nm, rd, o, hh in zip(names, reads, out, hits):
        for q in range(max(int(o["num_hits"]), 1) if int(o["nar"]) == 1 else 1):
            n2.append(nm)
            r2.append(rd)
            res.append(dict(nar=int(o["nar"]), hit=hh[q], pe_aligned=0))
    return n2, r2, res


def check_hist(nars, expect):
    got = np.bincount(np.asarray(nars), minlength=32)
    for name, code in NAR_CODE.items():
        assert got[code] == expect.get(name, 0), (name, got[code], expect.get(name, 0))


@pytest.mark.parametrize("case", sorted(c for c in CASES if c.startswith("se_")))
def test_se_matches_reference_sam(oracle, golden_dir, g2_path, case):
    kw, _ = kalign_args(CASES[case]["args"])
    names, reads = samutil.read_fasta_xz(os.path.join(golden_dir, CASES[case].get("reads", "sam_%s.fa.xz" % case)))
    h = oracle.open(g2_path if CASES[case].get("index") == "g2" else os.path.join(golden_dir, "g1.sfx"))
    oracle.set_max_iter(h, 5000 if kw["pmode"] == 0 else 10000)
    r = oracle.kalign_batch(h, reads, **kw)
    clust = [int(a[2:]) for a in CASES[case]["args"] if a in ("-r3", "-r4")]
    if clust:  # AssignMultiMatches; the reference ran with 4 threads
        multi = ((r["out"]["hit_rslt"] == 1) & (r["out"]["inst"] > 1)).sum()
        got = oracle.assign_multi_matches(r["out"], r["hits"], clust[0], max(len(x) for x in reads), threads=4)
        assert multi > 100 and 20 < got < multi
        r["out"]["num_hits"][r["out"]["nar"] != 1] = 0
    rand = "-r2" in CASES[case]["args"]
    if rand:
        assert (r["out"]["num_hits"] > 1).sum() > 5
        pick_rand_hits(r["out"], r["hits"])
    names, reads, res = expand_all_hits(names, reads, r["out"], r["hits"])
    _, recs = samutil.read_sam_xz(os.path.join(golden_dir, "sam_%s.sam.xz" % case))
    if kw.get("pe_mode", 0) >= 2 and not rand and not clust:  # the reference's own tallies are per reported locus in this mode (KAligner.cpp:571-600)
        assert CASES[case]["nar"]["AA"] == len(recs) == sum(1 for x in res if x["nar"] == 1)
        assert (r["out"]["num_hits"] > 1).sum() > 5
    else:
        check_hist(r["out"]["nar"], CASES[case]["nar"])
    assert sorted(samutil.sam_records(names, reads, res, CHROMS)) == sorted(recs)
    oracle.close(h)


@pytest.mark.parametrize("case", sorted(c for c in CASES if c.startswith("pe_")))
def test_pe_matches_reference_sam(oracle, golden_dir, case):
    kw, pe = kalign_args(CASES[case]["args"])
    n1, r1 = samutil.read_fasta_xz(os.path.join(golden_dir, "sam_%s_1.fa.xz" % case))
    n2, r2 = samutil.read_fasta_xz(os.path.join(golden_dir, "sam_%s_2.fa.xz" % case))
    h = oracle.open(os.path.join(golden_dir, "g1.sfx"))
    oracle.set_max_iter(h, 5000)
    out = oracle_kalign_pe(oracle, h, r1, r2, threads=4, **pe, **kw)
    check_hist(out["nar"], CASES[case]["nar"])
    names = [x for p in zip(n1, n2) for x in p]
    reads = [x for p in zip(r1, r2) for x in p]
    res = [dict(nar=int(o["nar"]), hit=o["hit"], pe_aligned=int(o["pe_aligned"])) for o in out]
    _, recs = samutil.read_sam_xz(os.path.join(golden_dir, "sam_%s.sam.xz" % case))
    assert sorted(samutil.sam_records(names, reads, res, CHROMS, paired=True)) == sorted(recs)
    if case == "pe_u1":
        assert out["rescued"].sum() > 0  # the mate-rescue path (AlignPairedRead) is exercised
    oracle.close(h)
