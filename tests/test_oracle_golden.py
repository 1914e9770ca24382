"""Pins the CPU restatement (oracle/k4oracle.c) to the golden vectors captured from the real reference
(tests/golden/make_golden.py) and, where oracle/_ref was built, to the live reference on fresh inputs."""
import glob
import os

import numpy as np
import pytest

import synth
from oracle_bindings import HIT_DTYPE, Oracle, Ref, ref_available

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = sorted(os.path.basename(p)[6:-4] for p in glob.glob(os.path.join(GOLDEN, "align_*.npz"))
               if not os.path.basename(p).startswith("align_ext_"))  # (the optional-phase vectors: test_*_ext.py)


def check_against(res, exp, max_hits):
    for k in ("rslt", "inst", "low", "nxt"):
        assert np.array_equal(res[k], exp[k]), (k, np.nonzero(res[k] != exp[k])[0][:8])
    for i in range(len(res["rslt"])):
        nh = min(int(exp["inst"][i]), max_hits) if exp["rslt"][i] in (1, 2, 3) else 0
        assert np.array_equal(res["hits"][i, :nh], exp["hits"][i, :nh]), (i, res["hits"][i, :nh], exp["hits"][i, :nh])


def test_cases_present():
    assert len(CASES) >= 12


@pytest.mark.parametrize("case", CASES)
def test_oracle_matches_reference_golden(oracle, golden_dir, g1_el5_path, case):
    g = np.load(os.path.join(golden_dir, "align_%s.npz" % case))
    tm, cl, cd, sl, mcl, md, strand, mh, maxiter = [int(x) for x in g["params"]]
    h = oracle.open(g1_el5_path if case.endswith("_el5") else os.path.join(golden_dir, "g1.sfx"))
    assert oracle.el_size(h) == (5 if case.endswith("_el5") else 4)
    oracle.set_max_iter(h, maxiter)
    res = oracle.align_reads_batch(h, (g["reads"], g["offs"], g["lens"]), tm, cl, cd, sl, mcl, md, strand, mh)
    check_against(res, g, mh)
    oracle.close(h)


def test_known_answers_survey_appendix_c(oracle, golden_dir):
    """(Rslt,inst,LowMM,NxtLowMM) for a unique read with 0..3 substitutions at C2 parameters."""
    h = oracle.open(os.path.join(golden_dir, "g1.sfx"))
    seq = np.array(oracle.seq(h))
    ents = oracle.entries(h)
    e = ents[1]
    rng = np.random.default_rng(5)
    for _ in range(20):
        start = int(rng.integers(0, e["seq_len"] - 100))
        rd = seq[e["start_ofs"] + start: e["start_ofs"] + start + 100].copy()
        if (rd > 3).any():
            continue
        base = oracle.align_reads_batch(h, [rd], 2, 33, 33, 8, 0, 1, 0, 1)
        if base["inst"][0] != 1 or base["nxt"][0] != 2:
            continue  # landed in a planted repeat
        expect = {0: (1, 1, 0, 2), 1: (1, 1, 1, 3), 2: (1, 1, 2, 4), 3: (0, 0, 4, 4)}
        for ns, exp in expect.items():
            r2 = rd.copy()
            for k in range(ns):
                r2[7 + 31 * k] = (r2[7 + 31 * k] + 1) % 4
            res = oracle.align_reads_batch(h, [r2], 2, 33, 33, 8, 0, 1, 0, 1)
            got = (res["rslt"][0], res["inst"][0], res["low"][0], res["nxt"][0])
            assert got == exp, (ns, got)
            if ns <= 2:
                hit = res["hits"][0, 0]
                assert (hit["chrom_id"], hit["match_loci"], chr(hit["strand"]), hit["mismatches"]) == (2, start, "+", ns)
    oracle.close(h)


def test_sfx_reader_and_writer_roundtrip(oracle, golden_dir, tmp_path):
    """Container layout: reading the reference-written file and re-writing it reproduces every
    byte outside the free-text header fields."""
    src = os.path.join(golden_dir, "g1.sfx")
    h = oracle.open(src)
    assert oracle.concat_len(h) == 125425 and oracle.el_size(h) == 4
    ents = oracle.entries(h)
    assert [e["name"] for e in ents] == ["chr1", "chr2", "chr3", "chr4", "chr5"]
    assert ents[1]["start_ofs"] == 60001 and ents[1]["end_ofs"] == 100000
    out = str(tmp_path / "rt.sfx")
    oracle.write(h, out)
    a, b = open(src, "rb").read(), open(out, "rb").read()
    assert len(a) == len(b) == 1224 + 20 + 5 * 125425 + 8 + 111 * 5
    assert a[:133] == b[:133]          # magic .. dataset name (numeric header fields identical)
    assert a[1224:] == b[1224:]        # block, suffix array and entries byte-identical
    oracle.close(h)


def test_own_suffix_sort_matches_reference_modulo_ties(oracle, golden_dir):
    """Q9: an independent builder matches the reference SA except inside groups of suffixes that are identical
    through their EOS (the reference's parallel quicksort leaves those in arbitrary order)."""
    names, chroms = synth.golden_genome()
    hb = oracle.build(names, chroms, dataset="g1")
    hf = oracle.open(os.path.join(golden_dir, "g1.sfx"))
    assert np.array_equal(oracle.seq(hb), oracle.seq(hf))
    sa_b, sa_f = oracle.sa(hb), oracle.sa(hf)
    assert np.array_equal(np.sort(sa_b), np.arange(len(sa_b)))
    diff = np.nonzero(sa_b != sa_f)[0]
    seq = np.array(oracle.seq(hf))

    def through_eos(p):
        e = p
        while seq[e] != 7:
            e += 1
        return bytes(seq[p:e + 1])

    assert len(diff) < 50
    for i in diff:
        assert through_eos(sa_b[i]) == through_eos(sa_f[i])
    oracle.close(hb)
    oracle.close(hf)


def test_min_core_len(oracle, golden_dir):
    """KAligner.cpp:9367-9393: 5 Mbp -> 11, 200 Mbp -> 13 (oracle log), here 125 kbp -> 8."""
    h = oracle.open(os.path.join(golden_dir, "g1.sfx"))
    assert oracle.min_core_len(h, 0) == (8, 8)
    assert oracle.min_core_len(h, 2) == (6, 9)
    assert oracle.min_core_len(h, 3) == (10, 6)
    oracle.close(h)


def test_kalign_level_nar(oracle, golden_dir):
    """AlignRead classification on the golden C2 reads: NAR histogram consistent with the AlignReads results."""
    g = np.load(os.path.join(golden_dir, "align_c2_s2.npz"))
    h = oracle.open(os.path.join(golden_dir, "g1.sfx"))
    oracle.set_max_iter(h, 5000)
    r = oracle.kalign_batch(h, (g["reads"], g["offs"], g["lens"]), max_subs=2, min_core_len=8, max_num_slides=8)
    out = r["out"]
    # reads with too many N are EN; everything else must agree with the raw AlignReads golden at CoreLen 33
    lens = g["lens"]
    nN = np.array([(g["reads"][int(o):int(o) + int(l)] == 4).sum() for o, l in zip(g["offs"], lens)])
    en = nN > 1
    assert np.array_equal(out["nar"] == 2, en)
    ok = ~en
    assert np.array_equal(out["hit_rslt"][ok], g["rslt"][ok])
    assert np.array_equal(out["nar"][ok & (g["rslt"] == 1)], np.full((ok & (g["rslt"] == 1)).sum(), 1))
    assert np.array_equal(out["nar"][ok & (g["rslt"] == 0)], np.full((ok & (g["rslt"] == 0)).sum(), 3))
    assert np.array_equal(out["nar"][ok & (g["rslt"] == 3)], np.full((ok & (g["rslt"] == 3)).sum(), 5))
    acc = ok & (g["rslt"] == 1)
    assert np.array_equal(r["hits"][acc, 0], g["hits"][acc, 0])
    oracle.close(h)


@pytest.mark.skipif(not ref_available(), reason="oracle/_ref not built (no /root/reference here)")
def test_oracle_matches_live_reference_on_fresh_inputs(oracle, tmp_path):
    R = Ref()
    names, chroms = synth.make_genome([30000, 20000, 500], seed=99, repeats=25, repeat_len=180, repeat_div=0.02,
                                      n_runs=3, tandem=4)
    path = str(tmp_path / "fresh.sfx")
    R.build_sfx(path, names, chroms)
    hr = R.open(path, 40, 0)
    ho = oracle.open(path)
    oracle.set_max_iter(ho, 40)
    for (rl, tm, cl, cd, sl, mh, md, strand) in [(100, 2, 33, 33, 8, 1, 1, 0), (120, 4, 24, 24, 10, 6, 2, 0),
                                                 (80, 1, 40, 40, 7, 2, 1, 2)]:
        reads, _ = synth.make_reads(chroms, 700, rl, seed=rl * 7 + tm, n_prob=0.05, edge_frac=0.1, random_frac=0.05)
        ro = oracle.align_reads_batch(ho, reads, tm, cl, cd, sl, 0, md, strand, mh)
        rr = R.align_reads_batch(hr, reads, tm, cl, cd, sl, 0, md, strand, mh)
        check_against(ro, rr, mh)
    R.close(hr)
    oracle.close(ho)
