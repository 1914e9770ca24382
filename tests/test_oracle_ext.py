"""Pins oracle/k4oracle_ext.c -- the restatement of the OPTIONAL phases of CSfxArray::AlignReads (chimeric trimming,
microInDels, splice junctions) -- to the vectors the real reference returned (tests/golden/make_golden_ext.py)
and, where oracle/_ref was built, to the live reference on fresh inputs."""
import glob
import os

import numpy as np
import pytest

import synth
from oracle_bindings import (EXT_CHIMERIC, EXT_INDEL, EXT_INSERT, EXT_SPLICE, NAR_ACCEPTED, NAR_MICROINDEL, NAR_SPLICEJCTN,
                             NAR_TRIM, Oracle, Ref, ext_trims, ref_available)

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = sorted(os.path.basename(p)[10:-4] for p in glob.glob(os.path.join(GOLDEN, "align_ext_*.npz")))
KEYS = ("tot_mm", "core_len", "core_delta", "max_slides", "min_core_len", "mm_delta", "strand", "max_hits",
        "min_chimeric_len", "micro_indel_len", "max_splice_junct_len")


def ext_params(g):
    return dict(zip(KEYS, (int(x) for x in g["params"])))


def check_ext(res, exp):
    for k in ("rslt", "inst", "low", "nxt", "hits", "seg2"):
        d = res[k] != exp[k]
        if d.ndim > 1:
            d = d.any(axis=1)
        assert not d.any(), (k, np.nonzero(d)[0][:8], res[k][d][:3], exp[k][d][:3])


def test_cases_present():
    assert len(CASES) >= 13


@pytest.mark.parametrize("case", CASES)
def test_oracle_matches_reference_golden_ext(oracle, golden_dir, g3_path, g3_el5_path, case):
    g = np.load(os.path.join(golden_dir, "align_ext_%s.npz" % case))
    h = oracle.open(g3_el5_path if case.endswith("_el5") else g3_path)
    oracle.set_max_iter(h, 5000)
    check_ext(oracle.align_reads_ext_batch(h, (g["reads"], g["offs"], g["lens"]), **ext_params(g)), g)
    oracle.close(h)


def test_golden_vectors_cover_every_kind(golden_dir):
    """the vectors hold what they are meant to pin: trimmed chimeric hits on both strands, insertions and deletions,
    junctions with and without canonical splice sites (the +50 / +25 score), multi-instance chimeric results"""
    g = np.load(os.path.join(golden_dir, "align_ext_all_100.npz"))
    h0 = g["hits"][:, 0]
    acc = g["rslt"] == 1
    fl = h0["reserved"]
    tl, tr = ext_trims(h0)
    chim = acc & ((fl & EXT_CHIMERIC) != 0)
    assert (chim & (h0["strand"] == ord("+")) & (tl > 0)).sum() > 20 and (chim & (h0["strand"] == ord("-")) & (tr > 0)).sum() > 20
    ins = acc & ((fl & EXT_INDEL) != 0) & ((fl & EXT_INSERT) != 0)
    dele = acc & ((fl & EXT_INDEL) != 0) & ((fl & EXT_INSERT) == 0)
    assert ins.sum() > 10 and dele.sum() > 10
    spl = acc & ((fl & EXT_SPLICE) != 0)
    sc = g["seg2"]["score"][spl]
    assert spl.sum() > 50 and len(set(sc.tolist())) > 3
    assert (g["seg2"]["match_len"][spl] + h0["match_len"][spl] == 100).all()
    g5 = np.load(os.path.join(golden_dir, "align_ext_chim50_mh5.npz"))
    assert ((g5["rslt"] == 1) & (g5["inst"] > 1)).sum() > 5


def test_adaptive_trim_known_answers(oracle):
    """CSfxArray::AdaptiveTrim (SfxArray.cpp:5561-5795) on hand-made alignments"""
    rng = np.random.default_rng(1)
    t = rng.integers(0, 4, 100).astype(np.uint8)
    p = t.copy()
    assert oracle.adaptive_trim(p, t, 100, 2) == (100, 100, 0, 0, 0)            # full length, no mismatch
    p[50] = (p[50] + 1) % 4
    assert oracle.adaptive_trim(p, t, 100, 2) == (100, 100, 0, 0, 1)
    p[1] = (p[1] + 1) % 4                                                       # a mismatch inside the 3-base flank
    assert oracle.adaptive_trim(p, t, 100, 2)[0] == 0
    p = t.copy()
    p[:20] = (p[:20] + 1) % 4                                                   # a foreign 5' flank of 20
    assert oracle.adaptive_trim(p, t, 50, 2) == (80, 80, 20, 0, 0)
    p[-10:] = (p[-10:] + 2) % 4                                                 # and a 3' one of 10
    assert oracle.adaptive_trim(p, t, 50, 2) == (70, 70, 20, 10, 0)
    assert oracle.adaptive_trim(p, t, 75, 2)[0] == 0                            # too little left
    assert oracle.adaptive_trim(p[:20], t[:20], 15, 2)[0] == -100               # shorter than cMinATSeqLen: eBSFerrParams


@pytest.mark.skipif(not ref_available(), reason="oracle/_ref/libk4ref.so not built (needs /root/reference)")
def test_oracle_matches_live_reference_ext(oracle, tmp_path):
    R = Ref()
    names, chroms = synth.make_genome([50000, 35000], seed=321, repeats=20, repeat_len=300, repeat_div=0.03, n_runs=3)
    sites = synth.plant_splice_sites(chroms, 40, seed=322)
    path = str(tmp_path / "x.sfx")
    R.build_sfx(path, names, chroms)
    hr = R.open(path, 5000, 0)
    h = oracle.open(path)
    oracle.set_max_iter(h, 5000)
    rng = np.random.default_rng(9)
    for rl, kw in ((100, dict(tot_mm=2, core_len=33, core_delta=33, max_slides=8, min_core_len=8)),
                   (121, dict(tot_mm=4, core_len=24, core_delta=24, max_slides=10, min_core_len=8, mm_delta=2))):
        reads = synth.make_reads(chroms, 150, rl, seed=5)[0]
        for kind in ("chimeric", "indel", "splice"):
            reads += synth.make_ext_reads(chroms, 150, rl, kind, seed=int(rng.integers(1, 1 << 30)), sites=sites, max_subs=3)
        for mh, ext in ((1, dict(min_chimeric_len=55)), (4, dict(min_chimeric_len=30, micro_indel_len=12)),
                        (1, dict(micro_indel_len=20, max_splice_junct_len=2500)),
                        (2, dict(min_chimeric_len=45, micro_indel_len=3, max_splice_junct_len=800))):
            a = oracle.align_reads_ext_batch(h, reads, max_hits=mh, **kw, **ext)
            b = R.align_reads_ext_batch(hr, reads, max_hits=mh, **kw, **ext)
            check_ext(a, b)
    # AdaptiveTrim on random alignments
    for _ in range(1500):
        L = int(rng.integers(25, 260))
        t = rng.integers(0, 4, L).astype(np.uint8)
        p = t.copy()
        k = int(rng.integers(0, 14))
        pos = rng.integers(0, L, k)
        p[pos] = (p[pos] + rng.integers(1, 4, k)) % 4
        for side in (0, 1):
            if rng.random() < 0.35:
                f = int(rng.integers(1, L // 3 + 1))
                if side:
                    p[-f:] = rng.integers(0, 4, f)
                else:
                    p[:f] = rng.integers(0, 4, f)
        args = (int(rng.integers(15, L + 1)), int(rng.integers(0, 9)), int(rng.integers(0, 7)))
        assert oracle.adaptive_trim(p, t, *args) == R.adaptive_trim(hr, p, t, *args)
    R.close(hr)
    oracle.close(h)


def test_post_stages_known_answers(oracle, g3_path, golden_dir):
    """AutoTrimFlanks (KAligner.cpp:1714-1917) and the orphan-junction filters (:2406-2594) on results of the `-a` / `-A`
    vectors: reads whose junction no other read shares (within 3 bp at both ends) lose their alignment"""
    g = np.load(os.path.join(golden_dir, "align_ext_splice5000.npz"))
    h = oracle.open(g3_path)
    oracle.set_max_iter(h, 5000)
    reads = (g["reads"], g["offs"], g["lens"])
    r = oracle.kalign_ext_batch(h, reads, max_subs=2, max_splice_junct_len=5000, min_core_len=8)
    out, hits, seg2 = r["out"].copy(), r["hits"].copy(), r["seg2"].copy()
    spl = (out["nar"] == NAR_ACCEPTED) & ((hits[:, 0]["reserved"] & EXT_SPLICE) != 0)
    assert spl.sum() > 100
    n_elim = oracle.auto_trim_flanks(h, reads, out, hits, seg2, 5)
    assert n_elim == (out["nar"] == NAR_TRIM).sum()
    assert (out["nar"][spl] == NAR_ACCEPTED).all()                      # two-segment hits are never flank-trimmed
    tl, tr = ext_trims(hits[:, 0])
    one = (out["nar"] == NAR_ACCEPTED) & ~spl
    assert (tl[one] + tr[one] < g["lens"][one] // 2 + 1).all() and (tl[one] > 0).sum() > 0 and (tr[one] > 0).sum() > 0
    removed = oracle.remove_orphan_juncts(EXT_SPLICE, out, hits, seg2)
    kept = spl & (out["nar"] == NAR_ACCEPTED)
    assert removed == (out["nar"] == NAR_SPLICEJCTN).sum() == spl.sum() - kept.sum() and 0 < removed < spl.sum()
    # brute force: a kept junction has a partner within 3 bp at both ends among the junction reads (neighbours in the
    # sorted order are the closest candidates)
    st = hits[:, 0]["match_loci"].astype(np.int64) + hits[:, 0]["match_len"] - 1
    en = seg2["match_loci"].astype(np.int64)
    ch = hits[:, 0]["chrom_id"]
    idx = np.nonzero(spl)[0]
    for i in idx[kept[idx]][:50]:
        near = [j for j in idx if j != i and ch[j] == ch[i] and abs(st[j] - st[i]) <= 3 and abs(en[j] - en[i]) <= 3]
        assert near, i
    oracle.close(h)


@pytest.mark.parametrize("window,seed", [(400, 1), (900, 2), (1500, 3), (6000, 4)])
def test_chimeric_mate_rescue_vs_live_reference(oracle, golden_dir, window, seed):
    """CSfxArray::AlignPairedRead with MinChimericLen (SfxArray.cpp:8571-8767): both branches -- the linear scan below 1000 loci and
    the exact-core seeds of IterateExactsRange above -- on mates whose flanks are foreign sequence, placed relative to made-up
    anchors on the golden genome; oracle vs the live reference (oracle/_ref/libk4ref.so; skipped where it is not built)."""
    from oracle_bindings import Ref, ref_available

    if not ref_available():
        pytest.skip("oracle/_ref/libk4ref.so is not built here")
    names, chroms = synth.golden_genome()
    rng = np.random.default_rng(900 + seed)
    R = Ref()
    path = os.path.join(golden_dir, "g1.sfx")
    hr = R.open(path)
    ho = oracle.open(path)
    n_placed = n_chim = 0
    for t in range(140):
        c = int(rng.integers(0, 3))
        g = chroms[c]
        L = int(rng.choice([100, 125, 150]))
        b3 = bool(rng.integers(0, 2))
        anti = bool(rng.integers(0, 2))
        a_len = 100
        a_start = int(rng.integers(window + 300, len(g) - window - 400))
        a_end = a_start + a_len - 1
        # where the mate really lies: inside the insert window on the proper side
        frag = int(rng.integers(L + 20, window))
        m_start = (a_start + frag - L) if b3 else (a_end - frag + 1)
        if m_start < 0 or m_start + L >= len(g):
            continue
        mate = g[m_start:m_start + L].copy()
        kind = rng.random()
        if kind < 0.6:  # foreign flanks: a chimeric placement
            f5 = int(rng.integers(0, L * 35 // 100)) if rng.random() < 0.7 else 0
            f3 = int(rng.integers(0, L * 35 // 100)) if rng.random() < 0.7 else 0
            mate[:f5] = rng.integers(0, 4, f5)
            if f3:
                mate[L - f3:] = rng.integers(0, 4, f3)
        for _ in range(int(rng.integers(0, 4))):
            p = int(rng.integers(0, L))
            mate[p] = (mate[p] + 1 + rng.integers(0, 3)) % 4
        if kind > 0.93:
            mate = rng.integers(0, 4, L).astype(np.uint8)  # nothing to find
        if anti:
            mate = (3 - mate)[::-1].copy()
        mcl = int(rng.choice([50, 60, 75, 99]))
        max_mm = int(rng.choice([2, 3, 5]))
        core_len = max(8, L // (max_mm + 1))
        core_delta = max(L // 10 - 1, core_len)
        args = (b3, anti, c + 1, a_start, a_end, 50, window + L, max_mm, mate, mcl, core_len, core_delta)
        r_ref, h_ref = R.align_paired_read_x(hr, *args)
        r_o, h_o = oracle.align_paired_read_x(ho, *args)
        assert r_ref == r_o, (t, r_ref, r_o, args[:8])
        if r_ref == 1:
            assert h_ref.tobytes() == h_o.tobytes(), (t, h_ref, h_o)
            n_placed += 1
            n_chim += int((int(h_o["reserved"]) >> 24) & 1)
    assert n_placed > 60 and n_chim > 15
    R.close(hr)
    oracle.close(ho)
