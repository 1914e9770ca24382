"""GPU parity tests proper (-m gpu): the HIP path, called through the C ABI, against
 (1) the golden vectors captured from the real reference, (2) the CPU oracle on fresh seeded inputs,
 (3) size-independent properties (embedded truth, .sfx round trip, suffix-array order)."""
import glob
import os

import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = sorted(os.path.basename(p)[6:-4] for p in glob.glob(os.path.join(GOLDEN, "align_*.npz"))
               if not os.path.basename(p).startswith("align_ext_"))  # (the optional-phase vectors: test_*_ext.py)


@pytest.fixture(scope="module")
def k4():
    import kit4b_amd

    kit4b_amd.lib()  # raises if the HIP extension is missing: no fallback
    return kit4b_amd


def check_against(res, exp, max_hits, what=""):
    for k in ("rslt", "inst", "low", "nxt"):
        bad = np.nonzero(res[k] != exp[k])[0]
        assert len(bad) == 0, (what, k, bad[:8], res[k][bad[:8]], exp[k][bad[:8]])
    for i in range(len(res["rslt"])):
        nh = min(int(exp["inst"][i]), max_hits) if exp["rslt"][i] in (1, 2, 3) else 0
        assert np.array_equal(res["hits"][i, :nh], exp["hits"][i, :nh]), (what, i, res["hits"][i, :nh], exp["hits"][i, :nh])
        assert not res["hits"][i, nh:].view(np.uint8).any(), (what, i, "stale hit slots")


@pytest.mark.parametrize("table", ["16-byte entries (default)", "12-byte entries"])
@pytest.mark.parametrize("kmer_k", [0, 5, 11])
@pytest.mark.parametrize("case", CASES)
def test_golden_vectors_from_reference(k4, golden_dir, g1_el5_path, monkeypatch, case, kmer_k, table):
    if kmer_k and case not in ("c2_s2", "c3_pe150_el5", "maxiter3", "short60"):
        pytest.skip("k sweep on a subset")
    if table.startswith("12"):  # the {lb, pos0, sig} form a device short of memory falls back to (k4_index.hip: open_common)
        monkeypatch.setenv("K4_FORCE_KTAB64", "0")
    g = np.load(os.path.join(golden_dir, "align_%s.npz" % case))
    tm, cl, cd, sl, mcl, md, strand, mh, maxiter = [int(x) for x in g["params"]]
    ix = k4.SfxIndex.open(g1_el5_path if case.endswith("_el5") else os.path.join(golden_dir, "g1.sfx"), kmer_k=kmer_k)
    try:
        info = ix.info()
        assert info["sfx_el_size"] == (5 if case.endswith("_el5") else 4) and info["concat_len"] == 125425
        ix.set_max_iter(maxiter)
        res = ix.align_reads_batch((g["reads"], g["offs"], g["lens"]), tm, cl, cd, sl, mcl, md, strand, mh)
        check_against(res, g, mh, case)
        c = ix.counters()
        assert c["n_reads"] == len(g["lens"])
    finally:
        ix.close()


def test_fresh_inputs_vs_oracle_raw_and_kalign(k4, oracle):
    """2 Mbp genome with repeats/N runs, index built by the oracle's suffix sort, 30k reads of mixed lengths."""
    names, chroms = synth.make_genome([900000, 600000, 400000, 99000, 1000], seed=4242, repeats=200, repeat_len=400,
                                      repeat_div=0.005, n_runs=10, n_run_len=50, tandem=20)
    ho = oracle.build(names, chroms, dataset="fresh", threads=8)
    n = oracle.concat_len(ho)
    ents = k4.make_entries(names, [len(c) for c in chroms])
    sa_raw = np.ctypeslib.as_array(
        __import__("ctypes").cast(oracle.L.k4o_sa_bytes(ho), __import__("ctypes").POINTER(__import__("ctypes").c_uint8)),
        shape=(n * 4,))
    ix = k4.SfxIndex.from_host(np.array(oracle.seq(ho)), sa_raw, 4, ents, dataset="fresh")
    try:
        oracle.set_max_iter(ho, 5000)
        ix.set_max_iter(5000)
        assert ix.min_core_len(0) == oracle.min_core_len(ho, 0)
        for (rl, tm, cl, cd, sl, mh, md, strand, nr) in [(100, 2, 33, 33, 8, 1, 1, 0, 12000), (150, 3, 37, 37, 12, 10, 1, 0, 8000),
                                                        (100, 5, 16, 16, 8, 3, 2, 0, 5000), (36, 1, 18, 18, 3, 2, 1, 0, 3000),
                                                        (300, 6, 42, 42, 24, 4, 1, 0, 2000), (600, 6, 85, 85, 48, 2, 1, 0, 300)]:
            reads, _ = synth.make_reads(chroms, nr, rl, seed=rl * 31 + tm, n_prob=0.03, edge_frac=0.05, random_frac=0.03)
            ro = oracle.align_reads_batch(ho, reads, tm, cl, cd, sl, 0, md, strand, mh, threads=8)
            rg = ix.align_reads_batch(reads, tm, cl, cd, sl, 0, md, strand, mh)
            check_against(rg, ro, mh, "raw len %d" % rl)
        # kalign level: mixed read lengths in one batch, SE and PE classification
        reads = []
        for rl, nr in ((100, 6000), (75, 2000), (151, 3000), (250, 800)):
            r, _ = synth.make_reads(chroms, nr, rl, seed=rl + 5, n_prob=0.05, edge_frac=0.05, random_frac=0.03, sub_lambda=1.5)
            reads += r
        for kw in (dict(max_subs=2), dict(max_subs=5, max_ml=10, pe_mode=1), dict(max_subs=3, min_edit_dist=2, pmode=1),
                   dict(max_subs=0), dict(max_subs=2, max_ns=0, strand=1)):
            eo = oracle.kalign_batch(ho, reads, threads=8, **kw)
            eg = ix.kalign_batch(reads, **kw)
            mh = max(1, kw.get("max_ml", 1))
            for f in ("hit_rslt", "inst", "low_mm", "nxt_mm", "nar", "num_hits"):
                bad = np.nonzero(eo["out"][f] != eg["out"][f])[0]
                assert len(bad) == 0, (kw, f, bad[:5], eo["out"][f][bad[:5]], eg["out"][f][bad[:5]])
            for i in range(len(reads)):
                r = eo["out"][i]
                nh = min(int(r["inst"]), mh) if r["hit_rslt"] in (1, 2, 3) else 0
                assert np.array_equal(eo["hits"][i, :nh], eg["hits"][i, :nh]), (kw, i)
            assert (eg["out"]["nar"] == 2).sum() > 0 and (eg["out"]["nar"] == 1).sum() > 0
    finally:
        ix.close()
        oracle.close(ho)


def test_truth_property_iid_genome_gpu_built_index(k4, oracle):
    """Size-independent property: on an i.i.d. genome a read is AA at its truth locus with
    Mismatches == nsubs iff nsubs <= MaxTotMM, else NL.  Index built entirely on the GPU (suffix sort included)."""
    import torch

    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(synth.GENOME_SEED)
    lens = [3000000, 2000000, 1500000, 500000]
    n = sum(lens) + len(lens)
    seq = torch.empty(n, dtype=torch.uint8, device=dev)
    o = 0
    for ln in lens:
        seq[o:o + ln] = torch.randint(0, 4, (ln,), dtype=torch.uint8, device=dev, generator=g)
        seq[o + ln] = 7
        o += ln + 1
    sa = torch.empty(n, dtype=torch.int32, device=dev)
    k4.build_sa_device(n, 4, seq.data_ptr(), sa.data_ptr())
    names = ["chr%d" % (i + 1) for i in range(len(lens))]
    ents = k4.make_entries(names, lens)
    ix = k4.SfxIndex.from_device(n, 4, seq.data_ptr(), sa.data_ptr(), ents, keep=(sa,))
    try:
        # the GPU suffix array equals the oracle's (ties broken by offset on both sides)
        seq_h = seq.cpu().numpy()
        chroms = []
        o = 0
        for ln in lens:
            chroms.append(seq_h[o:o + ln])
            o += ln + 1
        ho = oracle.build(names, chroms, threads=8)
        assert np.array_equal(oracle.sa(ho), sa.cpu().numpy().astype(np.int64) & 0xFFFFFFFF)
        reads, truth = synth.make_reads(chroms, 40000, 100, seed=77, sub_lambda=1.0)
        res = ix.kalign_batch(reads, max_subs=2)
        out, hits = res["out"], res["hits"][:, 0]
        ok = truth[:, 3] <= 2
        assert (out["nar"][ok] == 1).all() and (out["nar"][~ok] == 3).all()
        assert np.array_equal(hits["chrom_id"][ok], truth[ok, 0]) and np.array_equal(hits["match_loci"][ok], truth[ok, 1])
        assert np.array_equal(hits["mismatches"][ok], truth[ok, 3])
        assert np.array_equal(hits["strand"][ok] == ord("-"), truth[ok, 2] == 1)
        # and read-for-read identical to the oracle
        eo = oracle.kalign_batch(ho, reads, max_subs=2, threads=8)
        assert np.array_equal(eo["out"], out) and np.array_equal(eo["hits"][:, 0], hits)
        c = ix.counters()
        assert c["n_slow"] < 40 and c["n_lookup"] == eo["counters"]["n_lookup"] and c["n_cand"] == eo["counters"]["n_cand"]
        oracle.close(ho)
    finally:
        ix.close()


def test_gpu_suffix_sort_with_repeats_and_n(k4, oracle):
    """Suffix order incl. deep ties (long exact repeats, tandem blocks, N runs, tiny contigs), 4- and 5-byte elements."""
    import torch

    names, chroms = synth.make_genome([200000, 120000, 3000, 40, 7], seed=11, repeats=60, repeat_len=900, repeat_div=0.0,
                                      n_runs=8, n_run_len=200, tandem=15)
    ho = oracle.build(names, chroms, threads=8)
    n = oracle.concat_len(ho)
    seq = torch.from_numpy(np.array(oracle.seq(ho))).cuda()
    for el in (4, 5):
        sa = torch.zeros(n * el + 16, dtype=torch.uint8, device="cuda")
        k4.build_sa_device(n, el, seq.data_ptr(), sa.data_ptr())
        raw = sa.cpu().numpy()[: n * el].reshape(n, el)
        v = raw[:, :4].copy().view("<u4").reshape(n).astype(np.int64)
        if el == 5:
            assert not raw[:, 4].any()
        assert np.array_equal(v, oracle.sa(ho)), el
    oracle.close(ho)


def test_write_sfx_roundtrip(k4, golden_dir, tmp_path):
    """.sfx out == .sfx in for everything the format defines (block + entries byte-identical)."""
    src = os.path.join(golden_dir, "g1.sfx")
    ix = k4.SfxIndex.open(src)
    out = str(tmp_path / "rt.sfx")
    ix.write_sfx(out)
    a, b = open(src, "rb").read(), open(out, "rb").read()
    assert len(a) == len(b) and a[:133] == b[:133] and a[1224:] == b[1224:]
    e = ix.entry(2)
    assert (e["name"], e["seq_len"], e["start_ofs"]) == ("chr2", 40000, 60001)
    assert ix.get_ident("CHR3") == 3
    s = ix.get_seq(2, 100, 50)
    assert bytes(s) == a[1224 + 20 + 60001 + 100: 1224 + 20 + 60001 + 150]
    ix.close()


def test_empty_and_degenerate_batches(k4, golden_dir):
    ix = k4.SfxIndex.open(os.path.join(golden_dir, "g1.sfx"))
    r = ix.align_reads_batch([], 2, 33, 33, 8)
    assert len(r["rslt"]) == 0
    # a read shorter than the core, a read of all N, a 1-base read: the reference semantics are "no cores fit"
    reads = [np.zeros(20, np.uint8), np.full(100, 4, np.uint8), np.array([2], np.uint8)]
    r = ix.align_reads_batch(reads, 2, 33, 33, 8)
    assert list(r["rslt"]) == [0, 0, 0] and list(r["inst"]) == [0, 0, 0]
    with pytest.raises(k4.K4Error):
        ix.align_reads_batch(reads, 2, 0, 33, 8)
    ix.close()


# ---- paired ends -------------------------------------------------------------------------------------------------------
import json  # noqa: E402

import samutil  # noqa: E402
from oracle_bindings import oracle_kalign_pe  # noqa: E402

SAM_CASES = json.load(open(os.path.join(GOLDEN, "sam_cases.json")))
CHROMS = ["chr1", "chr2", "chr3", "chr4", "chr5"]


def _kalign_args(args):
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
        elif a.startswith("-c"): kw["min_chimeric_len"] = int(a[2:])    # chimeric trimming (paired-end cases; the SE ones: kalign_args)
    return kw, pe


def _pick_rand_on_device(ix, out, hits):
    """`-r2`: the draws the reference makes with one thread (C library rand(), never seeded, once per read within the
    instance limit, load order), applied by k4_select_hits_dev."""
    import ctypes

    libc = ctypes.CDLL(None)
    libc.srand(1)
    within = (out["nar"] == 1) & (out["num_hits"] >= 1)
    assert (out["num_hits"][within] > 1).sum() > 5
    choice = np.zeros(len(out), np.uint32)
    for i in np.nonzero(within)[0]:
        choice[i] = libc.rand()  # the entry point reduces it modulo NumHits
    out2, hits2 = ix.select_hits(out, hits, choice)
    assert (out2["num_hits"][within] == 1).all() and (out2["nar"] == out["nar"]).all()
    return out2, hits2


@pytest.mark.parametrize("case", sorted(c for c in SAM_CASES if SAM_CASES[c].get("index") == "g3"))
def test_reference_sam_end_to_end_optional_phases(k4, golden_dir, g3_path, case):
    """`kalign -c / -a / -A / -x`: the records and the NAR tallies the reference wrote, from the GPU's
    AlignRead-level results + the device post stages (flank autotrim, orphan-junction filters)."""
    from test_oracle_sam_golden import check_hist, kalign_args

    kw, post = kalign_args(SAM_CASES[case]["args"])
    ix = k4.SfxIndex.open(g3_path)
    ix.set_max_iter(5000)
    names, reads = samutil.read_fasta_xz(os.path.join(golden_dir, "sam_%s.fa.xz" % case))
    r = ix.kalign_ext_batch(reads, **kw)
    out, hits, _ = ix.post_stages(reads, r["out"], r["hits"], r["seg2"], min_flank_exacts=post.get("min_flank_exacts", 0),
                                  orphan_splice=bool(kw.get("max_splice_junct_len")), orphan_indel=bool(kw.get("micro_indel_len")))
    check_hist(out["nar"], SAM_CASES[case]["nar"])
    res = [dict(nar=int(o["nar"]), hit=hh[0], pe_aligned=0, seg2=s2) for o, hh, s2 in zip(out, hits, r["seg2"])]
    _, recs = samutil.read_sam_xz(os.path.join(golden_dir, "sam_%s.sam.xz" % case))
    assert sorted(samutil.sam_records(names, reads, res, ["chr1", "chr2", "chr3"])) == sorted(recs)
    ix.close()


@pytest.mark.parametrize("case", sorted(c for c in SAM_CASES if SAM_CASES[c].get("index") != "g3"))
def test_reference_sam_end_to_end(k4, golden_dir, g2_path, case):
    """The records `ngskit4b kalign` wrote (SE and PE incl. mate rescue) are reproduced from the GPU results."""
    kw, pe = _kalign_args(SAM_CASES[case]["args"])
    ix = k4.SfxIndex.open(g2_path if SAM_CASES[case].get("index") == "g2" else os.path.join(golden_dir, "g1.sfx"))
    ix.set_max_iter(5000 if kw["pmode"] == 0 else 10000)
    _, recs = samutil.read_sam_xz(os.path.join(golden_dir, "sam_%s.sam.xz" % case))
    if case.startswith("se_"):
        names, reads = samutil.read_fasta_xz(os.path.join(golden_dir, SAM_CASES[case].get("reads", "sam_%s.fa.xz" % case)))
        r = ix.kalign_batch(reads, **kw)
        from test_oracle_sam_golden import expand_all_hits

        clust = [int(a[2:]) for a in SAM_CASES[case]["args"] if a in ("-r3", "-r4")]
        if clust:  # AssignMultiMatches on the device
            multi = ((r["out"]["hit_rslt"] == 1) & (r["out"]["inst"] > 1)).sum()
            r["out"], r["hits"], got = ix.assign_multi(r["out"], r["hits"], clust[0], max(len(x) for x in reads))
            assert multi > 100 and 20 < got < multi and (r["hits"]["reserved"] == 0).all()
            r["out"]["num_hits"][r["out"]["nar"] != 1] = 0
        if "-r2" in SAM_CASES[case]["args"]:
            r["out"], r["hits"] = _pick_rand_on_device(ix, r["out"], r["hits"])
        names, reads, res = expand_all_hits(names, reads, r["out"], r["hits"])
        got = samutil.sam_records(names, reads, res, CHROMS)
        nars = r["out"]["nar"]
    else:
        n1, r1 = samutil.read_fasta_xz(os.path.join(golden_dir, "sam_%s_1.fa.xz" % case))
        n2, r2 = samutil.read_fasta_xz(os.path.join(golden_dir, "sam_%s_2.fa.xz" % case))
        out = ix.kalign_pe_batch(r1, r2, **pe, **kw)
        names = [x for p in zip(n1, n2) for x in p]
        reads = [x for p in zip(r1, r2) for x in p]
        res = [dict(nar=int(o["nar"]), hit=o["hit"], pe_aligned=int(o["pe_aligned"])) for o in out]
        got = samutil.sam_records(names, reads, res, CHROMS, paired=True)
        nars = out["nar"]
    assert sorted(got) == sorted(recs)
    hist = np.bincount(nars, minlength=32)
    if "-r5" in SAM_CASES[case]["args"]:  # the reference tallies reported loci, not reads, in this mode
        assert SAM_CASES[case]["nar"]["AA"] == len(recs)
        ix.close()
        return
    for name, code in {"AA": 1, "EN": 2, "NL": 3, "MH": 4, "ML": 5, "UI": 13, "OI": 14, "UP": 15, "IS": 16, "IT": 17}.items():
        assert hist[code] == SAM_CASES[case]["nar"].get(name, 0), (name, hist[code])
    ix.close()


@pytest.mark.parametrize("lo,hi", [(100, 1500), (150, 5000)])
def test_mate_rescue_over_windows_of_1000_loci_and_more(k4, oracle, golden_dir, lo, hi):
    """Insert windows of 1000 loci or more: the reference itself has no behaviour there (its zero-length-seed loop walks off the
    suffix array: tests/test_oracle_sam_golden.py::test_reference_dies_on_wide_rescue_window); device and oracle apply the
    linear-scan rule (SfxArray.cpp:8731-8766) to the whole window -- packed XOR/popcount scan vs the symbol loop."""
    names, chroms = synth.golden_genome()
    pe1, pe2, _ = synth.make_pe_reads(chroms, 2500, 125, seed=77 + hi, sub_lambda=2.2, n_prob=0.04, random_mate_frac=0.05,
                                      frag_min=260, frag_max=min(hi, 1400))
    ix = k4.SfxIndex.open(os.path.join(golden_dir, "g1.sfx"))
    ho = oracle.open(os.path.join(golden_dir, "g1.sfx"))
    ix.set_max_iter(5000)
    oracle.set_max_iter(ho, 5000)
    kw = dict(pe_mode=1, pair_min_len=lo, pair_max_len=hi, pair_strand=False, max_subs=3)
    g = ix.kalign_pe_batch(pe1, pe2, **kw)
    o = oracle_kalign_pe(oracle, ho, pe1, pe2, threads=8, **kw)
    for f in ("nar", "num_hits", "inst", "low_mm", "pe_aligned", "rescued"):
        bad = np.nonzero(g[f] != o[f])[0]
        assert len(bad) == 0, (f, bad[:6], g[f][bad[:6]], o[f][bad[:6]])
    acc = g["nar"] == 1
    assert np.array_equal(g["hit"][acc], o["hit"][acc])
    assert g["rescued"].sum() > 20
    ix.close()
    oracle.close(ho)


@pytest.mark.parametrize("pe_mode,lo,hi,mcl", [(1, 200, 600, 50), (3, 150, 1400, 60), (1, 100, 2500, 75), (2, 200, 700, 50)])
def test_pe_chimeric_flow_vs_oracle(k4, oracle, golden_dir, pe_mode, lo, hi, mcl):
    """kalign -c with paired ends: AlignReads' chimeric pass for both ends, trimmed loci in AcceptProvPE / PEInsertSize, and
    AlignPairedRead's AdaptiveTrim branch for the rescue -- scanned windows (< 1000 loci) and windows seeded with exact cores
    (IterateExactsRange); device vs the oracle (itself pinned by the live reference and two reference SAMs)."""
    names, chroms = synth.golden_genome()
    pe1, pe2, _ = synth.make_pe_reads(chroms, 2200, 125, seed=640 + hi, sub_lambda=1.6, n_prob=0.02, random_mate_frac=0.04,
                                      frag_min=max(lo + 20, 260), frag_max=min(hi - 50, 1300))
    rng = np.random.default_rng(hi)
    for rd in pe1 + pe2:  # a third of the reads get foreign flanks
        if rng.random() < 0.33:
            L = len(rd)
            for side in (0, 1):
                if rng.random() < 0.65:
                    k = int(rng.integers(L * 5 // 100, L * 35 // 100))
                    if side == 0:
                        rd[:k] = rng.integers(0, 4, k)
                    else:
                        rd[L - k:] = rng.integers(0, 4, k)
    ix = k4.SfxIndex.open(os.path.join(golden_dir, "g1.sfx"))
    ho = oracle.open(os.path.join(golden_dir, "g1.sfx"))
    ix.set_max_iter(5000)
    oracle.set_max_iter(ho, 5000)
    kw = dict(pe_mode=pe_mode, pair_min_len=lo, pair_max_len=hi, pair_strand=False, max_subs=3, min_chimeric_len=mcl)
    g = ix.kalign_pe_batch(pe1, pe2, **kw)
    o = oracle_kalign_pe(oracle, ho, pe1, pe2, threads=8, **kw)
    for f in ("nar", "num_hits", "inst", "low_mm", "pe_aligned", "rescued"):
        bad = np.nonzero(g[f] != o[f])[0]
        assert len(bad) == 0, (f, bad[:6], g[f][bad[:6]], o[f][bad[:6]])
    acc = g["nar"] == 1
    assert np.array_equal(g["hit"][acc], o["hit"][acc])
    chim = (g["hit"]["reserved"][acc] >> 24) & 1
    assert chim.sum() > 100  # trimmed placements are reported
    if pe_mode in (1, 3):
        assert (g["rescued"][acc] & (chim > 0)).sum() > 5  # ... also from the rescue
    ix.close()
    oracle.close(ho)


@pytest.mark.parametrize("pe_mode,pair_strand", [(1, False), (2, False), (3, False), (4, False), (1, True)])
def test_pe_flow_vs_oracle(k4, oracle, golden_dir, pe_mode, pair_strand):
    names, chroms = synth.golden_genome()
    pe1, pe2, _ = synth.make_pe_reads(chroms, 3000, 125, seed=500 + pe_mode, sub_lambda=1.8, n_prob=0.03,
                                      random_mate_frac=0.05, frag_min=260, frag_max=700)
    ix = k4.SfxIndex.open(os.path.join(golden_dir, "g1.sfx"))
    ho = oracle.open(os.path.join(golden_dir, "g1.sfx"))
    ix.set_max_iter(5000)
    oracle.set_max_iter(ho, 5000)
    kw = dict(pe_mode=pe_mode, pair_min_len=220, pair_max_len=640, pair_strand=pair_strand, max_subs=2)
    g = ix.kalign_pe_batch(pe1, pe2, **kw)
    o = oracle_kalign_pe(oracle, ho, pe1, pe2, threads=8, **kw)
    for f in ("nar", "num_hits", "inst", "low_mm", "pe_aligned", "rescued"):
        bad = np.nonzero(g[f] != o[f])[0]
        assert len(bad) == 0, (f, bad[:6], g[f][bad[:6]], o[f][bad[:6]])
    acc = g["nar"] == 1
    assert np.array_equal(g["hit"][acc], o["hit"][acc])
    if pe_mode in (1, 3) and not pair_strand:
        assert g["rescued"].sum() > 0
    # the device-resident entry point on the same pairs (reads interleaved in HBM) returns the same records
    import torch

    inter = [x for pair in zip(pe1, pe2) for x in pair]
    lens = np.array([len(x) for x in inter], dtype=np.uint32)
    offs = np.concatenate([[0], np.cumsum(lens[:-1], dtype=np.uint64)]).astype(np.uint64)
    cat = np.concatenate(inter + [np.zeros(16, np.uint8)])
    dev = torch.device("cuda:0")
    d_reads, d_offs = torch.from_numpy(cat).to(dev), torch.from_numpy(offs.view(np.int64)).to(dev)
    d_lens = torch.from_numpy(lens.view(np.int32)).to(dev)
    d_out = torch.zeros(len(inter) * k4.PE_READ_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    kp = k4.KalignParams(2, 1, 1, 0, k4.STRAND_BOTH, 10, 1, 0, 0)
    pp = k4.PeParams(pe_mode, 220, 640, 1 if pair_strand else 0)
    ix.kalign_pe_batch_dev(kp, pp, len(pe1), int(lens.max()), d_reads.data_ptr(), d_offs.data_ptr(), d_lens.data_ptr(),
                           d_out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    gd = d_out.cpu().numpy().view(k4.PE_READ_DTYPE)
    assert np.array_equal(gd, g)
    ix.close()
    oracle.close(ho)


# ---- host programs: k4align (SAM out) and the CSfxArray facade ------------------------------------------------------
import lzma  # noqa: E402
import subprocess  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _unxz(src, dst):
    with lzma.open(src, "rb") as f, open(dst, "wb") as g:
        g.write(f.read())
    return dst


@pytest.mark.parametrize("case", sorted(SAM_CASES))
def test_k4align_writes_the_reference_sam(k4, golden_dir, g2_path, g3_path, tmp_path, case):
    """`k4align` (C++ over the C ABI) against the SAM `ngskit4b kalign` wrote for the same reads, index and options:
    identical header (but @PG) and identical records (the reference's order among equal keys is unspecified)."""
    exe = os.path.join(ROOT, "kit4b_amd", "k4align")
    assert os.path.exists(exe)
    out = str(tmp_path / "out.sam")
    sfx = {"g2": g2_path, "g3": g3_path}.get(SAM_CASES[case].get("index"), os.path.join(golden_dir, "g1.sfx"))
    cmd = [exe, "-I", sfx, "-o", out] + SAM_CASES[case]["args"]
    if case.startswith("se_"):
        cmd += ["-i", _unxz(os.path.join(golden_dir, SAM_CASES[case].get("reads", "sam_%s.fa.xz" % case)), str(tmp_path / "r.fa"))]
    else:
        cmd += ["-i", _unxz(os.path.join(golden_dir, "sam_%s_1.fa.xz" % case), str(tmp_path / "r1.fa")),
                "-u", _unxz(os.path.join(golden_dir, "sam_%s_2.fa.xz" % case), str(tmp_path / "r2.fa"))]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    hdr, recs = samutil.read_sam_xz(os.path.join(golden_dir, "sam_%s.sam.xz" % case))
    got = open(out).read().splitlines()
    got_hdr = [l for l in got if l.startswith("@")]
    got_recs = [l for l in got if not l.startswith("@")]
    assert [l for l in got_hdr if not l.startswith("@PG")] == [l for l in hdr if not l.startswith("@PG")]
    assert sorted(got_recs) == sorted(recs)
    # coordinate order: (RNAME in header order, POS) non-decreasing
    order = {l.split("\t")[2][3:]: i for i, l in enumerate(h for h in hdr if h.startswith("@SQ"))}
    keys = [(order[l.split("\t")[2]], int(l.split("\t")[3])) for l in got_recs]
    assert keys == sorted(keys)
    if "-r5" in SAM_CASES[case]["args"]:  # the reference tallies reported loci, not reads, in this mode
        assert ("%d alignments written" % SAM_CASES[case]["nar"]["AA"]) in p.stderr
        return
    for name, n in SAM_CASES[case]["nar"].items():
        assert ("%d (%s)" % (n, name)) in p.stderr, (name, n)


def test_csfxarray_facade_program(k4, golden_dir):
    """include/k4_sfxarray.hpp used the way CKAligner uses CSfxArray; known answers of SURVEY App. C."""
    exe = os.path.join(ROOT, "kit4b_amd", "k4_facade_test")
    p = subprocess.run([exe, os.path.join(golden_dir, "g1.sfx")], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    lines = p.stdout.splitlines()
    assert lines[0] == "entries 5 totlen 125420 dataset g1"
    assert "entry 2 chr2 40000 ident 2" in lines
    expect = {0: "rslt 1 inst 1 low 0 nxt 2", 1: "rslt 1 inst 1 low 1 nxt 3", 2: "rslt 1 inst 1 low 2 nxt 4",
              3: "rslt 0 inst 0 low 4 nxt 4"}
    n = 0
    for l in lines:
        if l.startswith("probe"):
            f = l.split()
            loci, subs = int(f[1]), int(f[3])
            assert expect[subs] in l, l
            if subs <= 2:
                assert ("chrom 2 loci %d strand + mm %d len 100" % (loci, subs)) in l, l
            n += 1
    assert n == 16
    chim = [l for l in lines if l.startswith("chimeric rslt")][0].split()
    assert chim[2] == "1" and chim[4] == "1" and chim[6] == "1" and chim[8] == "5000" and chim[12] == "0" and chim[14] == "0"
    assert 28 <= int(chim[10]) <= 33  # TrimLeft: the foreign flank (a chance match may shorten / the 3-base rule lengthen it)
    assert "carried-in rslt -3 msgs 1" in lines
    assert "best rslt 1 inst 1 chrom 2 loci 1200 strand + mm 1" in lines      # LocateBestMatches, same signature
    assert "pair rslt 1 chrom 2 loci 1200 strand - mm 0" in lines              # AlignPairedRead, same signature
    th = [l for l in lines if l.startswith("threads 8 probes 480 hits ")]
    assert th and th[0].endswith("identical 1") and int(th[0].split()[5]) > 100  # 8 threads on one object == 1 thread
    assert any(l.startswith("header sfx version ") and l.endswith(" blocks 1 dataset g1") for l in lines)  # GetSfxHeader
    assert lines[-1] == "flags 0 prev 0 now 1 solid 0"


def test_best_matches_raw_call_vs_oracle(k4, oracle, golden_dir):
    """k4_best_matches_batch (CSfxArray::LocateBestMatches) against the oracle's restatement, which the reference's own
    `-N` SAM pins (tests/golden/sam_se_r5_R8_N.*): return value, instance count and the sorted hit list."""
    import ctypes as C

    names, chroms = synth.golden_genome()
    reads, _ = synth.make_reads(chroms, 2500, 100, seed=8181, sub_lambda=1.6, n_prob=0.02, edge_frac=0.05)
    ix = k4.SfxIndex.open(os.path.join(golden_dir, "g1.sfx"))
    ho = oracle.open(os.path.join(golden_dir, "g1.sfx"))
    for max_iter, max_hits, tot_mm in ((5000, 6, 3), (40, 3, 2), (5000, 1, 0)):
        ix.set_max_iter(max_iter)
        oracle.set_max_iter(ho, max_iter)
        g = ix.best_matches_batch(reads, tot_mm, 25, 25, 8, max_hits=max_hits)
        L = oracle.L
        L.k4o_locate_best_matches.argtypes = [C.c_void_p] + [C.c_int] * 5 + [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int),
                                                                             C.c_void_p, C.c_int, C.c_void_p]
        for i, rd in enumerate(reads):
            buf = np.ascontiguousarray(rd, dtype=np.uint8).copy()
            hits = np.zeros(max_hits + 1, dtype=g["hits"].dtype)
            inst = C.c_int(0)
            r = L.k4o_locate_best_matches(ho, tot_mm, 25, 25, 8, 0, buf.ctypes.data, len(buf), max_hits, C.byref(inst),
                                          hits.ctypes.data, max_iter, None)
            assert (r, inst.value) == (int(g["rslt"][i]), int(g["inst"][i])), (i, r, inst.value, g["rslt"][i], g["inst"][i])
            assert np.array_equal(hits[: inst.value], g["hits"][i][: inst.value]), i
            assert not g["hits"][i][inst.value:].view(np.uint8).any()
        if max_hits > 1:
            assert (g["inst"] > 1).sum() > 3
        else:
            assert (g["rslt"] == 2).sum() > 3  # one slot only: further matches were sloughed
    ix.close()
    oracle.close(ho)


def test_cores_shorter_than_the_kmer_table(k4, oracle, golden_dir):
    """Cores shorter than k use a prefix range of the table (several buckets, the first possibly empty)."""
    names, chroms = synth.golden_genome()
    ho = oracle.open(os.path.join(golden_dir, "g1.sfx"))
    oracle.set_max_iter(ho, 5000)
    reads, _ = synth.make_reads(chroms[:3], 1500, 40, seed=3, sub_lambda=1.0, edge_frac=0.1)
    r15, _ = synth.make_reads(chroms[:3], 300, 15, seed=4, sub_lambda=0.3)
    for kmer_k in (12, 14):
        ix = k4.SfxIndex.open(os.path.join(golden_dir, "g1.sfx"), kmer_k=kmer_k)
        ix.set_max_iter(5000)
        for rd, (tm, cl, cd, sl, mh) in ((reads, (3, 8, 8, 4, 2)), (reads, (2, 10, 5, 6, 1)), (r15, (1, 6, 6, 2, 3))):
            ro = oracle.align_reads_batch(ho, rd, tm, cl, cd, sl, 0, 1, 0, mh, threads=8)
            rg = ix.align_reads_batch(rd, tm, cl, cd, sl, 0, 1, 0, mh)
            check_against(rg, ro, mh, "k=%d cl=%d" % (kmer_k, cl))
        eo = oracle.kalign_batch(ho, r15, max_subs=2, threads=8)
        eg = ix.kalign_batch(r15, max_subs=2)
        assert np.array_equal(eo["out"], eg["out"])
        ix.close()
    oracle.close(ho)


def test_deep_repeats_vs_oracle(k4, oracle):
    """Thousands of candidates per core (long tandem arrays): exercises MaxIter, the dedupe set beyond the fast path's
    list, the small-table overflow of the general kernel and its big-table second pass."""
    rng = np.random.default_rng(77)
    c1 = rng.integers(0, 4, size=120000, dtype=np.uint8)
    c1[20000:27000] = np.resize(np.array([0, 2], np.uint8), 7000)           # (AG)n, 7 kb
    c1[60000:66000] = np.resize(np.array([1, 1, 3, 0, 2], np.uint8), 6000)  # 5-mer array, 6 kb
    c1[90000:94000] = 0                                                      # poly-A, 4 kb
    c2 = rng.integers(0, 4, size=50000, dtype=np.uint8)
    c2[10000:13000] = np.resize(np.array([0, 2], np.uint8), 3000)
    names, chroms = ["chr1", "chr2"], [c1, c2]
    ho = oracle.build(names, chroms, threads=8)
    n = oracle.concat_len(ho)
    import ctypes as C

    sa_raw = np.ctypeslib.as_array(C.cast(oracle.L.k4o_sa_bytes(ho), C.POINTER(C.c_uint8)), shape=(n * 4,))
    ix = k4.SfxIndex.from_host(np.array(oracle.seq(ho)), sa_raw, 4, k4.make_entries(names, [len(c) for c in chroms]))
    try:
        reads = []
        for start in list(range(19950, 27050, 37)) + list(range(59960, 66040, 41)) + list(range(89950, 94050, 53)):
            r = c1[start:start + 100].copy()
            if start % 3 == 0:
                r[50] = (r[50] + 1) % 4
            reads.append(r if start % 2 else synth.revcomp(r))
        r2, _ = synth.make_reads(chroms, 300, 100, seed=5)
        reads += r2
        for max_iter in (5000, 50, 20000):
            oracle.set_max_iter(ho, max_iter)
            ix.set_max_iter(max_iter)
            for (tm, cl, cd, sl, mh) in [(2, 33, 33, 8, 1), (3, 25, 25, 8, 10)]:
                ro = oracle.align_reads_batch(ho, reads, tm, cl, cd, sl, 0, 1, 0, mh, threads=8)
                rg = ix.align_reads_batch(reads, tm, cl, cd, sl, 0, 1, 0, mh)
                check_against(rg, ro, mh, "deep repeats maxiter %d" % max_iter)
            assert ro["inst"].max() > 1000 or max_iter == 50
    finally:
        ix.close()
        oracle.close(ho)


def test_long_and_ragged_reads_vs_oracle(k4, oracle, golden_dir):
    """Read lengths from 15 to 2000 bp in one batch: the 4/5/8/16-chunk fast kernels, the > 512 bp general path and
    reads too short for any core, at both the AlignReads and the AlignRead level."""
    names, chroms = synth.golden_genome()
    ix = k4.SfxIndex.open(os.path.join(golden_dir, "g1.sfx"))
    ho = oracle.open(os.path.join(golden_dir, "g1.sfx"))
    ix.set_max_iter(5000)
    oracle.set_max_iter(ho, 5000)
    reads = []
    for rl, nr in ((15, 20), (24, 30), (49, 50), (129, 200), (161, 200), (257, 150), (400, 100), (513, 60), (900, 40),
                   (2000, 20)):
        r, _ = synth.make_reads(chroms[:3], nr, rl, seed=rl, sub_lambda=max(1.0, rl / 80.0), n_prob=0.05, edge_frac=0.1)
        reads += r
    for kw in (dict(max_subs=2), dict(max_subs=5, max_ml=10, pe_mode=1)):
        eo = oracle.kalign_batch(ho, reads, threads=8, **kw)
        eg = ix.kalign_batch(reads, **kw)
        assert np.array_equal(eo["out"], eg["out"]), np.nonzero(eo["out"] != eg["out"])[0][:5]
        mh = kw.get("max_ml", 1)
        for i in range(len(reads)):
            r = eo["out"][i]
            nh = min(int(r["inst"]), mh) if r["hit_rslt"] in (1, 2, 3) else 0
            assert np.array_equal(eo["hits"][i, :nh], eg["hits"][i, :nh]), i
    ro = oracle.align_reads_batch(ho, reads, 4, 20, 20, 30, 0, 1, 0, 3, threads=8)
    rg = ix.align_reads_batch(reads, 4, 20, 20, 30, 0, 1, 0, 3)
    check_against(rg, ro, 3, "ragged raw")
    assert ix.counters()["n_slow"] > 0
    ix.close()
    oracle.close(ho)


def test_index_above_4gbp_5byte_elements(k4, oracle):
    """>= 2^32 symbols: 5-byte suffix elements (SfxArray.h:184), 64-bit table fields, 64-bit suffix sort.  4.5 Gbp built
    on the GPU; truth property on 1 M reads and read-for-read equality with the oracle on a sample."""
    import torch

    free, _ = torch.cuda.mem_get_info()
    if free < 215 * (1 << 30):
        pytest.skip("needs ~210 GB of free HBM")
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(synth.GENOME_SEED + 5)
    n_chrom, chrom_len = 36, 125_000_000
    n = n_chrom * (chrom_len + 1)
    assert n > (1 << 32)
    seq = torch.empty(n, dtype=torch.uint8, device=dev)
    for c in range(n_chrom):
        o = c * (chrom_len + 1)
        seq[o:o + chrom_len] = torch.randint(0, 4, (chrom_len,), dtype=torch.uint8, device=dev, generator=g)
        seq[o + chrom_len] = 7
    sa = torch.empty(n * 5 + 16, dtype=torch.uint8, device=dev)
    k4.build_sa_device(n, 5, seq.data_ptr(), sa.data_ptr())
    names = ["chr%d" % (i + 1) for i in range(n_chrom)]
    ix = k4.SfxIndex.from_device(n, 5, seq.data_ptr(), sa.data_ptr(), k4.make_entries(names, [chrom_len] * n_chrom),
                                 keep=(sa,))
    try:
        info = ix.info()
        assert info["sfx_el_size"] == 5 and info["kmer_k"] == 16
        ix.set_max_iter(5000)
        # reads from the far end of the concatenation (offsets above 2^32) and from the start
        rng = np.random.default_rng(9)
        nr, L = 200000, 100
        chrom = np.concatenate([rng.integers(33, 36, nr // 2), rng.integers(0, 3, nr // 2)])
        start = rng.integers(0, chrom_len - L, nr)
        gofs = torch.from_numpy(chrom * (chrom_len + 1) + start).to(dev)
        rd = seq[gofs[:, None] + torch.arange(L, device=dev)[None, :]].cpu().numpy()
        nsubs = np.minimum(rng.poisson(1.0, nr), 8)
        strand = rng.integers(0, 2, nr)
        reads = []
        for i in range(nr):
            r = rd[i].copy()
            for p_ in rng.choice(L, size=nsubs[i], replace=False):
                r[p_] = (r[p_] + rng.integers(1, 4)) % 4
            reads.append(synth.revcomp(r) if strand[i] else r)
        res = ix.kalign_batch(reads, max_subs=2)
        out, hits = res["out"], res["hits"][:, 0]
        ok = nsubs <= 2
        assert (out["nar"][ok] == 1).all() and (out["nar"][~ok] == 3).all()
        assert np.array_equal(hits["chrom_id"][ok], chrom[ok] + 1) and np.array_equal(hits["match_loci"][ok], start[ok])
        assert np.array_equal(hits["mismatches"][ok], nsubs[ok])
        # oracle on the same 27 GB index (host copy), a 20 k sample
        seq_h = seq.cpu().numpy()
        sa_h = sa[: n * 5].cpu().numpy()
        from oracle_bindings import Entry as OEntry

        oe = (OEntry * n_chrom)()
        for i in range(n_chrom):
            oe[i].entry_id = i + 1; oe[i].fblock_id = 1; oe[i].name = names[i].encode(); oe[i].seq_len = chrom_len
            oe[i].start_ofs = i * (chrom_len + 1); oe[i].end_ofs = i * (chrom_len + 1) + chrom_len - 1
        ho = oracle.L.k4o_from_parts(n, 5, seq_h.ctypes.data, sa_h.ctypes.data, n_chrom, oe, b"big")
        oracle.set_max_iter(ho, 5000)
        sample = list(range(0, 10000)) + list(range(nr // 2, nr // 2 + 10000))
        eo = oracle.kalign_batch(ho, [reads[i] for i in sample], max_subs=2, threads=16)
        assert np.array_equal(eo["out"], out[sample]) and np.array_equal(eo["hits"][:, 0], hits[sample])
        oracle.close(ho)
    finally:
        ix.close()


@pytest.mark.gpu
def test_c5_scale_15gbp_se_and_pe_truth(k4):
    """BASELINE config C5 at full index size on one GPU: 120 x 125 Mbp = 15 Gbp built on the device (5-byte elements,
    bucketed suffix sort, 64-bit table), 150 bp reads, `-s3` (MaxTotMM 5), PE `-U2 -d200 -D600`.  The oracle cannot run
    at this size; the check is the analytic truth property of SURVEY.md 8(d): a read is accepted, at its true locus and
    with Mismatches == induced substitutions, iff it carries <= MaxTotMM substitutions, else it is NL."""
    import torch

    free, _ = torch.cuda.mem_get_info()
    if free < 240 * (1 << 30):
        pytest.skip("needs ~230 GB of free HBM")
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(synth.GENOME_SEED + 15)
    n_chrom, chrom_len = 120, 125_000_000
    n = n_chrom * (chrom_len + 1)
    seq = torch.empty(n, dtype=torch.uint8, device=dev)
    for c in range(n_chrom):
        o = c * (chrom_len + 1)
        seq[o:o + chrom_len] = torch.randint(0, 4, (chrom_len,), dtype=torch.uint8, device=dev, generator=g)
        seq[o + chrom_len] = 7
    sa = torch.empty(n * 5 + 16, dtype=torch.uint8, device=dev)
    k4.build_sa_device(n, 5, seq.data_ptr(), sa.data_ptr())
    names = ["chr%d" % (i + 1) for i in range(n_chrom)]
    ix = k4.SfxIndex.from_device(n, 5, seq.data_ptr(), sa.data_ptr(), k4.make_entries(names, [chrom_len] * n_chrom),
                                 keep=(sa,))
    try:
        ix.set_max_iter(5000)
        rng = np.random.default_rng(15)
        L, nfrag = 150, 60000
        chrom = rng.integers(0, n_chrom, nfrag)
        flen = rng.integers(300, 501, nfrag)
        fstart = rng.integers(0, chrom_len - 500, nfrag)
        fstrand = rng.integers(0, 2, nfrag)
        base = torch.from_numpy(chrom * (chrom_len + 1) + fstart).to(dev)
        left = seq[base[:, None] + torch.arange(L, device=dev)[None, :]].cpu().numpy()
        rofs = torch.from_numpy(chrom * (chrom_len + 1) + fstart + flen - L).to(dev)
        right = seq[rofs[:, None] + torch.arange(L, device=dev)[None, :]].cpu().numpy()

        def mutate(r):
            ns = int(min(rng.poisson(2.0), 9))
            for p_ in rng.choice(L, size=ns, replace=False):
                r[p_] = (r[p_] + rng.integers(1, 4)) % 4
            return r, ns

        pe1, pe2, ns1, ns2 = [], [], np.zeros(nfrag, int), np.zeros(nfrag, int)
        loci1, loci2 = np.zeros(nfrag, np.int64), np.zeros(nfrag, np.int64)
        for i in range(nfrag):
            if fstrand[i] == 0:  # PE1 = fragment 5' end, PE2 = revcomp of its 3' end
                a, b = left[i].copy(), synth.revcomp(right[i])
                loci1[i], loci2[i] = fstart[i], fstart[i] + flen[i] - L
            else:
                a, b = synth.revcomp(right[i]), left[i].copy()
                loci1[i], loci2[i] = fstart[i] + flen[i] - L, fstart[i]
            a, ns1[i] = mutate(a)
            b, ns2[i] = mutate(b)
            pe1.append(a)
            pe2.append(b)
        # ---- SE: every end on its own ----------------------------------------------------------------------------------
        res = ix.kalign_batch(pe1 + pe2, max_subs=3)
        out, hits = res["out"], res["hits"][:, 0]
        ns = np.concatenate([ns1, ns2])
        ok = ns <= 5
        assert (~ok).sum() > 100
        assert (out["nar"][ok] == 1).all() and (out["nar"][~ok] == 3).all()
        assert np.array_equal(hits["chrom_id"][ok], np.concatenate([chrom, chrom])[ok] + 1)
        assert np.array_equal(hits["match_loci"][ok], np.concatenate([loci1, loci2])[ok])
        assert np.array_equal(hits["mismatches"][ok], ns[ok])
        st = np.concatenate([np.where(fstrand == 0, ord("+"), ord("-")), np.where(fstrand == 0, ord("-"), ord("+"))])
        assert np.array_equal(hits["strand"][ok], st[ok])
        # ---- PE: pairs whose two ends are both within MaxTotMM are accepted as proper pairs at the true loci -----------
        pe = ix.kalign_pe_batch(pe1, pe2, pe_mode=2, pair_min_len=200, pair_max_len=600, max_subs=3)
        both = (ns1 <= 5) & (ns2 <= 5)
        o1, o2 = pe[0::2], pe[1::2]
        assert (o1["nar"][both] == 1).all() and (o2["nar"][both] == 1).all()
        assert (o1["pe_aligned"][both] == 1).all() and (o2["pe_aligned"][both] == 1).all()
        assert np.array_equal(o1["hit"]["match_loci"][both], loci1[both])
        assert np.array_equal(o2["hit"]["match_loci"][both], loci2[both])
        assert np.array_equal(o1["hit"]["chrom_id"][both], chrom[both] + 1)
        neither = (ns1 > 5) & (ns2 > 5)
        assert (o1["nar"][neither] != 1).all() and (o2["nar"][neither] != 1).all()
    finally:
        ix.close()


@pytest.mark.parametrize("layout", ["el5_kt64", "el5_kt32", "el4_kt32"])
@pytest.mark.parametrize("case", ["se_s2", "pe_u1", "se_r5_R8_N", "se_r5_R6_X", "pe_c50_u1", "pe_c60_u3_wide"])
def test_reference_sam_on_5byte_index_with_64bit_table(k4, golden_dir, g1_el5_path, monkeypatch, case, layout):
    """The same golden SAMs through the other index layouts: 5-byte suffix elements with 16-byte table entries (what a
    >= 2^32-symbol block uses) and the 12-byte table entries of a device short of memory, with either element size -- every
    (EL, table) instantiation of the step, general, pairing and rescue kernels (4-byte elements with 16-byte entries is
    the default every other test runs)."""
    monkeypatch.setenv("K4_FORCE_KTAB64", "1" if layout.endswith("kt64") else "0")
    kw, pe = _kalign_args(SAM_CASES[case]["args"])
    ix = k4.SfxIndex.open(g1_el5_path if layout.startswith("el5") else os.path.join(golden_dir, "g1.sfx"))
    assert ix.info()["sfx_el_size"] == (5 if layout.startswith("el5") else 4)
    ix.set_max_iter(5000)
    _, recs = samutil.read_sam_xz(os.path.join(golden_dir, "sam_%s.sam.xz" % case))
    if case.startswith("se_"):
        names, reads = samutil.read_fasta_xz(os.path.join(golden_dir, "sam_%s.fa.xz" % case))
        r = ix.kalign_batch(reads, **kw)
        from test_oracle_sam_golden import expand_all_hits

        names, reads, res = expand_all_hits(names, reads, r["out"], r["hits"])
        got = samutil.sam_records(names, reads, res, CHROMS)
    else:
        n1, r1 = samutil.read_fasta_xz(os.path.join(golden_dir, "sam_%s_1.fa.xz" % case))
        n2, r2 = samutil.read_fasta_xz(os.path.join(golden_dir, "sam_%s_2.fa.xz" % case))
        out = ix.kalign_pe_batch(r1, r2, **pe, **kw)
        names = [x for p in zip(n1, n2) for x in p]
        reads = [x for p in zip(r1, r2) for x in p]
        res = [dict(nar=int(o["nar"]), hit=o["hit"], pe_aligned=int(o["pe_aligned"])) for o in out]
        got = samutil.sam_records(names, reads, res, CHROMS, paired=True)
    assert sorted(got) == sorted(recs)
    ix.close()


def _fabricated_loci(seed, n_reads, max_ml, span):
    """AlignRead-like results made up directly: unique and multi-aligned reads piled densely on three chromosomes, the
    third holding loci of multi-aligned reads only (so that loci get won by clustering with other multi-aligned reads and
    the orphan walk has chains to follow)."""
    import kit4b_amd as k4m

    rng = np.random.default_rng(seed)
    out = np.zeros(n_reads, dtype=k4m.RESULT_DTYPE)
    hits = np.zeros((n_reads, max_ml), dtype=k4m.HIT_DTYPE)
    hot = rng.integers(0, span, size=12)  # piles on chromosome 3
    for i in range(n_reads):
        u = rng.random()
        if u < 0.05:
            out[i]["hit_rslt"], out[i]["nar"] = 0, 3
            continue
        if u < 0.08:  # over the instance limit
            out[i]["hit_rslt"], out[i]["inst"], out[i]["nar"] = 3, max_ml + 1, 5
            continue
        inst = 1 if u < 0.55 else int(rng.integers(2, max_ml + 1))
        ln = int(rng.integers(50, 121))
        seen = set()
        for q in range(inst):
            while True:
                if inst > 1 and rng.random() < 0.45:
                    c, p = 3, int(hot[rng.integers(0, len(hot))] + rng.integers(-60, 61)) % span
                else:
                    c, p = int(rng.integers(1, 3)), int(rng.integers(0, span))
                s = int(rng.choice([43, 45]))
                if (c, p, s) not in seen:
                    seen.add((c, p, s))
                    break
            hits[i, q] = (c, p, ln, s, int(rng.integers(0, 4)), 0)
        out[i]["hit_rslt"], out[i]["inst"], out[i]["low_mm"] = 1, inst, 0
        out[i]["nar"], out[i]["num_hits"] = (1, 1) if inst == 1 else (5, inst)
    return out, hits


@pytest.mark.parametrize("ml_mode,seed,n_reads,span", [(4, 1, 6000, 3000), (3, 2, 6000, 3000), (4, 3, 30000, 40000), (3, 4, 2000, 400),
                                                       (4, 5, 2000, 400)])
def test_assign_multi_matches_vs_oracle(k4, oracle, ml_mode, seed, n_reads, span):
    """k4_assign_multi_dev == the oracle's AssignMultiMatches restatement (pinned to the reference by the se_r3 / se_r4
    goldens) on made-up loci dense enough for every branch: scores next to unique and next to multi-aligned reads, ties,
    winners by clustering with multi-aligned reads and the orphan walk over them."""
    out, hits = _fabricated_loci(seed, n_reads, 6, span)
    o_out, o_hits = out.copy(), hits.copy()
    want = oracle.assign_multi_matches(o_out, o_hits, ml_mode, 120, threads=1)
    ix = k4.SfxIndex.open(os.path.join(GOLDEN, "g1.sfx"))
    g_out, g_hits, got = ix.assign_multi(out, hits, ml_mode, 120)
    ix.close()
    assert got == want and want > 10
    assert (g_out == o_out).all()
    acc = g_out["nar"] == 1
    assert (g_hits[acc, 0] == o_hits[acc, 0]).all() and (g_hits["reserved"] == 0).all()
    changed = acc & (out["nar"] == 5)
    assert changed.sum() == want
