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
NAR_CODE = {"AA": 1, "EN": 2, "NL": 3, "MH": 4, "ML": 5, "ET": 6, "OJ": 7, "OM": 8, "UI": 13, "OI": 14, "UP": 15, "IS": 16, "IT": 17,
            "NP": 18}
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
        elif a in ("-r1", "-r3", "-r4"): kw["pe_mode"] = 1              # eMLdist / eMLuniq / eMLmulti: multi-aligned reads keep their loci
        elif a == "-X": kw["pe_mode"] = max(kw.get("pe_mode", 0), 3)   # ... reads over the -R limit clamped to it
        elif a == "-N": kw["pe_mode"] = 4                               # ... through LocateBestMatches
        elif a.startswith("-c"): kw["min_chimeric_len"] = int(a[2:])    # the optional AlignReads phases
        elif a.startswith("-a"): kw["micro_indel_len"] = int(a[2:])
        elif a.startswith("-A"): kw["max_splice_junct_len"] = int(a[2:])
        elif a.startswith("-x"): pe["min_flank_exacts"] = int(a[2:])
        elif a.startswith("-Q"): kw["strand"] = int(a[2:])             # 0 either, 1 Watson, 2 Crick
        elif a.startswith("-n"): kw["max_ns"] = int(a[2:])
        elif a == "-E": pe["pair_strand"] = True                        # both ends expected on the same strand
    # `-A` without `-c` / `-x` forces the flank autotrim to -s exact bases (KAlignerCL.cpp:829-830)
    if kw.get("max_splice_junct_len") and not kw.get("min_chimeric_len") and not pe.get("min_flank_exacts"):
        pe["min_flank_exacts"] = kw["max_subs"]
    return kw, pe


def ext_case(args):
    return any(a[:2] in ("-c", "-a", "-A", "-x") for a in args)


def oracle_se_ext(oracle, h, reads, kw, post):
    """CKAligner::AlignRead with -c / -a / -A, then the filters of CKAligner::Align (KAligner.cpp:653-686)"""
    from oracle_bindings import EXT_INDEL, EXT_SPLICE

    r = oracle.kalign_ext_batch(h, reads, **kw)
    if post.get("min_flank_exacts"):
        oracle.auto_trim_flanks(h, reads, r["out"], r["hits"], r["seg2"], post["min_flank_exacts"])
    if kw.get("max_splice_junct_len"):
        oracle.remove_orphan_juncts(EXT_SPLICE, r["out"], r["hits"], r["seg2"])
    if kw.get("micro_indel_len"):
        oracle.remove_orphan_juncts(EXT_INDEL, r["out"], r["hits"], r["seg2"])
    return r


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


@pytest.mark.parametrize("case", sorted(c for c in CASES if c.startswith("se_") and ext_case(CASES[c]["args"])))
def test_se_ext_matches_reference_sam(oracle, golden_dir, g3_path, case):
    """kalign -c / -a / -A / -x: soft-clipped, I / D / N CIGARs, scaled MAPQ, orphan filters, flank autotrim"""
    kw, post = kalign_args(CASES[case]["args"])
    names, reads = samutil.read_fasta_xz(os.path.join(golden_dir, "sam_%s.fa.xz" % case))
    h = oracle.open(g3_path)
    oracle.set_max_iter(h, 5000)
    r = oracle_se_ext(oracle, h, reads, kw, post)
    check_hist(r["out"]["nar"], CASES[case]["nar"])
    res = [dict(nar=int(o["nar"]), hit=hh[0], pe_aligned=0, seg2=s2) for o, hh, s2 in zip(r["out"], r["hits"], r["seg2"])]
    _, recs = samutil.read_sam_xz(os.path.join(golden_dir, "sam_%s.sam.xz" % case))
    assert sorted(samutil.sam_records(names, reads, res, ["chr1", "chr2", "chr3"])) == sorted(recs)
    cig = [l.split("\t")[5] for l in recs]
    if kw.get("min_chimeric_len"):
        assert sum("S" in c for c in cig) > 50
    if kw.get("micro_indel_len"):
        assert sum("I" in c for c in cig) > 20 and sum("D" in c for c in cig) > 20
    if kw.get("max_splice_junct_len"):
        assert sum("N" in c for c in cig) > 50
    oracle.close(h)


@pytest.mark.parametrize("case", sorted(c for c in CASES if c.startswith("se_") and not ext_case(CASES[c]["args"])))
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


def test_reference_dies_on_wide_rescue_window(golden_dir, tmp_path):
    """Why there is no golden SAM for `-U1 -d100 -D1500`: outside its chimeric mode AlignPairedRead (SfxArray.cpp:8616-8620,
    8685-8690) hands IterateExactsRange (:3461-3553) a seed of length 0 for insert windows of 1000 loci or more; that loop ends
    only on a mismatch, so it runs past the end of the suffix array.  The reference binary (built by oracle/Makefile, present in
    the build container only) is run on the pe_u1 reads: it must die on a signal, not write a SAM."""
    import subprocess

    ngs = os.path.join(os.path.dirname(golden_dir), "..", "oracle", "_ref", "ngskit4b")
    if not os.path.exists(ngs):
        pytest.skip("oracle/_ref/ngskit4b is not built here")
    files = []
    for s_ in ("1", "2"):
        n, r = samutil.read_fasta_xz(os.path.join(golden_dir, "sam_pe_u1_%s.fa.xz" % s_))
        fa = tmp_path / ("r%s.fa" % s_)
        with open(fa, "w") as f:
            for a, b in zip(n, r):
                f.write(">%s\n%s\n" % (a, "".join("ACGTN"[min(int(x), 4)] for x in b)))
        files.append(str(fa))
    r = subprocess.run([ngs, "kalign", "-I", os.path.join(golden_dir, "g1.sfx"), "-o", str(tmp_path / "o.sam"), "-T", "2", "-F",
                        str(tmp_path / "log.txt"), "-s2", "-U1", "-d100", "-D1500", "-i", files[0], "-u", files[1]],
                       capture_output=True, timeout=300)
    assert r.returncode < 0, r.returncode  # killed by a signal (SIGSEGV)
    # the same reads with a window below 1000 loci run to completion (the committed golden sam_pe_u1 is -d200 -D600)
    r = subprocess.run([ngs, "kalign", "-I", os.path.join(golden_dir, "g1.sfx"), "-o", str(tmp_path / "o2.sam"), "-T", "2", "-F",
                        str(tmp_path / "log2.txt"), "-s2", "-U1", "-d200", "-D600", "-i", files[0], "-u", files[1]],
                       capture_output=True, timeout=300)
    assert r.returncode == 0


ALL_CASES = json.load(open(os.path.join(GOLDEN, "sam_all_cases.json")))


@pytest.mark.parametrize("case", sorted(ALL_CASES))
def test_all_reads_mode_matches_reference_sam(oracle, golden_dir, g3_path, case):
    """kalign -M1 (eFMsamAll): the alignments of the base case, then one unaligned record per read that was not accepted --
    FLAG 4 (+ the pair bits), RNAME *, POS 0, MAPQ 128, CIGAR <len>M, the read as loaded, an empty field and YU:Z:<NAR> -- grouped
    by NAR in ascending order behind the alignments (WriteBAMReadHits / ReportBAMread, KAligner.cpp:5846-5866, 6253-6276)"""
    meta = ALL_CASES[case]
    base = meta["reads_of"]
    args = [a for a in meta["args"] if a != "-M1"]
    kw, pe = kalign_args(args)
    if base.startswith("pe_"):
        n1, r1 = samutil.read_fasta_xz(os.path.join(golden_dir, "sam_%s_1.fa.xz" % base))
        n2, r2 = samutil.read_fasta_xz(os.path.join(golden_dir, "sam_%s_2.fa.xz" % base))
        h = oracle.open(os.path.join(golden_dir, "g1.sfx"))
        oracle.set_max_iter(h, 5000)
        out = oracle_kalign_pe(oracle, h, r1, r2, threads=4, **pe, **kw)
        check_hist(out["nar"], meta["nar"])
        names = [x for p in zip(n1, n2) for x in p]
        reads = [x for p in zip(r1, r2) for x in p]
        res = [dict(nar=int(o["nar"]), hit=o["hit"], pe_aligned=int(o["pe_aligned"])) for o in out]
        got = samutil.sam_records(names, reads, res, CHROMS, paired=True, all_reads=True)
    else:
        names, reads = samutil.read_fasta_xz(os.path.join(golden_dir, "sam_%s.fa.xz" % base))
        h = oracle.open(g3_path if meta.get("index") == "g3" else os.path.join(golden_dir, "g1.sfx"))
        oracle.set_max_iter(h, 5000)
        if ext_case(args):
            r = oracle_se_ext(oracle, h, reads, kw, pe)
            res = [dict(nar=int(o["nar"]), hit=hh[0], pe_aligned=0, seg2=s2) for o, hh, s2 in zip(r["out"], r["hits"], r["seg2"])]
        else:
            r = oracle.kalign_batch(h, reads, **kw)
            res = [dict(nar=int(o["nar"]), hit=hh[0], pe_aligned=0) for o, hh in zip(r["out"], r["hits"])]
        check_hist(r["out"]["nar"], meta["nar"])
        got = samutil.sam_records(names, reads, res, ["chr1", "chr2", "chr3"] if meta.get("index") == "g3" else CHROMS, all_reads=True)
    _, recs = samutil.read_sam_xz(os.path.join(golden_dir, "sam_%s.sam.xz" % case))
    assert len(recs) == len(reads) and sorted(got) == sorted(recs)
    # order in the file: the alignments first (the base case's body, line for line), then the NAR codes ascending
    _, base_recs = samutil.read_sam_xz(os.path.join(golden_dir, "sam_%s.sam.xz" % base))
    assert recs[:len(base_recs)] == base_recs
    codes = [samutil.NAR_CODES.index(l.rsplit("YU:Z:", 1)[1]) for l in recs[len(base_recs):]]
    assert codes == sorted(codes) and len(set(codes)) >= 3
    oracle.close(h)


EXTRA_CASES = json.load(open(os.path.join(GOLDEN, "sam_extra_cases.json")))


@pytest.mark.parametrize("case", sorted(c for c in EXTRA_CASES if "_x" not in c))  # (flank autotrim over pairs: the GPU test only)
def test_runs_with_other_arguments_match_reference_sam(oracle, golden_dir, case):
    """the reads of another case under other arguments: kalign's defaults for the pairing options (`-u` alone: -U2, 100..1000) and
    alignments to one strand only (-Q1 / -Q2, single-end and through the paired-end flow)"""
    meta = EXTRA_CASES[case]
    base = meta["reads_of"]
    kw, pe = kalign_args(meta["args"])
    h = oracle.open(os.path.join(golden_dir, "g1.sfx"))
    oracle.set_max_iter(h, {0: 5000, 1: 10000, 2: 20000, 3: 2500}[kw["pmode"]])  # -m: KAligner.cpp's MaxIter per sensitivity
    _, recs = samutil.read_sam_xz(os.path.join(golden_dir, "sam_%s.sam.xz" % case))
    y = sum(int(a[2:]) for a in meta["args"] if a.startswith("-y"))
    Y = sum(int(a[2:]) for a in meta["args"] if a.startswith("-Y"))
    nth = max([int(a[2:]) for a in meta["args"] if a.startswith("-#")] + [1])  # -#<n>: every n-th read / pair, the first included (:11983-11989)
    trim = lambda rs: [r[y:len(r) - Y] for r in rs][::nth]  # noqa: E731  (-y / -Y: off the ends when loading, KAligner.cpp:12254-12260)
    if base.startswith("pe_"):
        if pe["pe_mode"] == 0:
            pe["pe_mode"] = 2  # KAlignerCL.cpp:546-553
        n1, r1 = samutil.read_fasta_xz(os.path.join(golden_dir, "sam_%s_1.fa.xz" % base))
        n2, r2 = samutil.read_fasta_xz(os.path.join(golden_dir, "sam_%s_2.fa.xz" % base))
        r1, r2, n1, n2 = trim(r1), trim(r2), n1[::nth], n2[::nth]
        out = oracle_kalign_pe(oracle, h, r1, r2, threads=4, **pe, **kw)
        check_hist(out["nar"], meta["nar"])
        names = [x for p in zip(n1, n2) for x in p]
        reads = [x for p in zip(r1, r2) for x in p]
        res = [dict(nar=int(o["nar"]), hit=o["hit"], pe_aligned=int(o["pe_aligned"])) for o in out]
        got = samutil.sam_records(names, reads, res, CHROMS, paired=True)
    else:
        names, reads = samutil.read_fasta_xz(os.path.join(golden_dir, "sam_%s.fa.xz" % base))
        reads, names = trim(reads), names[::nth]
        lo = max([int(a[2:]) for a in meta["args"] if a.startswith("-l")] + [50])   # the length filter of LoadRawReads (defaults 50 .. 500)
        hi = min([int(a[2:]) for a in meta["args"] if a.startswith("-L")] + [500])
        keep = [k for k, rd in enumerate(reads) if lo <= len(rd) <= hi]
        reads, names = [reads[k] for k in keep], [names[k] for k in keep]
        r = oracle.kalign_batch(h, reads, **kw)
        check_hist(r["out"]["nar"], meta["nar"])
        got = samutil.sam_records(names, reads, [dict(nar=int(o["nar"]), hit=hh[0], pe_aligned=0) for o, hh in zip(r["out"], r["hits"])], CHROMS)
    assert sorted(got) == sorted(recs) and len(recs) == meta["nar"]["AA"] > 0
    if "-Q1" in meta["args"] and not base.startswith("pe_"):
        assert all(int(l.split("\t")[1]) & 16 == 0 for l in recs)
    if "-Q2" in meta["args"] and not base.startswith("pe_"):
        assert all(int(l.split("\t")[1]) & 16 for l in recs)
    oracle.close(h)
