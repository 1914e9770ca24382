"""End-to-end golden outputs from the REAL reference front end (`oracle/_ref/ngskit4b`, built by `make -C oracle ngskit4b`).

    python tests/golden/make_golden_sam.py

For each case: the reads (FASTA, xz), the command line, the SAM the reference wrote (xz) and the NAR histogram it logged
(ReportAlignStats, ngskit4b/KAligner.cpp:3600-3830).  Data only.  The index is tests/golden/g1.sfx.
"""
import json
import lzma
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import synth  # noqa: E402

NGS = os.path.join(ROOT, "oracle", "_ref", "ngskit4b")

CASES = {
    # name: (kalign args, read generator)
    "se_s2": (["-s2"], lambda ch: synth.make_reads(ch, 3000, 100, seed=4321, n_prob=0.04, edge_frac=0.08, random_frac=0.04)[0]),
    "se_s0": (["-s0"], lambda ch: synth.make_reads(ch, 1500, 100, seed=4322, sub_lambda=0.4, edge_frac=0.05)[0]),
    "se_s5_e2_m1": (["-s5", "-e2", "-m1"], lambda ch: synth.make_reads(ch, 1500, 120, seed=4323, sub_lambda=2.5, n_prob=0.03)[0]),
    # MLMode eMLall: every locus of a multi-aligned read is reported, up to -R
    "se_r5_R12": (["-s2", "-r5", "-R12"], lambda ch: synth.make_reads(ch, 3000, 100, seed=4324, n_prob=0.02, edge_frac=0.05)[0]),
    # the same with reads over the limit clamped to -R (-X), and with LocateBestMatches instead of AlignReads (-N)
    "se_r5_R6_X": (["-s2", "-r5", "-R6", "-X"], lambda ch: synth.make_reads(ch, 3000, 100, seed=4325, n_prob=0.02, edge_frac=0.05)[0]),
    # MLMode eMLrand: one instance, picked by rand() in load order; the reference is run with ONE thread for this case (with
    # more its draws depend on thread timing)
    "se_r2_R8": (["-s2", "-r2", "-R8"], lambda ch: synth.make_reads(ch, 3000, 100, seed=4327, n_prob=0.02, edge_frac=0.05)[0]),
    # reads of 50 .. 700 bases in one file (-L800): every length class of the kernels against the reference's own run
    "se_lengths": (["-s3", "-l45", "-L800"], lambda ch: sum((synth.make_reads(ch, 250, L, seed=4400 + L, sub_lambda=1.0 + L / 100.0, n_prob=0.01, edge_frac=0.03)[0]
                                                             for L in (50, 64, 100, 129, 160, 161, 256, 257, 300, 513, 700)), [])),
    "se_r5_R8_N": (["-s3", "-r5", "-R8", "-N"], lambda ch: synth.make_reads(ch, 3000, 110, seed=4326, sub_lambda=1.5, n_prob=0.02, edge_frac=0.05)[0]),
}
# MLMode eMLuniq / eMLmulti (`-r3` / `-r4`): AssignMultiMatches (KAligner.cpp:5092) gives a multi-aligned read the locus that
# clusters with other reads; on the repeat-family genome synth.cluster_genome(), whose index the reference builds here too
CLUSTER_CASES = {
    "se_r3_R8": ["-s2", "-r3", "-R8"],
    "se_r4_R8": ["-s2", "-r4", "-R8"],
}
# the optional AlignReads phases and the stages kalign runs with them, on the g3 genome of make_golden_ext.py
# (planted introns): -c chimeric trimming (soft clips), -a microInDels (I / D, orphan filter), -A splice junctions (N, flank
# autotrim to -s exact bases, orphan filter), -x flank autotrim alone
EXT_CASES = {
    "se_c50": (["-s2", "-c50"], lambda ch, sites: synth.make_reads(ch, 500, 100, seed=5001, n_prob=0.02, edge_frac=0.05)[0]
               + synth.make_ext_reads(ch, 900, 100, "chimeric", seed=5002, max_subs=2)),
    "se_a12": (["-s2", "-a12"], lambda ch, sites: synth.make_reads(ch, 500, 100, seed=5003, n_prob=0.02)[0]
               + synth.make_variant_reads(ch, 60, 8, 100, "indel", seed=5004) + synth.make_ext_reads(ch, 150, 100, "indel", seed=5005)),
    "se_A3000": (["-s2", "-A3000"], lambda ch, sites: synth.make_reads(ch, 500, 100, seed=5006, n_prob=0.02, sub_lambda=1.5)[0]
                 + synth.make_variant_reads(ch, 45, 8, 100, "splice", seed=5007, sites=sites)
                 + synth.make_ext_reads(ch, 100, 100, "splice", seed=5008, sites=sites[45:])),
    "se_all_120": (["-s3", "-c60", "-a10", "-A2500"], lambda ch, sites: synth.make_reads(ch, 300, 120, seed=5009)[0]
                   + synth.make_ext_reads(ch, 400, 120, "chimeric", seed=5010, max_subs=3)
                   + synth.make_variant_reads(ch, 40, 6, 120, "indel", seed=5011)
                   + synth.make_variant_reads(ch, 40, 6, 120, "splice", seed=5012, sites=sites)),
    "se_x4": (["-s4", "-x4"], lambda ch, sites: synth.make_reads(ch, 1500, 100, seed=5013, sub_lambda=3.0, n_prob=0.02, edge_frac=0.05)[0]),
}
ONLY = [a for a in sys.argv[1:] if not a.startswith("-")]  # case names: regenerate just these (others keep their files)
PE_CASES = {
    "pe_u2": (["-s2", "-U2", "-d200", "-D600"], dict(seed=99, n_prob=0.02, random_mate_frac=0.03)),
    "pe_u1": (["-s2", "-U1", "-d200", "-D600"], dict(seed=98, n_prob=0.02, random_mate_frac=0.03, sub_lambda=2.0)),
    # chimeric trimming with paired ends (-c: AlignReads' chimeric pass for both ends, trimmed loci in the pairing, AlignPairedRead's
    # AdaptiveTrim branch for the rescue): a third of the mates carry foreign flanks.  The second case has an insert window of 1000
    # loci and more, where the rescue is seeded with exact cores (IterateExactsRange)
    "pe_c50_u1": (["-s2", "-c50", "-U1", "-d200", "-D600"], dict(seed=97, n_prob=0.02, random_mate_frac=0.03, sub_lambda=1.5, chimeric=0.33)),
    "pe_c60_u3_wide": (["-s3", "-c60", "-U3", "-d150", "-D1400"], dict(seed=96, n_prob=0.01, random_mate_frac=0.02, sub_lambda=1.5, chimeric=0.33,
                                                                        frag_min=300, frag_max=1300)),
}


# `-M1` (eFMsamAll): the same runs with every loaded read in the SAM -- the reads that were not accepted follow the alignments as
# unaligned records with a YU:Z:<NAR> tag.  The reads are those of the base case (no second copy); kept in sam_all_cases.json
ALL_READS_CASES = {"se_s2_M1": "se_s2", "se_c50_M1": "se_c50", "pe_u1_M1": "pe_u1", "pe_c60_u3_wide_M1": "pe_c60_u3_wide"}


# read names (QNAME): the descriptor up to its first white space, at most 79 characters; FASTA descriptors start behind the blanks and
# tabs that follow '>' (FASTQ ones do not).  names.fa / names.fq (the first reads of se_s2) -> names_fa.sam / names_fq.sam, run with -M1
NAMES = ["A" * 100, "short/1 extra words", "tab\tafter", " leading_space", "\t \ttabs_and_blanks x", "x" * 79, "y" * 80, "z" * 78 + " q",
         "with|pipe:colon;semi", "UPPER_lower-123.4", "B" * 130 + " tail"]


def unaligned_fasta_cases(tmp):
    """-j / -J: the reads without an alignment / the multi-aligned reads as FASTA (ReportNoneAligned / ReportMultiAlign): se_s2's and
    pe_u1's reads (the SAM of these runs is the base case's) -> unal_<case>_none.fa.xz, unal_<case>_multi.fa.xz"""
    for base, args in (("se_s2", ["-s2"]), ("pe_u1", ["-s2", "-U1", "-d200", "-D600"])):
        files = []
        for flag, suffix in (("-i", "_1"), ("-u", "_2")) if base.startswith("pe") else (("-i", ""),):
            fa = os.path.join(tmp, "%s%s.unal.fa" % (base, suffix))
            with lzma.open(os.path.join(HERE, "sam_%s%s.fa.xz" % (base, suffix)), "rb") as f, open(fa, "wb") as g:
                g.write(f.read())
            files += [flag, fa]
        none, multi = os.path.join(tmp, base + ".none.fa"), os.path.join(tmp, base + ".multi.fa")
        subprocess.run([NGS, "kalign", "-I", os.path.join(HERE, "g1.sfx"), "-o", os.path.join(tmp, base + ".unal.sam"), "-T", "4", "-F",
                        os.path.join(tmp, base + ".unal.log"), "-j", none, "-J", multi] + args + files, check=True, capture_output=True, timeout=600)
        for src, tag in ((none, "none"), (multi, "multi")):
            with open(src, "rb") as f, lzma.open(os.path.join(HERE, "unal_%s_%s.fa.xz" % (base, tag)), "wb", preset=9) as g:
                g.write(f.read())
        print("unaligned fasta", base, open(none).read().count(">"), open(multi).read().count(">"))


def names_case(tmp):
    seqs = [l.strip() for l in lzma.open(os.path.join(HERE, "sam_se_s2.fa.xz"), "rt") if not l.startswith(">")]
    with open(os.path.join(HERE, "names.fa"), "w") as f, open(os.path.join(HERE, "names.fq"), "w") as q:
        for k, nm in enumerate(NAMES):
            f.write(">%s\n%s\n" % (nm, seqs[k]))
            q.write("@%s\n%s\n+\n%s\n" % (nm, seqs[k], "I" * len(seqs[k])))
    for ext in ("fa", "fq"):
        subprocess.run([NGS, "kalign", "-I", os.path.join(HERE, "g1.sfx"), "-o", os.path.join(HERE, "names_%s.sam" % ext), "-T", "1", "-F",
                        os.path.join(tmp, "names.log"), "-s2", "-M1", "-i", os.path.join(HERE, "names." + ext)], check=True, capture_output=True)


# runs that reuse another case's reads with other arguments (sam_extra_cases.json): -u without -U (kalign then takes -U2, insert
# sizes 100..1000)
EXTRA_CASES = {"pe_defaults": ("pe_u1", ["-s2"]),
               # -Q: alignments to one strand only (Align2Strand of AlignReads; the PE flow's single-end pass included)
               "se_Q1": ("se_s2", ["-s2", "-Q1"]), "se_Q2": ("se_s2", ["-s2", "-Q2"]),
               "pe_u1_Q1": ("pe_u1", ["-s2", "-U1", "-d200", "-D600", "-Q1"]), "pe_u3_Q2": ("pe_u1", ["-s2", "-U3", "-d200", "-D600", "-Q2"]),
               # -y / -Y: bases taken off the 5' / 3' end of every read when loading (the SAM shows the trimmed read)
               "se_y7_Y12": ("se_s2", ["-s2", "-y7", "-Y12"]),
               # -4: with more reference sequences than this only those with alignments are declared in the SAM header
               "se_s2_sq2": ("se_s2", ["-s2", "-4", "2"]),
               # -n: indeterminate bases allowed in a read (a percentage of the length for reads over 100 bases)
               "se_n0": ("se_s2", ["-s2", "-n0"]), "se_n3": ("se_s2", ["-s2", "-n3"]), "pe_u1_n4": ("pe_u1", ["-s2", "-U1", "-d200", "-D600", "-n4"]),
               # -m: sensitivity (MaxIter, core sizes), -e2: two edits to the next best
               "se_m2_e2": ("se_s2", ["-s3", "-m2", "-e2"]), "se_m3": ("se_s2", ["-s2", "-m3"]),
               # the remaining pairing modes: -U4 (unique ends, single-end fallback), -E (both ends on the same strand), -m1 with pairs
               "pe_u4": ("pe_u1", ["-s2", "-U4", "-d200", "-D600"]), "pe_u1_E": ("pe_u1", ["-s2", "-U1", "-d200", "-D600", "-E"]),
               "pe_u3_E_m1": ("pe_u1", ["-s3", "-U3", "-d150", "-D800", "-E", "-m1"]),
               # flank autotrim with pairs (AutoTrimFlanks over both ends; the pair survives or not as a whole: KAligner.cpp:653-686)
               "pe_u1_x4": ("pe_u1", ["-s4", "-U1", "-d200", "-D600", "-x4"]),
               # -r1: multi-aligned reads are looked at up to -R loci, for the statistics only
               "se_r1_R8": ("se_s2", ["-s2", "-r1", "-R8"]),
               # the length filter over a file of mixed lengths: 50-base reads under -l60, 513- and 700-base reads over -L300
               "se_lengths_l60_L300": ("se_lengths", ["-s3", "-l60", "-L300"]),
               # -#: every n-th read / pair of the file is loaded (the first included)
               "se_s2_nth3": ("se_s2", ["-s2", "-#3"]), "pe_u1_nth4": ("pe_u1", ["-s2", "-U1", "-d200", "-D600", "-#4"]), "pe_u1_y5_Y20": ("pe_u1", ["-s2", "-U1", "-d200", "-D600", "-y5", "-Y20", "-l120"])}


def foreign_flanks(reads, frac, seed):
    """a share of the reads gets 5..35 % of random sequence at its 5' and / or 3' end (chimeric reads)"""
    import numpy as np

    rng = np.random.default_rng(seed)
    for r in reads:
        if rng.random() >= frac:
            continue
        L = len(r)
        for side in (0, 1):
            if rng.random() < 0.65:
                k = int(rng.integers(L * 5 // 100, L * 35 // 100))
                if side == 0:
                    r[:k] = rng.integers(0, 4, k)
                else:
                    r[L - k:] = rng.integers(0, 4, k)
    return reads


def run(tmp, name, args, files, sfx=os.path.join(HERE, "g1.sfx")):
    sam = os.path.join(tmp, name + ".sam")
    log = os.path.join(tmp, name + ".log")
    cmd = [NGS, "kalign", "-I", sfx, "-o", sam, "-T", "1" if "-r2" in args else "4", "-F", log] + args + files
    subprocess.run(cmd, check=True, capture_output=True, timeout=600)
    hist = {}
    for line in open(log):
        m = re.search(r"\)\s+(\d+) \((\w\w)\) ", line)
        if m:
            hist[m.group(2)] = int(m.group(1))
    with open(sam, "rb") as f, lzma.open(os.path.join(HERE, "sam_%s.sam.xz" % name), "wb", preset=9) as g:
        g.write(f.read())
    return hist


def main():
    names, chroms = synth.golden_genome()
    meta = {}
    with tempfile.TemporaryDirectory() as tmp:
        if ONLY:
            meta = json.load(open(os.path.join(HERE, "sam_cases.json")))
        for name, (args, gen) in CASES.items():
            if ONLY and name not in ONLY:
                continue
            fa = os.path.join(tmp, name + ".fa")
            synth.write_fasta(fa, gen(chroms))
            hist = run(tmp, name, args, ["-i", fa])
            with open(fa, "rb") as f, lzma.open(os.path.join(HERE, "sam_%s.fa.xz" % name), "wb", preset=9) as g:
                g.write(f.read())
            meta[name] = dict(args=args, nar=hist)
            print(name, hist)
        if not ONLY or any(c in ONLY for c in CLUSTER_CASES):
            names2, chroms2 = synth.cluster_genome()
            g2fa, g2 = os.path.join(tmp, "g2.fa"), os.path.join(tmp, "g2.sfx")
            synth.write_fasta(g2fa, chroms2, names=names2)
            subprocess.run([NGS, "index", "-i", g2fa, "-o", g2, "-r", "g2", "-T", "4", "-F", os.path.join(tmp, "g2.log")],
                           check=True, capture_output=True)
            with open(g2, "rb") as f, lzma.open(os.path.join(HERE, "g2.sfx.xz"), "wb", preset=9) as g:
                g.write(f.read())
            fa = os.path.join(tmp, "cluster.fa")
            synth.write_fasta(fa, synth.make_reads(chroms2, 9000, 80, seed=4402, n_prob=0.01, edge_frac=0.02)[0])
            with open(fa, "rb") as f, lzma.open(os.path.join(HERE, "sam_se_cluster.fa.xz"), "wb", preset=9) as g:
                g.write(f.read())
            for name, args in CLUSTER_CASES.items():
                hist = run(tmp, name, args, ["-i", fa], sfx=g2)
                meta[name] = dict(args=args, nar=hist, index="g2", reads="sam_se_cluster.fa.xz")
                print(name, hist)
        if not ONLY or any(c in ONLY for c in EXT_CASES):
            import make_golden_ext

            names3, chroms3, sites3, _ = make_golden_ext.genome()
            g3 = os.path.join(tmp, "g3.sfx")
            with lzma.open(os.path.join(HERE, "g3.sfx.xz"), "rb") as f, open(g3, "wb") as g:
                g.write(f.read())
            for name, (args, gen) in EXT_CASES.items():
                if ONLY and name not in ONLY:
                    continue
                fa = os.path.join(tmp, name + ".fa")
                synth.write_fasta(fa, gen(chroms3, sites3))
                hist = run(tmp, name, args, ["-i", fa], sfx=g3)
                with open(fa, "rb") as f, lzma.open(os.path.join(HERE, "sam_%s.fa.xz" % name), "wb", preset=9) as g:
                    g.write(f.read())
                meta[name] = dict(args=args, nar=hist, index="g3")
                print(name, hist)
        for name, (args, kw) in PE_CASES.items():
            if ONLY and name not in ONLY:
                continue
            kw = dict(kw)
            chim = kw.pop("chimeric", 0.0)
            pe1, pe2, _ = synth.make_pe_reads(chroms, 2000, 150, **kw)
            if chim:
                pe1, pe2 = foreign_flanks(pe1, chim, kw["seed"] + 1000), foreign_flanks(pe2, chim, kw["seed"] + 2000)
            f1, f2 = os.path.join(tmp, name + "_1.fa"), os.path.join(tmp, name + "_2.fa")
            synth.write_fasta(f1, pe1)
            synth.write_fasta(f2, pe2)
            hist = run(tmp, name, args, ["-i", f1, "-u", f2])
            for k, fp in (("1", f1), ("2", f2)):
                with open(fp, "rb") as f, lzma.open(os.path.join(HERE, "sam_%s_%s.fa.xz" % (name, k)), "wb", preset=9) as g:
                    g.write(f.read())
            meta[name] = dict(args=args, nar=hist)
            print(name, hist)
        if not ONLY or "names" in ONLY:
            names_case(tmp)
        if not ONLY or "unaligned_fasta" in ONLY:
            unaligned_fasta_cases(tmp)
        if not ONLY or any(c in ONLY for c in EXTRA_CASES):
            extra_meta = {}
            for name, (base, args) in EXTRA_CASES.items():
                files = []
                for flag, suffix in (("-i", "_1"), ("-u", "_2")) if base.startswith("pe") else (("-i", ""),):
                    fa = os.path.join(tmp, "%s%s.extra.fa" % (base, suffix))
                    with lzma.open(os.path.join(HERE, "sam_%s%s.fa.xz" % (base, suffix)), "rb") as f, open(fa, "wb") as g:
                        g.write(f.read())
                    files += [flag, fa]
                hist = run(tmp, name, args, files)
                extra_meta[name] = dict(args=args, nar=hist, reads_of=base)
                print(name, hist)
            json.dump(extra_meta, open(os.path.join(HERE, "sam_extra_cases.json"), "w"), indent=1, sort_keys=True)
        if not ONLY or any(c in ONLY for c in ALL_READS_CASES):
            base_meta = meta if not ONLY else json.load(open(os.path.join(HERE, "sam_cases.json")))
            all_meta = {}
            for name, base in ALL_READS_CASES.items():
                b = base_meta[base]
                sfx = os.path.join(HERE, "g1.sfx")
                if b.get("index"):
                    sfx = os.path.join(tmp, b["index"] + ".sfx")
                    if not os.path.exists(sfx):
                        with lzma.open(os.path.join(HERE, b["index"] + ".sfx.xz"), "rb") as f, open(sfx, "wb") as g:
                            g.write(f.read())
                files = []
                for flag, suffix in (("-i", "_1"), ("-u", "_2")) if base.startswith("pe") else (("-i", ""),):
                    fa = os.path.join(tmp, "%s%s.all.fa" % (base, suffix))
                    with lzma.open(os.path.join(HERE, "sam_%s%s.fa.xz" % (base, suffix)), "rb") as f, open(fa, "wb") as g:
                        g.write(f.read())
                    files += [flag, fa]
                hist = run(tmp, name, b["args"] + ["-M1"], files, sfx=sfx)
                all_meta[name] = dict(args=b["args"] + ["-M1"], nar=hist, reads_of=base, **({"index": b["index"]} if b.get("index") else {}))
                print(name, hist)
            json.dump(all_meta, open(os.path.join(HERE, "sam_all_cases.json"), "w"), indent=1, sort_keys=True)
    if ONLY and not any(c in ONLY for c in list(CASES) + list(CLUSTER_CASES) + list(EXT_CASES) + list(PE_CASES)):
        return  # only cases that reuse reads were asked for: sam_cases.json stays as it is
    json.dump(meta, open(os.path.join(HERE, "sam_cases.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
