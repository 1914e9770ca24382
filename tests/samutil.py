"""Test helpers: FASTA in, and the SAM fields kit4b derives from alignment results
(CKAligner::ReportBAMread ngskit4b/KAligner.cpp:5957-6320, CSAMfile::AddAlignment libkit4b/SAMfile.cpp:2194-2377)."""
import lzma

import numpy as np

CODE = {c: i for i, c in enumerate("ACGTN")}
CODE.update({c.lower(): i for c, i in list(CODE.items())})


def read_fasta_xz(path):
    names, seqs = [], []
    with lzma.open(path, "rt") as f:
        for line in f:
            line = line.rstrip("\n")
            if line.startswith(">"):
                names.append(line[1:].split()[0])
                seqs.append([])
            elif line:
                seqs[-1].append(line)
    reads = [np.array([CODE.get(c, 4) for c in "".join(s)], dtype=np.uint8) for s in seqs]
    return names, reads


def read_sam_xz(path):
    hdr, recs = [], []
    with lzma.open(path, "rt") as f:
        for line in f:
            line = line.rstrip("\n")
            (hdr if line.startswith("@") else recs).append(line)
    return hdr, recs


def revcomp_str(s):
    return s[::-1].translate(str.maketrans("ACGTN", "TGCAN"))


def sam_records(names, reads, results, chrom_names, paired=False):
    """results: per read dict(nar, hit(chrom_id, match_loci, match_len, strand), pe_aligned); for paired input reads are
    interleaved PE1, PE2.  Returns the SAM lines kit4b would write for the accepted reads (unsorted)."""
    out = []
    for i, (nm, rd, r) in enumerate(zip(names, reads, results)):
        if r["nar"] != 1:
            continue
        h = r["hit"]
        minus = h["strand"] == ord("-")
        flag = 0x10 if minus else 0
        rnext, pnext, tlen = "*", 0, 0
        if paired:
            mate = results[i ^ 1]
            flag |= 0x1 | 0x2 | (0x40 if i % 2 == 0 else 0x80)
            if r["pe_aligned"] and mate["pe_aligned"] and mate["nar"] == 1:
                mh = mate["hit"]
                if mh["strand"] == ord("-"):
                    flag |= 0x20
                rnext = "="
                pnext = int(mh["match_loci"]) + 1
                s, e = int(h["match_loci"]), int(mh["match_loci"])
                tlen = (e - s) + int(mh["match_len"]) if s <= e else (s - e) + int(h["match_len"])
            else:
                flag |= 0x8
        seq = "".join("ACGTN"[b] if b <= 3 else "N" for b in rd)
        if minus:
            seq = revcomp_str(seq)
        out.append("\t".join([nm, str(flag), chrom_names[int(h["chrom_id"]) - 1], str(int(h["match_loci"]) + 1), "254",
                              "%dM" % int(h["match_len"]), rnext, str(pnext), str(tlen), seq, "*"]))
    return out
