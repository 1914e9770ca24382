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


def _trims(h):
    e = int(h["reserved"])  # k4_hit.ext: TrimLeft | TrimRight << 12 | flags (chimeric 1<<24, InDel 1<<25, insert 1<<26, splice 1<<27)
    return e & 0xFFF, (e >> 12) & 0xFFF, e


def adj_start(h):
    """CKAligner::AdjStartLoci, KAligner.cpp:1633-1640"""
    tl, tr, _ = _trims(h)
    return int(h["match_loci"]) + (tl if h["strand"] == ord("+") else tr)


def adj_len(h):
    tl, tr, _ = _trims(h)
    return int(h["match_len"]) - tl - tr


def cigar_mapq(h, seg2, read_len):
    """ReportBAMread, KAligner.cpp:6148-6233: soft clips around a trimmed hit, N / I / D between two segments; MAPQ 254 (less
    20 for a junction, 10 for a microInDel) scaled by the aligned fraction of the read"""
    tl, tr, e = _trims(h)
    plus = h["strand"] == ord("+")
    lead, trail = (tl, tr) if plus else (tr, tl)
    ops = ("%dS" % lead if lead else "") + "%dM" % adj_len(h) + ("%dS" % trail if trail else "")
    mq, aligned = 254, adj_len(h)
    if seg2 is not None and e & ((1 << 25) | (1 << 27)):
        gap = int(seg2["match_loci"]) - (int(h["match_loci"]) + int(h["match_len"]))
        if e & (1 << 27):
            mq -= 20
            ops += "%dN" % gap
        else:
            mq -= 10
            ops += ("%dI" % (read_len - (int(h["match_len"]) + int(seg2["match_len"])))) if e & (1 << 26) else "%dD" % abs(gap)
        ops += "%dM" % int(seg2["match_len"])
        aligned += int(seg2["match_len"])
    return ops, min(254, max(1, int(mq * (aligned / read_len))))


def sam_records(names, reads, results, chrom_names, paired=False):
    """results: per read dict(nar, hit(chrom_id, match_loci, match_len, strand, reserved=ext), pe_aligned[, seg2]); for paired
    input reads are interleaved PE1, PE2.  Returns the SAM lines kit4b would write for the accepted reads (unsorted)."""
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
                pnext = adj_start(mh) + 1
                s, e = adj_start(h), adj_start(mh)
                tlen = (e - s) + adj_len(mh) if s <= e else (s - e) + adj_len(h)
            else:
                flag |= 0x8
        seq = "".join("ACGTN"[b] if b <= 3 else "N" for b in rd)
        if minus:
            seq = revcomp_str(seq)
        cigar, mapq = cigar_mapq(h, r.get("seg2"), len(rd))
        out.append("\t".join([nm, str(flag), chrom_names[int(h["chrom_id"]) - 1], str(adj_start(h) + 1), str(mapq),
                              cigar, rnext, str(pnext), str(tlen), seq, "*"]))
    return out
