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


NAR_CODES = ["NA", "AA", "EN", "NL", "MH", "ML", "ET", "OJ", "OM", "DP", "DS", "FC", "PR", "UI", "OI", "UP", "IS", "IT", "NP", "LC"]  # m_NARdesc, KAligner.cpp:48-67


def sam_records(names, reads, results, chrom_names, paired=False, all_reads=False):
    """results: per read dict(nar, hit(chrom_id, match_loci, match_len, strand, reserved=ext), pe_aligned[, seg2]); for paired
    input reads are interleaved PE1, PE2.  Returns the SAM lines kit4b would write for the accepted reads (unsorted); all_reads:
    kalign -M1, the other reads too, as unaligned records (ReportBAMread's last branch, KAligner.cpp:6253-6276)."""
    out = []
    for i, (nm, rd, r) in enumerate(zip(names, reads, results)):
        if r["nar"] != 1:
            if all_reads:
                flag = 0x4
                if paired:
                    mate = results[i ^ 1]
                    flag |= 0x1 | 0x2 | (0x40 if i % 2 == 0 else 0x80)
                    if r["pe_aligned"] and mate["pe_aligned"] and mate["nar"] == 1:
                        flag |= 0x20 if mate["hit"]["strand"] == ord("-") else 0
                    else:
                        flag |= 0x8
                seq = "".join("ACGTN"[b] if b <= 3 else "N" for b in rd)
                out.append("\t".join([nm, str(flag), "*", "0", "128", "%dM" % len(rd), "*", "0", "0", seq, "*", "", "YU:Z:" + NAR_CODES[int(r["nar"])]]))
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


# ---- BAM (SAM specification 4.2) ------------------------------------------------------------------------------------
def read_bgzf(path):
    """(uncompressed bytes, [(compressed offset, uncompressed offset, uncompressed length)] per BGZF block)"""
    import struct
    import zlib

    raw = open(path, "rb").read()
    out, blocks, o = [], [], 0
    u = 0
    while o < len(raw):
        assert raw[o:o + 4] == b"\x1f\x8b\x08\x04", "not a BGZF member at %d" % o
        xlen = struct.unpack("<H", raw[o + 10:o + 12])[0]
        extra = raw[o + 12:o + 12 + xlen]
        bsize, e = None, 0
        while e < xlen:
            si1, si2, slen = extra[e], extra[e + 1], struct.unpack("<H", extra[e + 2:e + 4])[0]
            if si1 == 66 and si2 == 67:
                bsize = struct.unpack("<H", extra[e + 4:e + 6])[0] + 1
            e += 4 + slen
        assert bsize, "BGZF member without a BC field"
        cdata = raw[o + 12 + xlen:o + bsize - 8]
        data = zlib.decompress(cdata, -15)
        crc, isize = struct.unpack("<II", raw[o + bsize - 8:o + bsize])
        assert isize == len(data) and crc == (zlib.crc32(data) & 0xFFFFFFFF)
        blocks.append((o, u, len(data)))
        out.append(data)
        u += len(data)
        o += bsize
    return b"".join(out), blocks


def read_bam(path, with_offsets=False):
    """(header text, [(name, length)] reference dictionary, records).  A record is a dict of the fixed fields plus name, cigar
    [(len, op)], seq (string), qual (bytes), aux (bytes) and, with_offsets, ubeg / uend (uncompressed offsets)."""
    import struct

    d, blocks = read_bgzf(path)
    assert d[:4] == b"BAM\x01"
    lt = struct.unpack("<i", d[4:8])[0]
    text = d[8:8 + lt].decode()
    o = 8 + lt
    nref = struct.unpack("<i", d[o:o + 4])[0]
    o += 4
    refs = []
    for _ in range(nref):
        ln = struct.unpack("<i", d[o:o + 4])[0]
        name = d[o + 4:o + 4 + ln - 1].decode()
        sl = struct.unpack("<i", d[o + 4 + ln:o + 8 + ln])[0]
        refs.append((name, sl))
        o += 8 + ln
    recs = []
    while o < len(d):
        bs = struct.unpack("<i", d[o:o + 4])[0]
        b = d[o + 4:o + 4 + bs]
        ref, pos, bmn, fnc, lseq, nref_, npos, tlen = struct.unpack("<iiIIiiii", b[:32])
        lname, nops = bmn & 0xFF, fnc & 0xFFFF
        p = 32
        name = b[p:p + lname - 1].decode()
        p += lname
        cigar = []
        for _ in range(nops):
            v = struct.unpack("<I", b[p:p + 4])[0]
            cigar.append((v >> 4, "MIDNSHP=X"[v & 15]))
            p += 4
        sq = b[p:p + (lseq + 1) // 2]
        p += (lseq + 1) // 2
        seq = "".join("=ACMGRSVTWYHKDBN"[(sq[k >> 1] >> (4 if k % 2 == 0 else 0)) & 15] for k in range(lseq))
        qual = b[p:p + lseq]
        p += lseq
        r = dict(ref=ref, pos=pos, bin=bmn >> 16, mapq=(bmn >> 8) & 0xFF, flag=fnc >> 16, l_seq=lseq, next_ref=nref_, next_pos=npos,
                 tlen=tlen, name=name, cigar=cigar, seq=seq, qual=qual, aux=b[p:])
        if with_offsets:
            r["ubeg"], r["uend"] = o, o + 4 + bs
        recs.append(r)
        o += 4 + bs
    return (text, refs, recs, blocks) if with_offsets else (text, refs, recs)


def bam_record_as_sam_fields(r, refs):
    """the eleven mandatory SAM fields of a decoded BAM record, as CSAMfile::AddAlignment prints them (SAMfile.cpp:2218-2256)"""
    cigar = "".join("%d%s" % c for c in r["cigar"]) or "*"
    rnext = "*" if r["next_ref"] == -1 else "="
    qual = "*" if (len(r["qual"]) == 0 or r["qual"][0] == 0xFF) else r["qual"].decode()
    return (r["name"], r["flag"], refs[r["ref"]][0] if r["ref"] >= 0 else "*", r["pos"] + 1, r["mapq"], cigar, rnext,
            0 if r["next_ref"] == -1 else r["next_pos"] + 1, r["tlen"], r["seq"], qual)


def sam_line_fields(line):
    f = line.split("\t")
    return (f[0], int(f[1]), f[2], int(f[3]), int(f[4]), f[5], f[6], int(f[7]), int(f[8]), f[9], f[10])


def reg2bin(beg, end):
    end -= 1
    for sh, base in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        if beg >> sh == end >> sh:
            return base + (beg >> sh)
    return 0


def reg2bins(beg, end):
    end -= 1
    out = [0]
    for sh, base in ((26, 1), (23, 9), (20, 73), (17, 585), (14, 4681)):
        out += list(range(base + (beg >> sh), base + (end >> sh) + 1))
    return out


def read_bai(path):
    """[{bin: [(beg, end)]}, [linear offsets]] per reference"""
    import struct

    d = open(path, "rb").read()
    assert d[:4] == b"BAI\x01"
    n = struct.unpack("<i", d[4:8])[0]
    o = 8
    out = []
    for _ in range(n):
        if o >= len(d):  # the reference stops after the last sequence that has alignments (n_ref still counts them all)
            out.append(({}, []))
            continue
        nb = struct.unpack("<i", d[o:o + 4])[0]
        o += 4
        bins = {}
        for _ in range(nb):
            b, nc = struct.unpack("<Ii", d[o:o + 8])
            o += 8
            bins[b] = [struct.unpack("<QQ", d[o + 16 * k:o + 16 * k + 16]) for k in range(nc)]
            o += 16 * nc
        ni = struct.unpack("<i", d[o:o + 4])[0]
        o += 4
        lin = list(struct.unpack("<%dQ" % ni, d[o:o + 8 * ni]))
        o += 8 * ni
        out.append((bins, lin))
    return out


def bai_fetch(recs, blocks, index, ref, beg, end):
    """names of the records overlapping [beg, end) of reference `ref`, found THROUGH the index (chunks of the candidate bins, cut
    at the linear index' minimum offset), as a BAM reader would; recs / blocks from read_bam(..., with_offsets=True)"""
    bins, lin = index[ref]
    min_off = lin[beg >> 14] if (beg >> 14) < len(lin) else (lin[-1] if lin else 0)
    c2u = {c: u for c, u, _ in blocks}

    def to_u(v):  # virtual offset -> uncompressed offset
        return c2u[v >> 16] + (v & 0xFFFF)

    spans = []
    for b in reg2bins(beg, end):
        for cb, ce in bins.get(b, []):
            if ce > min_off:
                spans.append((to_u(max(cb, min_off)), to_u(ce)))
    got = set()
    for r in recs:
        if r["ref"] != ref or not any(a <= r["ubeg"] < z for a, z in spans):
            continue
        span = sum(n for n, op in r["cigar"] if op in "MDN=X") or 1
        if r["pos"] < end and r["pos"] + span > beg:
            got.add((r["name"], r["flag"], r["pos"]))
    return got
