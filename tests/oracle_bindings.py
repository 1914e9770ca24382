"""ctypes bindings for the TEST-ONLY checkers under oracle/:

* ``Oracle``  -> oracle/libk4oracle.so  (our plain-C restatement, always buildable)
* ``Ref``     -> oracle/_ref/libk4ref.so (the real reference library; only where it was built)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "libk4oracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libk4ref.so")

STRAND_BOTH, STRAND_WATSON, STRAND_CRICK = 0, 1, 2
HR_NONE, HR_HITS, HR_MMDELTA, HR_HITINSTS, HR_RMMDELTA, HR_SEQERRS = 0, 1, 2, 3, 4, 5
NAR_ACCEPTED, NAR_NS, NAR_NOHIT, NAR_MMDELTA, NAR_MULTIALIGN = 1, 2, 3, 4, 5

# 16-byte hit record, identical in the oracle (k4o_hit) and the product (k4_hit, include/k4sfx.h)
HIT_DTYPE = np.dtype(
    [("chrom_id", "<u4"), ("match_loci", "<u4"), ("match_len", "<u2"), ("strand", "u1"), ("mismatches", "u1"),
     ("reserved", "<u4")]
)
RESULT_DTYPE = np.dtype(
    [("hit_rslt", "<i4"), ("inst", "<i4"), ("low_mm", "<i4"), ("nxt_mm", "<i4"), ("nar", "<i4"), ("num_hits", "<i4")]
)
# the hit record's fourth word ("reserved" in the dtype, k4_hit.ext / k4o_hit.ext in C): trims and tsHitLoci flags of the
# optional AlignReads phases; 0 on the default path
EXT_CHIMERIC, EXT_INDEL, EXT_INSERT, EXT_SPLICE, EXT_NONORPHAN = 1 << 24, 1 << 25, 1 << 26, 1 << 27, 1 << 28
NAR_TRIM, NAR_SPLICEJCTN, NAR_MICROINDEL = 6, 7, 8
# Seg[1] of a two-segment hit + Score (k4_seg2 / k4o_seg2), one per read
SEG2_DTYPE = np.dtype(
    [("chrom_id", "<u4"), ("match_loci", "<u4"), ("match_len", "<u2"), ("read_ofs", "<u2"), ("mismatches", "u1"),
     ("reserved", "u1"), ("score", "<u2")]
)


class ExtParams(C.Structure):
    _fields_ = [("min_chimeric_len", C.c_int), ("micro_indel_len", C.c_int), ("max_splice_junct_len", C.c_int)]


def ext_trims(hits):
    """(TrimLeft, TrimRight) of hit records"""
    e = hits["reserved"]
    return e & 0xFFF, (e >> 12) & 0xFFF


class KalignParams(C.Structure):
    _fields_ = [
        ("max_subs", C.c_int), ("min_edit_dist", C.c_int), ("max_ns", C.c_int), ("pmode", C.c_int),
        ("strand", C.c_int), ("max_ml", C.c_int), ("pe_mode", C.c_int), ("min_core_len", C.c_int),
        ("max_num_slides", C.c_int), ("min_chimeric_len", C.c_int), ("micro_indel_len", C.c_int),
        ("max_splice_junct_len", C.c_int),
    ]


class PeParams(C.Structure):
    _fields_ = [("pe_mode", C.c_int), ("pair_min_len", C.c_int), ("pair_max_len", C.c_int), ("pair_strand", C.c_int)]


PE_READ_DTYPE = np.dtype(
    [("nar", "<i4"), ("num_hits", "<i4"), ("inst", "<i4"), ("low_mm", "<i4"), ("pe_aligned", "<i4"), ("rescued", "<i4"),
     ("hit", HIT_DTYPE)]
)


class Counters(C.Structure):
    _fields_ = [("n_lookup", C.c_uint64), ("n_probe", C.c_uint64), ("n_cand", C.c_uint64)]


class Entry(C.Structure):
    _fields_ = [
        ("entry_id", C.c_uint32), ("fblock_id", C.c_uint32), ("name", C.c_char * 81), ("name_hash", C.c_uint16),
        ("seq_len", C.c_uint32), ("start_ofs", C.c_uint64), ("end_ofs", C.c_uint64),
    ]


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR], stdout=subprocess.DEVNULL)


def _u8p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def _seq_args(names, seqs):
    n = len(seqs)
    seqs = [np.ascontiguousarray(s, dtype=np.uint8) for s in seqs]
    names_a = (C.c_char_p * n)(*[nm.encode() if isinstance(nm, str) else nm for nm in names])
    ptrs = (C.c_void_p * n)(*[s.ctypes.data for s in seqs])
    lens = (C.c_uint32 * n)(*[len(s) for s in seqs])
    return n, names_a, ptrs, lens, seqs


def flatten_reads(reads):
    """list of uint8 arrays -> (concat, offs u64, lens u32)"""
    lens = np.array([len(r) for r in reads], dtype=np.uint32)
    offs = np.zeros(len(reads), dtype=np.uint64)
    if len(reads):
        offs[1:] = np.cumsum(lens[:-1], dtype=np.uint64)
        cat = np.ascontiguousarray(np.concatenate(reads).astype(np.uint8))
    else:
        cat = np.zeros(0, dtype=np.uint8)
    return cat, offs, lens


class Oracle:
    """oracle/libk4oracle.so"""

    def __init__(self):
        if not os.path.exists(ORACLE_SO) or os.path.getmtime(ORACLE_SO) < os.path.getmtime(
            os.path.join(ORACLE_DIR, "k4oracle.c")
        ):
            build_oracle()
        L = C.CDLL(ORACLE_SO)
        L.k4o_open.restype = C.c_void_p
        L.k4o_open.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
        L.k4o_build.restype = C.c_void_p
        L.k4o_build.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p, C.c_int, C.c_int]
        L.k4o_from_parts.restype = C.c_void_p
        L.k4o_from_parts.argtypes = [C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_char_p]
        L.k4o_write.argtypes = [C.c_void_p, C.c_char_p]
        L.k4o_close.argtypes = [C.c_void_p]
        for f, rt in [("k4o_concat_len", C.c_uint64), ("k4o_el_size", C.c_uint32), ("k4o_seq", C.c_void_p),
                      ("k4o_sa_bytes", C.c_void_p), ("k4o_num_entries", C.c_uint32),
                      ("k4o_entries", C.POINTER(Entry)), ("k4o_tot_seqs_len", C.c_uint64)]:
            getattr(L, f).restype = rt
            getattr(L, f).argtypes = [C.c_void_p]
        L.k4o_sa_at.restype = C.c_int64
        L.k4o_sa_at.argtypes = [C.c_void_p, C.c_int64]
        L.k4o_set_max_iter.argtypes = [C.c_void_p, C.c_int]
        L.k4o_locate_first_exact.restype = C.c_int64
        L.k4o_locate_first_exact.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_void_p]
        L.k4o_min_core_len.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        L.k4o_align_batch.argtypes = [C.c_void_p, C.POINTER(KalignParams), C.c_int64, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(Counters)]
        L.k4o_align_reads_batch.argtypes = [C.c_void_p] + [C.c_int] * 8 + [C.c_int64] + [C.c_void_p] * 8 + [
            C.c_int, C.POINTER(Counters)]
        L.k4o_revcomp.argtypes = [C.c_void_p, C.c_int]
        L.k4o_kalign_pe_batch.argtypes = [C.c_void_p, C.POINTER(KalignParams), C.POINTER(PeParams), C.c_int64] + [
            C.c_void_p] * 7 + [C.c_int]
        L.k4o_pe_insert_size.argtypes = [C.c_int, C.c_int, C.c_int, C.c_uint8, C.c_uint32, C.c_uint32, C.c_uint8,
                                         C.c_uint32, C.c_uint32]
        self.L = L

    # -- index ------------------------------------------------------------------------------------
    def open(self, path):
        err = C.create_string_buffer(512)
        h = self.L.k4o_open(path.encode(), err, 512)
        if not h:
            raise RuntimeError(err.value.decode())
        return h

    def build(self, names, seqs, dataset="syn", force_el=0, threads=4):
        n, names_a, ptrs, lens, keep = _seq_args(names, seqs)
        h = self.L.k4o_build(n, names_a, ptrs, lens, dataset.encode(), force_el, threads)
        if not h:
            raise RuntimeError("k4o_build failed")
        return h

    def write(self, h, path):
        if self.L.k4o_write(h, path.encode()) != 0:
            raise RuntimeError("k4o_write failed")

    def close(self, h):
        self.L.k4o_close(h)

    def concat_len(self, h):
        return self.L.k4o_concat_len(h)

    def el_size(self, h):
        return self.L.k4o_el_size(h)

    def seq(self, h):
        n = self.concat_len(h)
        return np.ctypeslib.as_array(C.cast(self.L.k4o_seq(h), C.POINTER(C.c_uint8)), shape=(n,))

    def sa(self, h):
        """suffix array as int64 numpy (copy)"""
        n, el = self.concat_len(h), self.el_size(h)
        raw = np.ctypeslib.as_array(C.cast(self.L.k4o_sa_bytes(h), C.POINTER(C.c_uint8)), shape=(n * el,))
        raw = raw.reshape(n, el)
        v = raw[:, :4].copy().view("<u4").reshape(n).astype(np.int64)
        if el == 5:
            v |= raw[:, 4].astype(np.int64) << 32
        return v

    def entries(self, h):
        n = self.L.k4o_num_entries(h)
        p = self.L.k4o_entries(h)
        return [dict(entry_id=p[i].entry_id, name=p[i].name.decode(), seq_len=p[i].seq_len,
                     start_ofs=p[i].start_ofs, end_ofs=p[i].end_ofs) for i in range(n)]

    def set_max_iter(self, h, it):
        self.L.k4o_set_max_iter(h, it)

    def min_core_len(self, h, pmode=0):
        s = C.c_int(0)
        m = self.L.k4o_min_core_len(h, pmode, C.byref(s))
        return m, s.value

    def locate_first_exact(self, h, probe):
        probe = np.ascontiguousarray(probe, dtype=np.uint8)
        return self.L.k4o_locate_first_exact(h, probe.ctypes.data, len(probe), 0, self.concat_len(h) - 1, None)

    # -- alignment ----------------------------------------------------------------------------------
    def align_reads_batch(self, h, reads, tot_mm, core_len, core_delta, max_slides, min_core_len=0, mm_delta=1,
                          strand=STRAND_BOTH, max_hits=1, threads=4):
        """CSfxArray::AlignReads over a batch; reads = list of arrays or (cat, offs, lens)."""
        cat, offs, lens = reads if isinstance(reads, tuple) else flatten_reads(reads)
        n = len(lens)
        rslt = np.zeros(n, np.int32); inst = np.zeros(n, np.int32); low = np.zeros(n, np.int32)
        nxt = np.zeros(n, np.int32)
        hits = np.zeros((n, max_hits), dtype=HIT_DTYPE)
        ctr = Counters()
        self.L.k4o_align_reads_batch(h, tot_mm, core_len, core_delta, max_slides, min_core_len, mm_delta, strand,
                                     max_hits, n, cat.ctypes.data, offs.ctypes.data, lens.ctypes.data,
                                     rslt.ctypes.data, inst.ctypes.data, low.ctypes.data, nxt.ctypes.data,
                                     hits.ctypes.data, threads, C.byref(ctr))
        return dict(rslt=rslt, inst=inst, low=low, nxt=nxt, hits=hits,
                    counters=dict(n_lookup=ctr.n_lookup, n_probe=ctr.n_probe, n_cand=ctr.n_cand))

    def kalign_batch(self, h, reads, max_subs=5, min_edit_dist=1, max_ns=1, pmode=0, strand=STRAND_BOTH, max_ml=1,
                     pe_mode=0, min_core_len=0, max_num_slides=0, threads=4):
        """CKAligner::AlignRead over a batch."""
        cat, offs, lens = reads if isinstance(reads, tuple) else flatten_reads(reads)
        n = len(lens)
        kp = KalignParams(max_subs, min_edit_dist, max_ns, pmode, strand, max_ml, pe_mode, min_core_len,
                          max_num_slides)
        out = np.zeros(n, dtype=RESULT_DTYPE)
        hits = np.zeros((n, max(1, max_ml)), dtype=HIT_DTYPE)
        ctr = Counters()
        self.L.k4o_align_batch(h, C.byref(kp), n, cat.ctypes.data, offs.ctypes.data, lens.ctypes.data,
                               out.ctypes.data, hits.ctypes.data, threads, C.byref(ctr))
        return dict(out=out, hits=hits,
                    counters=dict(n_lookup=ctr.n_lookup, n_probe=ctr.n_probe, n_cand=ctr.n_cand))


    # -- the optional phases of AlignReads (oracle/k4oracle_ext.c) ---------------------------------------------------
    def align_reads_ext_batch(self, h, reads, tot_mm, core_len, core_delta, max_slides, min_core_len=0, mm_delta=1,
                              strand=STRAND_BOTH, max_hits=1, min_chimeric_len=0, micro_indel_len=0,
                              max_splice_junct_len=0, threads=4):
        """CSfxArray::AlignReads with MinChimericLen / microInDelLen / MaxSpliceJunctLen over a batch"""
        cat, offs, lens = reads if isinstance(reads, tuple) else flatten_reads(reads)
        n = len(lens)
        rslt = np.zeros(n, np.int32); inst = np.zeros(n, np.int32); low = np.zeros(n, np.int32)
        nxt = np.zeros(n, np.int32)
        hits = np.zeros((n, max_hits), dtype=HIT_DTYPE)
        seg2 = np.zeros(n, dtype=SEG2_DTYPE)
        ext = ExtParams(min_chimeric_len, micro_indel_len, max_splice_junct_len)
        vp = C.c_void_p
        self.L.k4o_align_reads_ext_batch(vp(h), C.byref(ext), tot_mm, core_len, core_delta, max_slides, min_core_len, mm_delta,
                                         strand, max_hits, C.c_int64(n), vp(cat.ctypes.data), vp(offs.ctypes.data),
                                         vp(lens.ctypes.data), vp(rslt.ctypes.data), vp(inst.ctypes.data), vp(low.ctypes.data),
                                         vp(nxt.ctypes.data), vp(hits.ctypes.data), vp(seg2.ctypes.data), threads, None)
        return dict(rslt=rslt, inst=inst, low=low, nxt=nxt, hits=hits, seg2=seg2)

    def kalign_ext_batch(self, h, reads, max_subs=5, min_edit_dist=1, max_ns=1, pmode=0, strand=STRAND_BOTH, max_ml=1,
                         pe_mode=0, min_core_len=0, max_num_slides=0, min_chimeric_len=0, micro_indel_len=0,
                         max_splice_junct_len=0, threads=4):
        """CKAligner::AlignRead over a batch with `-c` / `-a` / `-A`"""
        cat, offs, lens = reads if isinstance(reads, tuple) else flatten_reads(reads)
        n = len(lens)
        kp = KalignParams(max_subs, min_edit_dist, max_ns, pmode, strand, max_ml, pe_mode, min_core_len, max_num_slides,
                          min_chimeric_len, micro_indel_len, max_splice_junct_len)
        out = np.zeros(n, dtype=RESULT_DTYPE)
        hits = np.zeros((n, max(1, max_ml)), dtype=HIT_DTYPE)
        seg2 = np.zeros(n, dtype=SEG2_DTYPE)
        vp = C.c_void_p
        self.L.k4o_align_ext_batch(vp(h), C.byref(kp), C.c_int64(n), vp(cat.ctypes.data), vp(offs.ctypes.data),
                                   vp(lens.ctypes.data), vp(out.ctypes.data), vp(hits.ctypes.data), vp(seg2.ctypes.data),
                                   threads, None)
        return dict(out=out, hits=hits, seg2=seg2)

    def align_paired_read_x(self, h, b3prime, antisense, chrom_id, start_loci, end_loci, min_insert, max_insert, max_allowed_mm,
                            read, min_chimeric_len, core_len, core_delta):
        """CSfxArray::AlignPairedRead in chimeric mode: (rslt, hit record with ext = trims | flags)"""
        self.L.k4o_align_paired_read_x.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_int,
                                                   C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        rd = np.ascontiguousarray(read, dtype=np.uint8)
        hit = np.zeros(1, dtype=HIT_DTYPE)
        r = self.L.k4o_align_paired_read_x(h, int(b3prime), int(antisense), chrom_id, start_loci, end_loci, min_insert, max_insert,
                                           max_allowed_mm, len(rd), min_chimeric_len, core_len, core_delta, rd.ctypes.data,
                                           hit.ctypes.data)
        return r, hit[0]

    def snp_csv(self, h, reads, nar, hits, min_snp_reads=5, qvalue=0.05, snp_nonref_pcnt=25.0, vcf=False):
        """kalign's SNP calling over one reported alignment per read (hits: [n] or [n, max_ml] records): (CSV text, number of SNPs);
        vcf=True: the records of the VCF form"""
        cat, offs, lens = reads if isinstance(reads, tuple) else flatten_reads(reads)
        hits = np.ascontiguousarray(hits)
        stride = 1 if hits.ndim == 1 else hits.shape[1]
        nar = np.ascontiguousarray(nar, dtype=np.int32)
        self.L.k4o_snp_text.restype = C.c_void_p
        self.L.k4o_snp_text.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_int, C.c_double, C.c_double, C.POINTER(C.c_int64)]
        self.L.k4o_free.argtypes = [C.c_void_p]
        n = C.c_int64(0)
        p = self.L.k4o_snp_text(h, 1 if vcf else 0, len(lens), nar.ctypes.data, hits.ctypes.data, stride, cat.ctypes.data, offs.ctypes.data,
                                lens.ctypes.data, min_snp_reads, qvalue, snp_nonref_pcnt, C.byref(n))
        if not p:
            raise RuntimeError("k4o_snp_csv failed")
        text = C.string_at(p).decode()
        self.L.k4o_free(p)
        return text, n.value

    def snp_wig(self, h, reads, nar, hits, min_snp_reads=5, qvalue=0.05, snp_nonref_pcnt=25.0):
        """the coverage WIG kalign writes beside the SNP file (text)"""
        cat, offs, lens = reads if isinstance(reads, tuple) else flatten_reads(reads)
        hits = np.ascontiguousarray(hits)
        stride = 1 if hits.ndim == 1 else hits.shape[1]
        nar = np.ascontiguousarray(nar, dtype=np.int32)
        self.L.k4o_snp_wig.restype = C.c_void_p
        self.L.k4o_snp_wig.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                       C.c_double, C.c_double]
        self.L.k4o_free.argtypes = [C.c_void_p]
        p = self.L.k4o_snp_wig(h, len(lens), nar.ctypes.data, hits.ctypes.data, stride, cat.ctypes.data, offs.ctypes.data, lens.ctypes.data,
                               min_snp_reads, qvalue, snp_nonref_pcnt)
        if not p:
            raise RuntimeError("k4o_snp_wig failed")
        text = C.string_at(p).decode()
        self.L.k4o_free(p)
        return text

    def snp_haplotypes(self, h, n_loci, reads, nar, hits, min_snp_reads=5, qvalue=0.05, snp_nonref_pcnt=25.0):
        """the haplotype file kalign writes beside the SNP file: n_loci 2 = .disnp.csv, 3 = .trisnp.csv (text)"""
        cat, offs, lens = reads if isinstance(reads, tuple) else flatten_reads(reads)
        hits = np.ascontiguousarray(hits)
        stride = 1 if hits.ndim == 1 else hits.shape[1]
        nar = np.ascontiguousarray(nar, dtype=np.int32)
        self.L.k4o_snp_haplotypes.restype = C.c_void_p
        self.L.k4o_snp_haplotypes.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_int, C.c_double, C.c_double]
        self.L.k4o_free.argtypes = [C.c_void_p]
        p = self.L.k4o_snp_haplotypes(h, n_loci, len(lens), nar.ctypes.data, hits.ctypes.data, stride, cat.ctypes.data, offs.ctypes.data,
                                      lens.ctypes.data, min_snp_reads, qvalue, snp_nonref_pcnt)
        if not p:
            raise RuntimeError("k4o_snp_haplotypes failed")
        text = C.string_at(p).decode()
        self.L.k4o_free(p)
        return text

    def adaptive_trim(self, probe, targ, min_trim_len, max_mm, min_flank=3):
        """CSfxArray::AdaptiveTrim -> (return value, TrimSeqLen, TrimStart, TrimEnd, TrimMMs)"""
        probe = np.ascontiguousarray(probe, dtype=np.uint8)
        targ = np.ascontiguousarray(targ, dtype=np.uint8)
        o = (C.c_uint32 * 4)()
        vp = C.c_void_p
        r = self.L.k4o_adaptive_trim(C.c_uint32(len(probe)), vp(probe.ctypes.data), vp(targ.ctypes.data),
                                     C.c_uint32(min_trim_len), C.c_uint32(max_mm), C.c_uint32(min_flank),
                                     C.byref(o, 0), C.byref(o, 4), C.byref(o, 8), C.byref(o, 12))
        return (r, o[0], o[1], o[2], o[3])

    def auto_trim_flanks(self, h, reads, out, hits, seg2, min_flank_exacts, pe=False):
        """CKAligner::AutoTrimFlanks, in place; returns the number of reads eliminated"""
        cat, offs, lens = reads if isinstance(reads, tuple) else flatten_reads(reads)
        vp = C.c_void_p
        self.L.k4o_auto_trim_flanks.restype = C.c_int64
        return self.L.k4o_auto_trim_flanks(vp(h), C.c_int(min_flank_exacts), C.c_int(1 if pe else 0), C.c_int64(len(lens)),
                                           vp(cat.ctypes.data), vp(offs.ctypes.data), vp(lens.ctypes.data),
                                           C.c_int(hits.shape[1]), vp(out.ctypes.data), vp(hits.ctypes.data),
                                           vp(seg2.ctypes.data))

    def remove_orphan_juncts(self, which, out, hits, seg2):
        """CKAligner::RemoveOrphanSpliceJuncts (which=EXT_SPLICE) / RemoveOrphanMicroInDels (EXT_INDEL), in place"""
        vp = C.c_void_p
        self.L.k4o_remove_orphan_juncts.restype = C.c_int64
        return self.L.k4o_remove_orphan_juncts(C.c_uint32(which), C.c_int64(len(out)), C.c_int(hits.shape[1]),
                                               vp(out.ctypes.data), vp(hits.ctypes.data), vp(seg2.ctypes.data))

    def assign_multi_matches(self, out, hits, ml_mode, max_reads_len, threads=4):
        """CKAligner::AssignMultiMatches (`-r3` / `-r4`) over kalign_batch(pe_mode=1) results, in place."""
        n, max_ml = hits.shape
        self.L.k4o_assign_multi_matches.restype = C.c_int64
        return self.L.k4o_assign_multi_matches(C.c_int(ml_mode), C.c_int(max_reads_len), C.c_int64(n), C.c_int(max_ml),
                                               C.c_void_p(out.ctypes.data), C.c_void_p(hits.ctypes.data), C.c_int(threads))


def _kalign_pe(L, fn, h, reads1, reads2, pe_mode, pair_min_len, pair_max_len, pair_strand, threads, **kw):
    c1, o1, l1 = reads1 if isinstance(reads1, tuple) else flatten_reads(reads1)
    c2, o2, l2 = reads2 if isinstance(reads2, tuple) else flatten_reads(reads2)
    assert len(l1) == len(l2)
    kp = KalignParams(kw.get("max_subs", 5), kw.get("min_edit_dist", 1), kw.get("max_ns", 1), kw.get("pmode", 0),
                      kw.get("strand", STRAND_BOTH), 10, 1, kw.get("min_core_len", 0), kw.get("max_num_slides", 0),
                      kw.get("min_chimeric_len", 0), 0, 0)
    pe = PeParams(pe_mode, pair_min_len, pair_max_len, 1 if pair_strand else 0)
    out = np.zeros(2 * len(l1), dtype=PE_READ_DTYPE)
    rc = fn(h, C.byref(kp), C.byref(pe), len(l1), c1.ctypes.data, o1.ctypes.data, l1.ctypes.data, c2.ctypes.data,
            o2.ctypes.data, l2.ctypes.data, out.ctypes.data, threads)
    if rc != 0:
        raise RuntimeError("kalign_pe_batch failed: %d" % rc)
    return out


def oracle_kalign_pe(O, h, reads1, reads2, pe_mode=2, pair_min_len=100, pair_max_len=1000, pair_strand=False, threads=4,
                     **kw):
    """CKAligner PE flow (ProcCoredApprox + ProcessPairedEnds) on the CPU oracle; out[2i] = PE1, out[2i+1] = PE2."""
    return _kalign_pe(O.L, O.L.k4o_kalign_pe_batch, h, reads1, reads2, pe_mode, pair_min_len, pair_max_len, pair_strand,
                      threads, **kw)


class _RefXHit(C.Structure):
    _fields_ = [("chrom_id", C.c_uint32), ("match_loci", C.c_uint64), ("match_len", C.c_uint16), ("strand", C.c_uint8),
                ("mismatches", C.c_uint8), ("trim_left", C.c_uint16), ("trim_right", C.c_uint16), ("flags", C.c_uint8),
                ("seg1_mismatches", C.c_uint8), ("score", C.c_uint16), ("seg1_chrom_id", C.c_uint32),
                ("seg1_match_loci", C.c_uint64), ("seg1_match_len", C.c_uint16), ("seg1_read_ofs", C.c_uint16)]


class _RefHit(C.Structure):
    _fields_ = [("chrom_id", C.c_uint32), ("match_loci", C.c_uint64), ("match_len", C.c_uint16),
                ("strand", C.c_uint8), ("mismatches", C.c_uint8)]


def ref_available():
    return os.path.exists(REF_SO)


class Ref:
    """oracle/_ref/libk4ref.so -- the real reference (CSfxArray) behind oracle/ref_harness.cpp."""

    def __init__(self):
        L = C.CDLL(REF_SO)
        L.k4ref_open.restype = C.c_void_p
        L.k4ref_open.argtypes = [C.c_char_p, C.c_int, C.c_int]
        L.k4ref_close.argtypes = [C.c_void_p]
        L.k4ref_tot_seqs_len.restype = C.c_uint64
        L.k4ref_tot_seqs_len.argtypes = [C.c_void_p]
        L.k4ref_build_sfx.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                      C.c_int]
        L.k4ref_align_reads.argtypes = [C.c_void_p] + [C.c_int] * 7 + [C.c_void_p, C.c_int, C.c_int,
                                                                        C.POINTER(C.c_int), C.POINTER(C.c_int),
                                                                        C.POINTER(C.c_int), C.POINTER(_RefHit)]
        self.L = L

    def build_sfx(self, path, names, seqs, dataset="syn", threads=4):
        n, names_a, ptrs, lens, keep = _seq_args(names, seqs)
        r = self.L.k4ref_build_sfx(path.encode(), dataset.encode(), n, names_a, ptrs, lens, threads, 0)
        if r < 0:
            raise RuntimeError("k4ref_build_sfx failed %d" % r)

    def open(self, path, max_iter=5000, core_kmer_len=0):
        h = self.L.k4ref_open(path.encode(), max_iter, core_kmer_len)  # costs a fixed 2 s sleep (SfxArray.cpp:1165)
        if not h:
            raise RuntimeError("k4ref_open failed")
        return h

    def close(self, h):
        self.L.k4ref_close(h)

    def align_reads_batch(self, h, reads, tot_mm, core_len, core_delta, max_slides, min_core_len=0, mm_delta=1,
                          strand=STRAND_BOTH, max_hits=1):
        if isinstance(reads, tuple):
            cat, offs, lens = reads
            reads = [cat[int(o):int(o) + int(l)] for o, l in zip(offs, lens)]
        n = len(reads)
        rslt = np.zeros(n, np.int32); inst = np.zeros(n, np.int32); low = np.zeros(n, np.int32)
        nxt = np.zeros(n, np.int32)
        hits = np.zeros((n, max_hits), dtype=HIT_DTYPE)
        rh = (_RefHit * max_hits)()
        for i, rd in enumerate(reads):
            buf = np.ascontiguousarray(rd, dtype=np.uint8).copy()
            a, b, c = C.c_int(0), C.c_int(0), C.c_int(0)
            rslt[i] = self.L.k4ref_align_reads(h, tot_mm, core_len, core_delta, max_slides, min_core_len, mm_delta,
                                               strand, buf.ctypes.data, len(buf), max_hits, C.byref(a), C.byref(b),
                                               C.byref(c), rh)
            assert np.array_equal(buf, np.asarray(rd, dtype=np.uint8)), "reference did not restore the probe"
            inst[i], low[i], nxt[i] = a.value, b.value, c.value
            for k in range(max_hits):
                hits[i, k] = (rh[k].chrom_id, rh[k].match_loci & 0xFFFFFFFF, rh[k].match_len, rh[k].strand,
                              rh[k].mismatches, 0)
        return dict(rslt=rslt, inst=inst, low=low, nxt=nxt, hits=hits)

    def align_reads_ext_batch(self, h, reads, tot_mm, core_len, core_delta, max_slides, min_core_len=0, mm_delta=1,
                              strand=STRAND_BOTH, max_hits=1, min_chimeric_len=0, micro_indel_len=0,
                              max_splice_junct_len=0):
        """the real CSfxArray::AlignReads with every argument; same record layout as Oracle.align_reads_ext_batch
        (slots the caller of AlignReads would not look at are zeroed)"""
        if isinstance(reads, tuple):
            cat, offs, lens = reads
            reads = [cat[int(o):int(o) + int(l)] for o, l in zip(offs, lens)]
        n = len(reads)
        rslt = np.zeros(n, np.int32); inst = np.zeros(n, np.int32); low = np.zeros(n, np.int32)
        nxt = np.zeros(n, np.int32)
        hits = np.zeros((n, max_hits), dtype=HIT_DTYPE)
        seg2 = np.zeros(n, dtype=SEG2_DTYPE)
        rh = (_RefXHit * max_hits)()
        fn = self.L.k4ref_align_reads_ext
        fn.argtypes = [C.c_void_p] + [C.c_int] * 10 + [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                                       C.POINTER(C.c_int), C.POINTER(_RefXHit)]
        for i, rd in enumerate(reads):
            buf = np.ascontiguousarray(rd, dtype=np.uint8).copy()
            a, b, c = C.c_int(0), C.c_int(0), C.c_int(0)
            rslt[i] = fn(h, min_chimeric_len, micro_indel_len, max_splice_junct_len, tot_mm, core_len, core_delta, max_slides,
                         min_core_len, mm_delta, strand, buf.ctypes.data, len(buf), max_hits, C.byref(a), C.byref(b), C.byref(c), rh)
            assert np.array_equal(buf, np.asarray(rd, dtype=np.uint8)), "reference did not restore the probe"
            inst[i], low[i], nxt[i] = a.value, b.value, c.value
            if rslt[i] not in (HR_HITS, HR_MMDELTA, HR_HITINSTS):
                continue
            for k in range(min(int(inst[i]), max_hits)):
                x = rh[k]
                fl = x.flags
                ext = (x.trim_left & 0xFFF) | ((x.trim_right & 0xFFF) << 12) | ((fl & 1) << 24) | (((fl >> 1) & 1) << 25) | \
                    (((fl >> 2) & 1) << 26) | (((fl >> 3) & 1) << 27) | (((fl >> 4) & 1) << 28)
                hits[i, k] = (x.chrom_id, x.match_loci & 0xFFFFFFFF, x.match_len, x.strand, x.mismatches, ext)
                if k == 0 and (fl & 0x0A):
                    seg2[i] = (x.seg1_chrom_id, x.seg1_match_loci & 0xFFFFFFFF, x.seg1_match_len, x.seg1_read_ofs,
                               x.seg1_mismatches, 0, x.score)
        return dict(rslt=rslt, inst=inst, low=low, nxt=nxt, hits=hits, seg2=seg2)

    def align_paired_read_x(self, h, b3prime, antisense, chrom_id, start_loci, end_loci, min_insert, max_insert, max_allowed_mm,
                            read, min_chimeric_len, core_len, core_delta, max_slides=10, min_hamming=1):
        """the real CSfxArray::AlignPairedRead; same record as Oracle.align_paired_read_x"""
        self.L.k4ref_align_paired_read_x.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32] + [C.c_int] * 9 + \
                                                    [C.c_void_p, C.POINTER(_RefXHit)]
        rd = np.ascontiguousarray(read, dtype=np.uint8).copy()
        x = _RefXHit()
        r = self.L.k4ref_align_paired_read_x(h, int(b3prime), int(antisense), chrom_id, start_loci, end_loci, min_insert, max_insert,
                                             max_allowed_mm, min_hamming, len(rd), min_chimeric_len, core_len, core_delta, max_slides,
                                             rd.ctypes.data, C.byref(x))
        hit = np.zeros(1, dtype=HIT_DTYPE)
        ext = (x.trim_left & 0xFFF) | ((x.trim_right & 0xFFF) << 12) | ((1 << 24) if x.flags & 1 else 0)
        hit[0] = (x.chrom_id, x.match_loci & 0xFFFFFFFF, x.match_len, x.strand, x.mismatches, ext)
        return r, hit[0]

    def adaptive_trim(self, h, probe, targ, min_trim_len, max_mm, min_flank=3):
        probe = np.ascontiguousarray(probe, dtype=np.uint8).copy()
        targ = np.ascontiguousarray(targ, dtype=np.uint8).copy()
        o = (C.c_uint32 * 4)()
        fn = self.L.k4ref_adaptive_trim
        fn.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        r = fn(h, len(probe), probe.ctypes.data, targ.ctypes.data, min_trim_len, max_mm, min_flank, o)
        return (r, o[0], o[1], o[2], o[3])
