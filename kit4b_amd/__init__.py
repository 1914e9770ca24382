"""kit4b_amd -- MI355X-native kit4b kalign hot path (seed lookup + mismatch-bounded extension).

This package is a thin ctypes binding over the C ABI of ``kit4b_amd/libk4sfx.so`` (include/k4sfx.h), which holds
the hand-written HIP kernels for gfx950.  It exists for the tests and bench.py; the product boundary is the C ABI
and the C++ ``CSfxArray`` facade in include/k4_sfxarray.hpp.  There is no CPU fallback: importing works anywhere,
every compute call needs the library and a GPU and raises ``K4Error`` otherwise.
"""
import ctypes as C
import os
import subprocess

import numpy as np

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG_DIR, os.environ.get("K4SFX_LIB_NAME", "libk4sfx.so"))  # (the override: development builds, tools/slow_prof.py)

STRAND_BOTH, STRAND_WATSON, STRAND_CRICK = 0, 1, 2
HR_NONE, HR_HITS, HR_MMDELTA, HR_HITINSTS, HR_RMMDELTA, HR_SEQERRS = 0, 1, 2, 3, 4, 5
NAR_ACCEPTED, NAR_NS, NAR_NOHIT, NAR_MMDELTA, NAR_MULTIALIGN = 1, 2, 3, 4, 5

HIT_DTYPE = np.dtype(
    [("chrom_id", "<u4"), ("match_loci", "<u4"), ("match_len", "<u2"), ("strand", "u1"), ("mismatches", "u1"),
     ("reserved", "<u4")]
)
RESULT_DTYPE = np.dtype(
    [("hit_rslt", "<i4"), ("inst", "<i4"), ("low_mm", "<i4"), ("nxt_mm", "<i4"), ("nar", "<i4"), ("num_hits", "<i4")]
)
# k4_hit.ext (the dtype's "reserved" word): TrimLeft | TrimRight << 12 | K4_EXT_*; k4_seg2: Seg[1] of a two-segment hit
EXT_CHIMERIC, EXT_INDEL, EXT_INSERT, EXT_SPLICE, EXT_NONORPHAN = 1 << 24, 1 << 25, 1 << 26, 1 << 27, 1 << 28
SEG2_DTYPE = np.dtype(
    [("chrom_id", "<u4"), ("match_loci", "<u4"), ("match_len", "<u2"), ("read_ofs", "<u2"), ("mismatches", "u1"),
     ("reserved", "u1"), ("score", "<u2")]
)
NAR_TRIM, NAR_SPLICEJCTN, NAR_MICROINDEL = 6, 7, 8


class K4Error(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("k4sfx error %d: %s" % (code, msg))
        self.code = code


class Entry(C.Structure):
    _fields_ = [("entry_id", C.c_uint32), ("fblock_id", C.c_uint32), ("name", C.c_char * 81),
                ("name_hash", C.c_uint16), ("seq_len", C.c_uint32), ("start_ofs", C.c_uint64),
                ("end_ofs", C.c_uint64)]


class Info(C.Structure):
    _fields_ = [("concat_len", C.c_uint64), ("tot_seqs_len", C.c_uint64), ("sfx_el_size", C.c_uint32),
                ("n_entries", C.c_uint32), ("kmer_k", C.c_uint32), ("n_exc_blocks", C.c_uint32),
                ("device_bytes", C.c_uint64), ("device", C.c_int32), ("max_iter", C.c_int32),
                ("dataset", C.c_char * 81)]


class AlignParams(C.Structure):
    _fields_ = [("tot_mm", C.c_int32), ("core_len", C.c_int32), ("core_delta", C.c_int32),
                ("max_core_slides", C.c_int32), ("min_core_len", C.c_int32), ("mm_delta", C.c_int32),
                ("strand", C.c_int32), ("max_hits", C.c_int32), ("min_chimeric_len", C.c_int32),
                ("micro_indel_len", C.c_int32), ("max_splice_junct_len", C.c_int32)]


class KalignParams(C.Structure):
    _fields_ = [("max_subs", C.c_int32), ("min_edit_dist", C.c_int32), ("max_ns", C.c_int32), ("pmode", C.c_int32),
                ("strand", C.c_int32), ("max_ml", C.c_int32), ("pe_mode", C.c_int32), ("min_core_len", C.c_int32),
                ("max_num_slides", C.c_int32), ("min_chimeric_len", C.c_int32), ("micro_indel_len", C.c_int32),
                ("max_splice_junct_len", C.c_int32)]


class PeParams(C.Structure):
    _fields_ = [("pe_mode", C.c_int32), ("pair_min_len", C.c_int32), ("pair_max_len", C.c_int32),
                ("pair_strand", C.c_int32)]


PE_READ_DTYPE = np.dtype(
    [("nar", "<i4"), ("num_hits", "<i4"), ("inst", "<i4"), ("low_mm", "<i4"), ("pe_aligned", "<i4"), ("rescued", "<i4"),
     ("hit", HIT_DTYPE)]
)


class ParseInfo(C.Structure):
    _fields_ = [("n_records", C.c_uint64), ("consumed", C.c_uint64), ("n_bases", C.c_uint64), ("max_len", C.c_uint32),
                ("format", C.c_uint32)]


class SamNames(C.Structure):
    _fields_ = [("d_text", C.c_void_p * 2), ("d_name_off", C.c_void_p * 2), ("d_name_len", C.c_void_p * 2)]


class SamStats(C.Structure):
    _fields_ = [("nar", C.c_uint64 * 20), ("plus", C.c_uint64), ("minus", C.c_uint64), ("n_lines", C.c_uint64)]


FASTA, FASTQ = 1, 2


class PipelineParams(C.Structure):
    _fields_ = [("paired", C.c_int32), ("kp", KalignParams), ("pe", PeParams), ("min_len", C.c_int32), ("max_len", C.c_int32),
                ("n_buffers", C.c_uint32), ("min_batch_units", C.c_uint32), ("chunk_bytes", C.c_uint64),
                ("expect_text_bytes", C.c_uint64 * 2)]


class PipelineView(C.Structure):
    _fields_ = [("n_units", C.c_int64), ("n_reads", C.c_int64), ("max_read_len", C.c_uint32), ("max_ml", C.c_int32),
                ("n_under", C.c_uint64), ("n_over", C.c_uint64), ("d_rr", C.c_void_p), ("d_hits", C.c_void_p),
                ("d_seg2", C.c_void_p), ("d_pe", C.c_void_p), ("d_reads", C.c_void_p), ("d_offs", C.c_void_p),
                ("d_lens", C.c_void_p), ("names", SamNames)]


class Counters(C.Structure):
    _fields_ = [("n_reads", C.c_uint64), ("n_lookup", C.c_uint64), ("n_probe", C.c_uint64), ("n_cand", C.c_uint64),
                ("n_slow", C.c_uint64), ("n_bases", C.c_uint64)]


_lib = None

# every symbol include/k4sfx.h declares
ABI_SYMBOLS = [
    "k4_open", "k4_open_async", "k4_open_wait", "k4_open_seconds", "k4_open_host", "k4_open_device", "k4_close", "k4_last_error", "k4_global_error", "k4_info",
    "k4_get_entry", "k4_get_ident", "k4_set_max_iter", "k4_set_fastq_quality", "k4_get_seq", "k4_write_sfx", "k4_build_sa_device",
    "k4_reserve", "k4_align_reads_batch", "k4_align_reads_batch_dev", "k4_kalign_batch", "k4_kalign_batch_dev",
    "k4_min_core_len", "k4_get_counters", "k4_reset_counters", "k4_abi_version", "k4_enable_kernel_timing",
    "k4_get_kernel_times", "k4_get_kernel_times_split", "k4_snp_csv_dev", "k4_snp_vcf_dev", "k4_snp_files_dev", "k4_snp_run_dev", "k4_free_host", "k4_format_bam_dev", "k4_format_sam_all_dev", "k4_pipeline_format_bam", "k4_pipeline_format_all", "k4_pipeline_format_bam_all", "k4_format_bam_all_dev", "k4_pipeline_set_trims", "k4_pipeline_set_sampling", "k4_unaligned_fasta_dev", "k4_prepare_reads_trim_dev", "k4_mate_rescue_batch", "k4_kalign_pe_batch", "k4_kalign_pe_batch_dev",
    "k4_parse_fastx_dev", "k4_prepare_reads_dev", "k4_format_sam_dev", "k4_free_device", "k4_alloc_device",
    "k4_copy_to_device", "k4_copy_to_host", "k4_upload_pageable", "k4_host_register", "k4_host_unregister", "k4_best_matches_batch", "k4_best_matches_batch_dev",
    "k4_get_sfx_header", "k4_set_description", "k4_select_hits_dev",
    "k4_assign_multi_dev", "k4_align_reads_ext_batch", "k4_align_reads_ext_batch_dev", "k4_kalign_ext_batch",
    "k4_kalign_ext_batch_dev", "k4_auto_trim_flanks_dev", "k4_remove_orphan_juncts_dev", "k4_format_sam_ext_dev",
    "k4_pipeline_open", "k4_pipeline_acquire", "k4_pipeline_submit", "k4_pipeline_submit_host", "k4_pipeline_wait_aligned",
    "k4_pipeline_format", "k4_pipeline_next_sam", "k4_pipeline_read_sam", "k4_pipeline_close", "k4_sfx_map", "k4_sfx_unmap",
    "k4_set_raw_header",
]


def build(verbose=False):
    """Compile libk4sfx.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    cmd = ["make", "-C", os.path.join(PKG_DIR, "csrc"), "-j4"]
    subprocess.check_call(cmd, stdout=None if verbose else subprocess.DEVNULL)


def lib():
    """The loaded C ABI; raises if the HIP extension has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise K4Error(-2, "%s is missing: run kit4b_amd.build() / make -C kit4b_amd/csrc" % LIB_PATH)
    # One HIP runtime per process: PyTorch bundles its own libamdhip64 (soname libamdhip64.so.7).  Loading torch
    # first makes libk4sfx.so's NEEDED libamdhip64.so.7 resolve to that already-loaded copy; the other order leaves
    # two runtimes in the process and the second one finds no device.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, u32, u64 = C.c_void_p, C.c_int, C.c_int64, C.c_uint32, C.c_uint64
    L.k4_open.argtypes = [C.c_char_p, i32, i32, C.POINTER(vp)]
    L.k4_open_async.argtypes = [C.c_char_p, i32, i32, C.POINTER(vp)]
    L.k4_open_wait.argtypes = [vp]
    L.k4_open_seconds.argtypes = [vp]
    L.k4_open_seconds.restype = C.c_double
    L.k4_open_host.argtypes = [u64, u32, vp, vp, u32, C.POINTER(Entry), C.c_char_p, i32, i32, C.POINTER(vp)]
    L.k4_open_device.argtypes = [u64, u32, vp, vp, i32, u32, C.POINTER(Entry), C.c_char_p, i32, i32, C.POINTER(vp)]
    L.k4_close.argtypes = [vp]
    L.k4_close.restype = None
    L.k4_last_error.argtypes = [vp]
    L.k4_last_error.restype = C.c_char_p
    L.k4_global_error.restype = C.c_char_p
    L.k4_info.argtypes = [vp, C.POINTER(Info)]
    L.k4_get_entry.argtypes = [vp, u32, C.POINTER(Entry)]
    L.k4_get_ident.argtypes = [vp, C.c_char_p]
    L.k4_set_max_iter.argtypes = [vp, i32]
    L.k4_set_fastq_quality.argtypes = [vp, i32]
    L.k4_get_seq.argtypes = [vp, u32, u32, vp, u32]
    L.k4_write_sfx.argtypes = [vp, C.c_char_p]
    L.k4_build_sa_device.argtypes = [u64, u32, vp, vp, i32]
    L.k4_reserve.argtypes = [vp, i64, C.c_int32, C.c_int32]
    L.k4_align_reads_batch.argtypes = [vp, C.POINTER(AlignParams), i64] + [vp] * 8
    L.k4_align_reads_batch_dev.argtypes = [vp, C.POINTER(AlignParams), i64, C.c_int32] + [vp] * 9
    L.k4_kalign_batch.argtypes = [vp, C.POINTER(KalignParams), i64] + [vp] * 5
    L.k4_best_matches_batch.argtypes = [vp, C.POINTER(AlignParams), i64] + [vp] * 6
    L.k4_best_matches_batch_dev.argtypes = [vp, C.POINTER(AlignParams), i64, C.c_int32] + [vp] * 7
    L.k4_kalign_batch_dev.argtypes = [vp, C.POINTER(KalignParams), i64, C.c_int32] + [vp] * 6
    L.k4_min_core_len.argtypes = [vp, i32, C.POINTER(C.c_int)]
    L.k4_get_counters.argtypes = [vp, C.POINTER(Counters)]
    L.k4_reset_counters.argtypes = [vp]
    L.k4_enable_kernel_timing.argtypes = [vp, i32]
    L.k4_get_kernel_times.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_int32)]
    L.k4_mate_rescue_batch.argtypes = [vp, i64, vp, vp, u64, vp, vp]
    L.k4_kalign_pe_batch.argtypes = [vp, C.POINTER(KalignParams), C.POINTER(PeParams), i64] + [vp] * 7
    L.k4_kalign_pe_batch_dev.argtypes = [vp, C.POINTER(KalignParams), C.POINTER(PeParams), i64, C.c_int32] + [vp] * 5
    L.k4_parse_fastx_dev.argtypes = [vp, vp, u64, u64, i32, i32, i64, vp, u64, vp, vp, vp, vp, C.POINTER(ParseInfo), vp]
    L.k4_prepare_reads_dev.argtypes = [vp, i32, i64, C.c_int32, C.c_int32, vp, vp, vp, vp, u64, vp, vp,
                                       C.POINTER(u64), C.POINTER(u64), C.POINTER(u32), vp]
    L.k4_assign_multi_dev.argtypes = [vp, i32, C.c_int32, i64, C.c_int32, vp, vp, C.POINTER(C.c_int64), vp]
    L.k4_select_hits_dev.argtypes = [vp, i64, C.c_int32, vp, vp, vp, vp]
    L.k4_format_sam_dev.argtypes = [vp, i32, i64, vp, vp, C.c_int32, vp, vp, vp, vp, C.POINTER(SamNames), C.POINTER(vp),
                                    C.POINTER(u64), C.POINTER(SamStats), vp, vp]
    L.k4_align_reads_ext_batch.argtypes = [vp, C.POINTER(AlignParams), i64] + [vp] * 9
    L.k4_align_reads_ext_batch_dev.argtypes = [vp, C.POINTER(AlignParams), i64, C.c_int32] + [vp] * 10
    L.k4_kalign_ext_batch.argtypes = [vp, C.POINTER(KalignParams), i64] + [vp] * 6
    L.k4_kalign_ext_batch_dev.argtypes = [vp, C.POINTER(KalignParams), i64, C.c_int32] + [vp] * 7
    L.k4_auto_trim_flanks_dev.argtypes = [vp, C.c_int32, i32, i64, C.c_int32, vp, vp, vp, vp, vp, C.POINTER(C.c_int64), vp]
    L.k4_remove_orphan_juncts_dev.argtypes = [vp, u32, i64, C.c_int32, vp, vp, vp, C.POINTER(C.c_int64), vp]
    L.k4_format_sam_ext_dev.argtypes = [vp, i32, i64, vp, vp, C.c_int32, vp, vp, vp, vp, vp, C.POINTER(SamNames), C.POINTER(vp),
                                        C.POINTER(u64), C.POINTER(SamStats), vp, vp]
    L.k4_format_bam_dev.argtypes = [vp, i32, i64, vp, vp, C.c_int32, vp, vp, vp, vp, vp, C.POINTER(SamNames), C.c_int32, C.POINTER(vp),
                                    C.POINTER(u64), C.POINTER(SamStats), vp, vp]
    L.k4_pipeline_format_bam.argtypes = [vp, C.c_int32, C.POINTER(SamStats), vp, C.POINTER(u64)]
    L.k4_pipeline_open.argtypes = [vp, C.POINTER(PipelineParams), C.POINTER(vp)]
    L.k4_pipeline_acquire.argtypes = [vp, i32, C.POINTER(vp), C.POINTER(u64)]
    L.k4_pipeline_submit.argtypes = [vp, i32, u64, i32]
    L.k4_pipeline_submit_host.argtypes = [vp, i32, vp, u64, i32]
    L.k4_pipeline_wait_aligned.argtypes = [vp, C.POINTER(PipelineView)]
    L.k4_pipeline_format.argtypes = [vp, C.POINTER(SamStats), vp, C.POINTER(u64)]
    L.k4_pipeline_next_sam.argtypes = [vp, C.POINTER(vp), C.POINTER(u64)]
    L.k4_pipeline_read_sam.argtypes = [vp, vp, u64, C.POINTER(u64)]
    L.k4_pipeline_close.argtypes = [vp]
    L.k4_pipeline_close.restype = None
    L.k4_free_device.argtypes = [vp]
    L.k4_free_device.restype = None
    L.k4_alloc_device.argtypes = [vp, u64, C.POINTER(vp)]
    L.k4_copy_to_device.argtypes = [vp, vp, vp, u64]
    L.k4_copy_to_host.argtypes = [vp, vp, vp, u64]
    L.k4_upload_pageable.argtypes = [i32, vp, vp, u64]
    L.k4_host_register.argtypes = [vp, u64]
    L.k4_host_unregister.argtypes = [vp]
    # (every argument list is declared: ctypes would otherwise pass a Python int -- a device address -- as a 32-bit C int)
    dbl, pvp, pu64 = C.c_double, C.POINTER(vp), C.POINTER(u64)
    fmt_head = [vp, i32, i64, vp, vp, C.c_int32, vp, vp, vp, vp, vp, C.POINTER(SamNames)]  # ix .. names of the *_all / *_ext formatters
    L.k4_format_sam_all_dev.argtypes = fmt_head + [pvp, pu64, C.POINTER(SamStats), vp, vp]
    L.k4_format_bam_all_dev.argtypes = fmt_head + [C.c_int32, pvp, pu64, C.POINTER(SamStats), vp, vp]
    L.k4_pipeline_format_all.argtypes = [vp, C.POINTER(SamStats), vp, pu64]
    L.k4_pipeline_format_bam_all.argtypes = [vp, C.c_int32, C.POINTER(SamStats), vp, pu64]
    L.k4_pipeline_set_trims.argtypes = [vp, C.c_int32, C.c_int32]
    L.k4_pipeline_set_sampling.argtypes = [vp, C.c_int32]
    L.k4_unaligned_fasta_dev.argtypes = [vp, i32, i64, vp, vp, vp, vp, vp, C.POINTER(SamNames), C.c_int32, pvp, pu64, pu64, vp]
    L.k4_prepare_reads_trim_dev.argtypes = [vp, i32, i64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, i64, vp, vp, vp, vp, u64,
                                            vp, vp, pu64, pu64, C.POINTER(u32), vp]
    snp_head = [vp, i32, i64, vp, vp, C.c_int32, vp, vp, vp, vp, C.c_int32, dbl, dbl]  # ix, pe .. snp_nonref_pcnt
    L.k4_snp_csv_dev.argtypes = snp_head + [pvp, pu64, pu64, vp]
    L.k4_snp_vcf_dev.argtypes = snp_head + [pvp, pu64, pu64, vp]
    L.k4_snp_files_dev.argtypes = [vp, i32] + snp_head[1:] + [pvp, pu64, pu64, pvp, pu64, vp]
    L.k4_snp_run_dev.argtypes = [vp, i32] + snp_head[1:] + [vp, vp]
    L.k4_free_host.argtypes = [vp]
    L.k4_free_host.restype = None
    L.k4_sfx_map.argtypes = [C.c_char_p, vp]
    L.k4_sfx_unmap.argtypes = [vp]
    L.k4_sfx_unmap.restype = None
    L.k4_set_raw_header.argtypes = [vp, vp]
    L.k4_set_description.argtypes = [vp, C.c_char_p, C.c_char_p]
    L.k4_get_sfx_header.argtypes = [vp, vp]
    L.k4_abi_version.argtypes = []
    L.k4_global_error.argtypes = []
    L.k4_get_kernel_times_split.argtypes = [vp, C.POINTER(dbl), C.POINTER(dbl), C.POINTER(C.c_int32)]
    _lib = L
    return L


def _flatten(reads):
    if isinstance(reads, tuple):
        cat, offs, lens = reads
        return (np.ascontiguousarray(cat, dtype=np.uint8), np.ascontiguousarray(offs, dtype=np.uint64),
                np.ascontiguousarray(lens, dtype=np.uint32))
    lens = np.array([len(r) for r in reads], dtype=np.uint32)
    offs = np.zeros(len(reads), dtype=np.uint64)
    if len(reads):
        offs[1:] = np.cumsum(lens[:-1], dtype=np.uint64)
        cat = np.ascontiguousarray(np.concatenate(reads).astype(np.uint8))
    else:
        cat = np.zeros(0, dtype=np.uint8)
    return cat, offs, lens


def gen_hash16(name):
    """CUtility::GenHash16 (libkit4b/Utility.cpp:402-420): the 16-bit hash of the lower-cased entry name"""
    if not name:
        return 0
    h = 19937
    for ch in name.lower():
        h = ((h ^ ch) * 3119) & 0xFFFFFFFF
        if h & 0x80000000:  # int arithmetic in the reference: the shift below is arithmetic
            h -= 1 << 32
        h ^= h >> 13
        h &= 0xFFFF
    return h or 19937


def make_entries(names, lens):
    """tsSfxEntry table for sequences concatenated with one EOS after each (CSfxArray::AddEntry, SfxArray.cpp:1735-1750)."""
    arr = (Entry * len(names))()
    ofs = 0
    for i, (nm, ln) in enumerate(zip(names, lens)):
        arr[i].entry_id = i + 1
        arr[i].fblock_id = 1
        arr[i].name = nm.encode() if isinstance(nm, str) else nm
        arr[i].name_hash = gen_hash16(arr[i].name)
        arr[i].seq_len = int(ln)
        arr[i].start_ofs = ofs
        arr[i].end_ofs = ofs + int(ln) - 1
        ofs += int(ln) + 1
    return arr


class SfxIndex:
    """HBM-resident index: the counterpart of an opened CSfxArray (libkit4b/SfxArray.h:524)."""

    def __init__(self, handle):
        self.h = C.c_void_p(handle)
        self._keep = []

    # -- construction -----------------------------------------------------------------------------------------
    @staticmethod
    def _check_open(rc, h):
        if rc != 0:
            raise K4Error(rc, lib().k4_global_error().decode())
        return SfxIndex(h.value)

    @classmethod
    def open(cls, path, device=0, kmer_k=0):
        h = C.c_void_p()
        return cls._check_open(lib().k4_open(path.encode(), device, kmer_k, C.byref(h)), h)

    @classmethod
    def from_host(cls, seq, sa_bytes, el_size, entries, dataset="syn", device=0, kmer_k=0):
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        sa_bytes = np.ascontiguousarray(sa_bytes, dtype=np.uint8)
        h = C.c_void_p()
        rc = lib().k4_open_host(len(seq), el_size, seq.ctypes.data, sa_bytes.ctypes.data, len(entries), entries,
                                dataset.encode(), device, kmer_k, C.byref(h))
        return cls._check_open(rc, h)

    @classmethod
    def from_device(cls, concat_len, el_size, d_seq_ptr, d_sa_ptr, entries, dataset="syn", device=0, kmer_k=0,
                    adopt_sa=True, keep=()):
        h = C.c_void_p()
        rc = lib().k4_open_device(concat_len, el_size, d_seq_ptr, d_sa_ptr, 1 if adopt_sa else 0, len(entries),
                                  entries, dataset.encode(), device, kmer_k, C.byref(h))
        ix = cls._check_open(rc, h)
        ix._keep = list(keep)  # tensors whose storage the index adopted
        return ix

    def close(self):
        if self.h:
            lib().k4_close(self.h)
            self.h = C.c_void_p()

    def _ck(self, rc):
        if rc < 0:
            raise K4Error(rc, lib().k4_last_error(self.h).decode())
        return rc

    # -- accessors ----------------------------------------------------------------------------------------------
    def info(self):
        o = Info()
        self._ck(lib().k4_info(self.h, C.byref(o)))
        return {f[0]: (getattr(o, f[0]).decode() if f[0] == "dataset" else getattr(o, f[0])) for f in Info._fields_}

    def entry(self, entry_id):
        e = Entry()
        self._ck(lib().k4_get_entry(self.h, entry_id, C.byref(e)))
        return dict(entry_id=e.entry_id, name=e.name.decode(), seq_len=e.seq_len, start_ofs=e.start_ofs,
                    end_ofs=e.end_ofs)

    def get_ident(self, name):
        return lib().k4_get_ident(self.h, name.encode())

    def set_max_iter(self, it):
        return lib().k4_set_max_iter(self.h, it)

    def set_fastq_quality(self, method):
        """kalign -g: 0 Sanger, 1 Illumina 1.3+, 2 Solexa, 3 ignore the quality lines (default)"""
        self._ck(lib().k4_set_fastq_quality(self.h, method))

    def get_seq(self, entry_id, loci, length):
        out = np.zeros(length, dtype=np.uint8)
        n = lib().k4_get_seq(self.h, entry_id, loci, out.ctypes.data, length)
        return out[:n]

    def write_sfx(self, path):
        self._ck(lib().k4_write_sfx(self.h, path.encode()))

    def min_core_len(self, pmode=0):
        s = C.c_int(0)
        m = self._ck(lib().k4_min_core_len(self.h, pmode, C.byref(s)))
        return m, s.value

    def counters(self):
        c = Counters()
        self._ck(lib().k4_get_counters(self.h, C.byref(c)))
        return {f[0]: getattr(c, f[0]) for f in Counters._fields_}

    def reset_counters(self):
        self._ck(lib().k4_reset_counters(self.h))

    def enable_kernel_timing(self, on=True):
        self._ck(lib().k4_enable_kernel_timing(self.h, 1 if on else 0))

    def kernel_times(self):
        """(summed k4k_align_step milliseconds, launches) since the last call; synchronises."""
        ms, n = C.c_double(0), C.c_int32(0)
        self._ck(lib().k4_get_kernel_times(self.h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def kernel_times_split(self):
        """(k4k_align_step ms, k4k_align_slow ms, batches) since the last call; synchronises."""
        ms, g, n = C.c_double(0), C.c_double(0), C.c_int32(0)
        self._ck(lib().k4_get_kernel_times_split(self.h, C.byref(ms), C.byref(g), C.byref(n)))
        return ms.value, g.value, n.value

    def reserve(self, max_reads, max_read_len, max_hits):
        self._ck(lib().k4_reserve(self.h, max_reads, max_read_len, max_hits))

    # -- the hot path (host buffers) ----------------------------------------------------------------------------
    def align_reads_batch(self, reads, tot_mm, core_len, core_delta, max_slides, min_core_len=0, mm_delta=1,
                          strand=STRAND_BOTH, max_hits=1):
        """CSfxArray::AlignReads (libkit4b/SfxArray.h:614) for a batch of fresh reads."""
        cat, offs, lens = _flatten(reads)
        n = len(lens)
        p = AlignParams(tot_mm, core_len, core_delta, max_slides, min_core_len, mm_delta, strand, max_hits)
        rslt = np.zeros(n, np.int32); inst = np.zeros(n, np.int32); low = np.zeros(n, np.int32)
        nxt = np.zeros(n, np.int32)
        hits = np.zeros((n, max_hits), dtype=HIT_DTYPE)
        self._ck(lib().k4_align_reads_batch(self.h, C.byref(p), n, cat.ctypes.data, offs.ctypes.data,
                                            lens.ctypes.data, rslt.ctypes.data, inst.ctypes.data, low.ctypes.data,
                                            nxt.ctypes.data, hits.ctypes.data))
        return dict(rslt=rslt, inst=inst, low=low, nxt=nxt, hits=hits)

    def align_reads_ext_batch(self, reads, tot_mm, core_len, core_delta, max_slides, min_core_len=0, mm_delta=1,
                              strand=STRAND_BOTH, max_hits=1, min_chimeric_len=0, micro_indel_len=0, max_splice_junct_len=0):
        """CSfxArray::AlignReads with MinChimericLen / microInDelLen / MaxSpliceJunctLen (SfxArray.cpp:7894-7930)."""
        cat, offs, lens = _flatten(reads)
        n = len(lens)
        p = AlignParams(tot_mm, core_len, core_delta, max_slides, min_core_len, mm_delta, strand, max_hits,
                        min_chimeric_len, micro_indel_len, max_splice_junct_len)
        rslt = np.zeros(n, np.int32); inst = np.zeros(n, np.int32); low = np.zeros(n, np.int32)
        nxt = np.zeros(n, np.int32)
        hits = np.zeros((n, max_hits), dtype=HIT_DTYPE)
        seg2 = np.zeros(n, dtype=SEG2_DTYPE)
        self._ck(lib().k4_align_reads_ext_batch(self.h, C.byref(p), n, cat.ctypes.data, offs.ctypes.data, lens.ctypes.data,
                                                rslt.ctypes.data, inst.ctypes.data, low.ctypes.data, nxt.ctypes.data,
                                                hits.ctypes.data, seg2.ctypes.data))
        return dict(rslt=rslt, inst=inst, low=low, nxt=nxt, hits=hits, seg2=seg2)

    def kalign_ext_batch(self, reads, max_subs=5, min_edit_dist=1, max_ns=1, pmode=0, strand=STRAND_BOTH, max_ml=1,
                         pe_mode=0, min_core_len=0, max_num_slides=0, min_chimeric_len=0, micro_indel_len=0,
                         max_splice_junct_len=0):
        """CKAligner::AlignRead with -c / -a / -A."""
        cat, offs, lens = _flatten(reads)
        n = len(lens)
        p = KalignParams(max_subs, min_edit_dist, max_ns, pmode, strand, max_ml, pe_mode, min_core_len, max_num_slides,
                         min_chimeric_len, micro_indel_len, max_splice_junct_len)
        out = np.zeros(n, dtype=RESULT_DTYPE)
        hits = np.zeros((n, max(1, max_ml)), dtype=HIT_DTYPE)
        seg2 = np.zeros(n, dtype=SEG2_DTYPE)
        self._ck(lib().k4_kalign_ext_batch(self.h, C.byref(p), n, cat.ctypes.data, offs.ctypes.data, lens.ctypes.data,
                                           out.ctypes.data, hits.ctypes.data, seg2.ctypes.data))
        return dict(out=out, hits=hits, seg2=seg2)

    def best_matches_batch(self, reads, tot_mm, core_len, core_delta, max_core_slides, strand=STRAND_BOTH, max_hits=5):
        """CSfxArray::LocateBestMatches over a batch: rslt (the call's return value), inst, hits[n, max_hits]."""
        cat, offs, lens = _flatten(reads)
        n = len(lens)
        p = AlignParams(tot_mm, core_len, core_delta, max_core_slides, 0, 1, strand, max_hits)
        rslt = np.zeros(n, dtype=np.int32)
        inst = np.zeros(n, dtype=np.int32)
        hits = np.zeros((n, max_hits), dtype=HIT_DTYPE)
        self._ck(lib().k4_best_matches_batch(self.h, C.byref(p), n, cat.ctypes.data, offs.ctypes.data, lens.ctypes.data,
                                             rslt.ctypes.data, inst.ctypes.data, hits.ctypes.data))
        return dict(rslt=rslt, inst=inst, hits=hits)

    def kalign_batch(self, reads, max_subs=5, min_edit_dist=1, max_ns=1, pmode=0, strand=STRAND_BOTH, max_ml=1,
                     pe_mode=0, min_core_len=0, max_num_slides=0):
        """CKAligner::AlignRead (ngskit4b/KAligner.cpp:9583) for a batch."""
        cat, offs, lens = _flatten(reads)
        n = len(lens)
        p = KalignParams(max_subs, min_edit_dist, max_ns, pmode, strand, max_ml, pe_mode, min_core_len,
                         max_num_slides)
        out = np.zeros(n, dtype=RESULT_DTYPE)
        hits = np.zeros((n, max(1, max_ml)), dtype=HIT_DTYPE)
        self._ck(lib().k4_kalign_batch(self.h, C.byref(p), n, cat.ctypes.data, offs.ctypes.data, lens.ctypes.data,
                                       out.ctypes.data, hits.ctypes.data))
        return dict(out=out, hits=hits)

    def kalign_pe_batch(self, reads1, reads2, pe_mode=2, pair_min_len=100, pair_max_len=1000, pair_strand=False,
                        max_subs=5, min_edit_dist=1, max_ns=1, pmode=0, strand=STRAND_BOTH, min_core_len=0,
                        max_num_slides=0, min_chimeric_len=0):
        """CKAligner's PE flow (ProcCoredApprox + ProcessPairedEnds); out[2i] = PE1, out[2i+1] = PE2.  min_chimeric_len
        (kalign -c): chimeric trimming of both ends and of rescued mates; the trims come back in hit.ext."""
        c1, o1, l1 = _flatten(reads1)
        c2, o2, l2 = _flatten(reads2)
        assert len(l1) == len(l2)
        kp = KalignParams(max_subs, min_edit_dist, max_ns, pmode, strand, 10, 1, min_core_len, max_num_slides, min_chimeric_len, 0, 0)
        pe = PeParams(pe_mode, pair_min_len, pair_max_len, 1 if pair_strand else 0)
        out = np.zeros(2 * len(l1), dtype=PE_READ_DTYPE)
        self._ck(lib().k4_kalign_pe_batch(self.h, C.byref(kp), C.byref(pe), len(l1), c1.ctypes.data, o1.ctypes.data,
                                          l1.ctypes.data, c2.ctypes.data, o2.ctypes.data, l2.ctypes.data,
                                          out.ctypes.data))
        return out

    # -- read ingest and SAM emit on the device ----------------------------------------------------------------------
    def parse_fastx(self, text, fmt=0, chunk_bytes=None, device=None):
        """FASTA / FASTQ bytes -> dict of device tensors (reads u8, offs i64, lens i32, name_off i64, name_len i32) plus
        the text tensor they point into.  chunk_bytes < len(text) exercises the chunked protocol."""
        import torch

        dev = torch.device("cuda", self.info()["device"]) if device is None else device
        raw = np.frombuffer(bytes(text), dtype=np.uint8)
        T = len(raw)
        d_text = torch.from_numpy(np.concatenate([raw, np.zeros(16, np.uint8)])).to(dev)
        cap = int((raw == 10).sum()) // 2 + 4
        d_reads = torch.zeros(T + 32, dtype=torch.uint8, device=dev)
        d_offs = torch.zeros(cap, dtype=torch.int64, device=dev)
        d_lens = torch.zeros(cap, dtype=torch.int32, device=dev)
        d_noff = torch.zeros(cap, dtype=torch.int64, device=dev)
        d_nlen = torch.zeros(cap, dtype=torch.int32, device=dev)
        st = torch.cuda.current_stream().cuda_stream
        pos, n, bases, max_len = 0, 0, 0, 0
        chunk = T if not chunk_bytes else chunk_bytes
        while pos < T:
            ln = min(chunk, T - pos)
            final = 1 if pos + ln == T else 0
            info = ParseInfo()
            self._ck(lib().k4_parse_fastx_dev(self.h, d_text.data_ptr() + pos, ln, pos, final, fmt, cap - n, d_reads.data_ptr(),
                                              bases, d_offs.data_ptr() + 8 * n, d_lens.data_ptr() + 4 * n,
                                              d_noff.data_ptr() + 8 * n, d_nlen.data_ptr() + 4 * n, C.byref(info), st))
            fmt = info.format or fmt
            if info.consumed == 0:
                if final:
                    break
                chunk *= 2  # a record longer than the chunk
                continue
            n += info.n_records
            bases += info.n_bases
            max_len = max(max_len, info.max_len)
            pos += info.consumed
        return {"n": n, "n_bases": bases, "max_len": max_len, "format": fmt, "text": d_text, "reads": d_reads,
                "offs": d_offs[:n], "lens": d_lens[:n], "name_off": d_noff[:n], "name_len": d_nlen[:n]}

    def prepare_reads(self, p1, p2=None, min_len=50, max_len=500):
        """length filter (+ PE interleave) over parse_fastx results; PE needs both parsed into ONE reads buffer, so
        p2's reads are appended behind p1's here."""
        import torch

        pe = p2 is not None
        n = p1["n"]
        dev = p1["reads"].device
        if pe:
            assert p2["n"] == n
            reads = torch.cat([p1["reads"][: p1["n_bases"]], p2["reads"][: p2["n_bases"] + 16]])
        else:
            reads = p1["reads"]
        tot = 2 * n if pe else n
        d_offs = torch.zeros(max(tot, 1), dtype=torch.int64, device=dev)
        d_lens = torch.zeros(max(tot, 1), dtype=torch.int32, device=dev)
        under, over, ml = C.c_uint64(), C.c_uint64(), C.c_uint32()
        self._ck(lib().k4_prepare_reads_dev(self.h, 1 if pe else 0, n, min_len, max_len, p1["offs"].data_ptr(),
                                            p1["lens"].data_ptr(), p2["offs"].data_ptr() if pe else None,
                                            p2["lens"].data_ptr() if pe else None, p1["n_bases"] if pe else 0,
                                            d_offs.data_ptr(), d_lens.data_ptr(), C.byref(under), C.byref(over), C.byref(ml),
                                            torch.cuda.current_stream().cuda_stream))
        return {"reads": reads, "offs": d_offs[:tot], "lens": d_lens[:tot], "n_under": under.value, "n_over": over.value,
                "max_len": ml.value, "n_units": n, "pe": pe}

    def format_sam(self, prep, p1, p2=None, rr=None, hits=None, max_ml=1, pe_recs=None, seg2=None, bam=False, sq_all=True, all_reads=False):
        """SAM body (bytes), stats dict and per-chromosome hit flags for device-resident results.  bam=True: the same
        alignments as uncompressed BAM records (k4_format_bam_dev), refIDs for a header of all (sq_all) / the hit sequences;
        all_reads=True: kalign -M1, the reads that were not accepted follow as unaligned records (k4_format_sam_all_dev)."""
        import torch

        names = SamNames()
        for w, p in enumerate([p1, p2]):
            if p is not None:
                names.d_text[w] = p["text"].data_ptr()
                names.d_name_off[w] = p["name_off"].data_ptr()
                names.d_name_len[w] = p["name_len"].data_ptr()
        d_sam, nbytes, stats = C.c_void_p(), C.c_uint64(), SamStats()
        ne = self.info()["n_entries"]
        chrom_hit = np.zeros(ne + 1, dtype=np.uint8)
        head = (self.h, 1 if prep["pe"] else 0, prep["n_units"], rr.data_ptr() if rr is not None else None,
                hits.data_ptr() if hits is not None else None, max_ml, pe_recs.data_ptr() if pe_recs is not None else None,
                seg2.data_ptr() if seg2 is not None else None, prep["reads"].data_ptr(), prep["offs"].data_ptr(),
                prep["lens"].data_ptr(), C.byref(names))
        tail = (C.byref(d_sam), C.byref(nbytes), C.byref(stats), chrom_hit.ctypes.data, torch.cuda.current_stream().cuda_stream)
        if bam:
            self._ck((lib().k4_format_bam_all_dev if all_reads else lib().k4_format_bam_dev)(*head, 1 if sq_all else 0, *tail))
        elif all_reads:
            self._ck(lib().k4_format_sam_all_dev(*head, *tail))
        else:
            self._ck(lib().k4_format_sam_ext_dev(*head, *tail))
        body = b""
        if nbytes.value:
            buf = np.empty(nbytes.value, dtype=np.uint8)
            self._ck(lib().k4_copy_to_host(self.h, buf.ctypes.data, d_sam, nbytes.value))
            body = buf.tobytes()
        lib().k4_free_device(d_sam)
        return body, {"nar": list(stats.nar), "plus": stats.plus, "minus": stats.minus, "n_lines": stats.n_lines}, chrom_hit

    def snp_csv(self, reads, out=None, hits=None, pe_recs=None, min_snp_reads=5, qvalue=0.05, snp_nonref_pcnt=25.0):
        """kalign's SNP calling (main CSV) over host-side results: SE (out + hits records) or PE (pe_recs, reads interleaved).
        Returns (CSV text, number of SNPs)."""
        import torch

        dev = torch.device("cuda", self.info()["device"])
        cat, offs, lens = _flatten(reads)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev)  # noqa: E731
        d_reads = torch.from_numpy(np.concatenate([cat, np.zeros(16, np.uint8)])).to(dev)
        d_offs, d_lens = t(offs), t(lens)
        L = lib()
        csv, nb, ns = C.c_void_p(), C.c_uint64(), C.c_uint64()
        if pe_recs is not None:
            d_pe = t(pe_recs)
            self._ck(L.k4_snp_csv_dev(self.h, 1, len(lens) // 2, None, None, 1, d_pe.data_ptr(), d_reads.data_ptr(), d_offs.data_ptr(),
                                      d_lens.data_ptr(), min_snp_reads, qvalue, snp_nonref_pcnt, C.byref(csv), C.byref(nb), C.byref(ns), 0))
        else:
            d_rr, d_hits = t(out), t(hits)
            max_ml = 1 if hits.ndim == 1 else hits.shape[1]
            self._ck(L.k4_snp_csv_dev(self.h, 0, len(lens), d_rr.data_ptr(), d_hits.data_ptr(), max_ml, None, d_reads.data_ptr(),
                                      d_offs.data_ptr(), d_lens.data_ptr(), min_snp_reads, qvalue, snp_nonref_pcnt, C.byref(csv), C.byref(nb),
                                      C.byref(ns), 0))
        text = C.string_at(csv.value, nb.value).decode()
        L.k4_free_host(csv)
        return text, ns.value

    def snp_files(self, reads, out=None, hits=None, pe_recs=None, min_snp_reads=5, qvalue=0.05, snp_nonref_pcnt=25.0, vcf=False):
        """every file of a kalign SNP run (k4_snp_run_dev) over host-side results, as a dict of texts: "snp" (CSV, or VCF), "wig"
        (.covsegs.wig), "disnp" (.disnp.csv), "trisnp" (.trisnp.csv), and "n_snps"."""
        import torch

        class SnpFiles(C.Structure):
            _fields_ = [("snp", C.c_void_p), ("snp_bytes", C.c_uint64), ("n_snps", C.c_uint64), ("wig", C.c_void_p), ("wig_bytes", C.c_uint64),
                        ("disnp", C.c_void_p), ("disnp_bytes", C.c_uint64), ("trisnp", C.c_void_p), ("trisnp_bytes", C.c_uint64)]

        dev = torch.device("cuda", self.info()["device"])
        cat, offs, lens = _flatten(reads)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev)  # noqa: E731
        d_reads = torch.from_numpy(np.concatenate([cat, np.zeros(16, np.uint8)])).to(dev)
        d_offs, d_lens = t(offs), t(lens)
        L = lib()
        f = SnpFiles()
        if pe_recs is not None:
            d_pe = t(pe_recs)
            self._ck(L.k4_snp_run_dev(self.h, 1 if vcf else 0, 1, len(lens) // 2, None, None, 1, d_pe.data_ptr(), d_reads.data_ptr(), d_offs.data_ptr(),
                                      d_lens.data_ptr(), min_snp_reads, qvalue, snp_nonref_pcnt, C.byref(f), 0))
        else:
            d_rr, d_hits = t(out), t(hits)
            max_ml = 1 if hits.ndim == 1 else hits.shape[1]
            self._ck(L.k4_snp_run_dev(self.h, 1 if vcf else 0, 0, len(lens), d_rr.data_ptr(), d_hits.data_ptr(), max_ml, None, d_reads.data_ptr(),
                                      d_offs.data_ptr(), d_lens.data_ptr(), min_snp_reads, qvalue, snp_nonref_pcnt, C.byref(f), 0))
        res = {"n_snps": f.n_snps}
        for k in ("snp", "wig", "disnp", "trisnp"):
            res[k] = C.string_at(getattr(f, k), getattr(f, k + "_bytes")).decode()
            L.k4_free_host(getattr(f, k))
        return res

    def post_stages(self, reads, out, hits, seg2, min_flank_exacts=0, orphan_splice=False, orphan_indel=False):
        """AutoTrimFlanks / RemoveOrphanSpliceJuncts / RemoveOrphanMicroInDels (KAligner.cpp:653-686) over host arrays of SE
        results, in the reference's order; returns (out, hits, counts)."""
        import torch

        dev = torch.device("cuda", self.info()["device"])
        cat, offs, lens = _flatten(reads)
        n, max_ml = hits.shape
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev)
        d_rr, d_hits, d_seg2 = t(out), t(hits), t(seg2)
        d_reads = torch.from_numpy(np.concatenate([cat, np.zeros(16, np.uint8)])).to(dev)
        d_offs, d_lens = t(offs), t(lens)
        cnt = {}
        c = C.c_int64(0)
        if min_flank_exacts > 0:
            self._ck(lib().k4_auto_trim_flanks_dev(self.h, min_flank_exacts, 0, n, max_ml, d_rr.data_ptr(), d_hits.data_ptr(),
                                                   d_reads.data_ptr(), d_offs.data_ptr(), d_lens.data_ptr(), C.byref(c), 0))
            cnt["trim"] = c.value
        for on, which, key in ((orphan_splice, EXT_SPLICE, "splice"), (orphan_indel, EXT_INDEL, "indel")):
            if on:
                self._ck(lib().k4_remove_orphan_juncts_dev(self.h, which, n, max_ml, d_rr.data_ptr(), d_hits.data_ptr(),
                                                           d_seg2.data_ptr(), C.byref(c), 0))
                cnt[key] = c.value
        return (d_rr.cpu().numpy().view(RESULT_DTYPE), d_hits.cpu().numpy().view(HIT_DTYPE).reshape(n, max_ml), cnt)

    def pipeline_sam(self, texts, kp, pe=None, min_len=50, max_len=500, chunk_bytes=0, ring=False, out=None, min_batch_units=0, all_reads=False, expect=True):
        """host text (bytes-like / pinned tensors: one for SE, two for PE) -> SAM body through the overlapped pipeline.
        ring=True feeds through acquire / submit (what k4align's reader threads do), else submit_host.  Returns (body or
        number of bytes written into `out`, stats dict, view)."""
        L = lib()
        prm = PipelineParams()
        prm.paired = 1 if len(texts) == 2 else 0
        prm.kp = kp
        if pe is not None:
            prm.pe = pe
        prm.min_len, prm.max_len, prm.chunk_bytes = min_len, max_len, chunk_bytes
        prm.min_batch_units = min_batch_units  # 0: the default (4 M); small values make several alignment batches of a small input
        bufs = []
        for e, t in enumerate(texts):
            if hasattr(t, "data_ptr"):
                bufs.append((t.data_ptr(), t.numel(), t))
            else:
                a = np.frombuffer(bytes(t), dtype=np.uint8)
                bufs.append((a.ctypes.data, len(a), a))
            prm.expect_text_bytes[e] = bufs[-1][1] if expect else 0  # (unknown sizes: the arrays grow by doubling)
        pl = C.c_void_p()
        self._ck(L.k4_pipeline_open(self.h, C.byref(prm), C.byref(pl)))
        try:
            if ring:
                pos = [0] * len(bufs)
                done = [False] * len(bufs)
                while not all(done):
                    for e, (ptr, n, _) in enumerate(bufs):
                        if done[e]:
                            continue
                        b, cap = C.c_void_p(), C.c_uint64()
                        self._ck(L.k4_pipeline_acquire(pl, e, C.byref(b), C.byref(cap)))
                        ln = min(cap.value, n - pos[e])
                        C.memmove(b.value, ptr + pos[e], ln)
                        pos[e] += ln
                        done[e] = pos[e] == n
                        self._ck(L.k4_pipeline_submit(pl, e, ln, 1 if done[e] else 0))
            else:
                for e, (ptr, n, _) in enumerate(bufs):
                    self._ck(L.k4_pipeline_submit_host(pl, e, ptr, n, 1))
            view = PipelineView()
            self._ck(L.k4_pipeline_wait_aligned(pl, C.byref(view)))
            stats, nbytes = SamStats(), C.c_uint64()
            self._ck((L.k4_pipeline_format_all if all_reads else L.k4_pipeline_format)(pl, C.byref(stats), None, C.byref(nbytes)))
            st = {"nar": list(stats.nar), "plus": stats.plus, "minus": stats.minus, "n_lines": stats.n_lines,
                  "n_units": view.n_units, "n_under": view.n_under, "n_over": view.n_over, "sam_bytes": nbytes.value}
            if out is not None:
                got = C.c_uint64()
                self._ck(L.k4_pipeline_read_sam(pl, out.data_ptr(), out.numel(), C.byref(got)))
                return got.value, st, None
            parts = []
            while True:
                p, n = C.c_void_p(), C.c_uint64()
                self._ck(L.k4_pipeline_next_sam(pl, C.byref(p), C.byref(n)))
                if n.value == 0:
                    break
                parts.append(C.string_at(p.value, n.value))
            return b"".join(parts), st, None
        finally:
            L.k4_pipeline_close(pl)

    # -- the hot path (device buffers; pointers are ints, e.g. torch.Tensor.data_ptr()) -------------------------
    def kalign_pe_batch_dev(self, params, pe_params, n_pairs, max_read_len, d_reads, d_offs, d_lens, d_out, stream=0):
        self._ck(lib().k4_kalign_pe_batch_dev(self.h, C.byref(params), C.byref(pe_params), n_pairs, max_read_len,
                                              d_reads, d_offs, d_lens, d_out, stream))

    def select_hits_dev(self, n, max_ml, d_rr, d_hits, d_choice, stream=0):
        """MLMode eMLrand (`-r2`, KAligner.cpp:9945-9962): keep hits[choice % NumHits] of every accepted read."""
        self._ck(lib().k4_select_hits_dev(self.h, n, max_ml, d_rr, d_hits, d_choice, stream))

    def assign_multi(self, out, hits, ml_mode, max_reads_len):
        """CKAligner::AssignMultiMatches (`-r3` / `-r4`, KAligner.cpp:5092) over kalign_batch(pe_mode=1) results held in
        host arrays; returns (out, hits, reads assigned)."""
        import torch

        dev = torch.device("cuda", self.info()["device"])
        n, max_ml = hits.shape
        d_rr = torch.from_numpy(out.view(np.uint8).copy()).to(dev)
        d_hits = torch.from_numpy(hits.reshape(-1).view(np.uint8).copy()).to(dev)
        na = C.c_int64(0)
        self._ck(lib().k4_assign_multi_dev(self.h, ml_mode, max_reads_len, n, max_ml, d_rr.data_ptr(), d_hits.data_ptr(),
                                           C.byref(na), 0))
        return (d_rr.cpu().numpy().view(RESULT_DTYPE), d_hits.cpu().numpy().view(HIT_DTYPE).reshape(n, max_ml), na.value)

    def select_hits(self, out, hits, choice):
        """select_hits_dev over host arrays (kalign_batch's out / hits, uint32 draws); returns the new (out, hits)."""
        import torch

        dev = torch.device("cuda", self.info()["device"])
        n, max_ml = hits.shape
        d_rr = torch.from_numpy(out.view(np.uint8).copy()).to(dev)
        d_hits = torch.from_numpy(hits.reshape(-1).view(np.uint8).copy()).to(dev)
        d_ch = torch.from_numpy(np.ascontiguousarray(choice, dtype=np.uint32).view(np.int32)).to(dev)
        self.select_hits_dev(n, max_ml, d_rr.data_ptr(), d_hits.data_ptr(), d_ch.data_ptr())
        torch.cuda.synchronize()
        return (d_rr.cpu().numpy().view(RESULT_DTYPE), d_hits.cpu().numpy().view(HIT_DTYPE).reshape(n, max_ml))

    def kalign_batch_dev(self, params, n, max_read_len, d_reads, d_offs, d_lens, d_out, d_hits, stream=0):
        self._ck(lib().k4_kalign_batch_dev(self.h, C.byref(params), n, max_read_len, d_reads, d_offs, d_lens, d_out,
                                           d_hits, stream))

    def kalign_ext_batch_dev(self, params, n, max_read_len, d_reads, d_offs, d_lens, d_out, d_hits, d_seg2, stream=0):
        self._ck(lib().k4_kalign_ext_batch_dev(self.h, C.byref(params), n, max_read_len, d_reads, d_offs, d_lens, d_out, d_hits,
                                               d_seg2, stream))

    def align_reads_batch_dev(self, params, n, max_read_len, d_reads, d_offs, d_lens, d_rslt, d_inst, d_low, d_nxt,
                              d_hits, stream=0):
        self._ck(lib().k4_align_reads_batch_dev(self.h, C.byref(params), n, max_read_len, d_reads, d_offs, d_lens,
                                                d_rslt, d_inst, d_low, d_nxt, d_hits, stream))


def build_sa_device(concat_len, el_size, d_seq_ptr, d_sa_ptr, device=0):
    """GPU suffix sort (CSfxArray::Finalise -> QSortSeq, SfxArray.cpp:1758,9739)."""
    rc = lib().k4_build_sa_device(concat_len, el_size, d_seq_ptr, d_sa_ptr, device)
    if rc != 0:
        raise K4Error(rc, lib().k4_global_error().decode())
