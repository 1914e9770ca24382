/* include/k4sfx.h -- C ABI of libk4sfx.so: the MI355X-native replacement for the kit4b kalign hot path
 * (seed lookup in the CSfxArray suffix-array index + mismatch-bounded full-read extension).
 *
 * The reference has no FFI layer: the boundary it exposes for this path is the C++ class CSfxArray
 * (libkit4b/SfxArray.h:524-1023) as called by CKAligner (ngskit4b/KAligner.cpp).  Each entry point below
 * names the reference method(s) it replaces.  Plain pointers and sizes only; no torch / HIP types.
 * The C++ facade that keeps the CSfxArray method names on top of this ABI is include/k4_sfxarray.hpp;
 * the binding a kit4b maintainer would add is shown in INTEGRATION.md.
 *
 * Result conventions (same as the reference):
 *   < 0          teBSFrsltCodes  (libkit4b/ErrorCodes.h:15-97), text via k4_last_error()
 *   0..4         tHRslt          (libkit4b/SfxArray.h:79-87) for per-read alignment results
 * There is NO CPU fallback: every compute entry point runs hand-written HIP kernels on gfx950 and fails
 * with K4_ERR_NO_DEVICE when no GPU is usable.
 */
#ifndef K4SFX_H
#define K4SFX_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define K4_ABI_VERSION 2  /* 2: optional AlignReads phases (k4_hit.ext, k4_seg2, *_ext entry points), pipeline, comm */

/* teBSFrsltCodes values used by this library (libkit4b/ErrorCodes.h:15-97) */
enum {
  K4_OK = 0,               /* eBSFSuccess */
  K4_ERR_PARAMS = -100,    /* eBSFerrParams */
  K4_ERR_MEM = -95,        /* eBSFerrMem */
  K4_ERR_NOT_SFX = -94,    /* eBSFerrNotBioseq */
  K4_ERR_NOT_FASTA = -93,  /* eBSFerrNotFasta */
  K4_ERR_OPEN_FILE = -90,  /* eBSFerrOpnFile */
  K4_ERR_CREATE_FILE = -89,/* eBSFerrCreateFile */
  K4_ERR_FILE_VER = -86,   /* eBSFerrFileVer */
  K4_ERR_FILE_ACCESS = -85,/* eBSFerrFileAccess */
  K4_ERR_ENTRY = -51,      /* eBSFerrEntry */
  K4_ERR_PARSE = -47,      /* eBSFerrParse */
  K4_ERR_INTERNAL = -1,    /* eBSFerrInternal */
  K4_ERR_NO_DEVICE = -2,   /* (new) no usable gfx950 device / HIP runtime error */
  K4_ERR_UNSUPPORTED = -3  /* (new) feature outside the hot-path scope (bisulfite, colourspace, ...) */
};

/* tHRslt, libkit4b/SfxArray.h:79-87 */
enum { K4_HR_NONE = 0, K4_HR_HITS = 1, K4_HR_MMDELTA = 2, K4_HR_HITINSTS = 3, K4_HR_RMMDELTA = 4,
       K4_HR_SEQERRS = 5, K4_HR_FATAL = 6 };
/* eALStrand, libkit4b/SfxArray.h:72-77 */
enum { K4_STRAND_BOTH = 0, K4_STRAND_WATSON = 1, K4_STRAND_CRICK = 2 };
/* eNAR values AlignRead can assign, ngskit4b/KAligner.h:136-158 */
enum { K4_NAR_UNALIGNED = 0, K4_NAR_ACCEPTED = 1, K4_NAR_NS = 2, K4_NAR_NOHIT = 3, K4_NAR_MMDELTA = 4,
       K4_NAR_MULTIALIGN = 5, K4_NAR_TRIM = 6, K4_NAR_SPLICEJCTN = 7, K4_NAR_MICROINDEL = 8 };

typedef struct k4_index k4_index; /* opaque: the HBM-resident index (replaces a loaded CSfxArray) */

/* One tsHitLoci (libkit4b/SfxArray.h:239-260): Seg[0] as the default path fills it at SfxArray.cpp:6264-6307, plus -- in
 * `ext` -- what the optional phases of AlignReads add: Seg[0].TrimLeft / TrimRight and the Flg* bits.  16 bytes; `ext` is 0
 * on the default path.  The second segment of a microInDel / splice-junction hit lives in a k4_seg2 record. */
typedef struct {
  uint32_t chrom_id;   /* Seg[0].ChromID: 1-based tsSfxEntry.EntryID */
  uint32_t match_loci; /* Seg[0].MatchLoci: 0-based in the chromosome (the reference truncates to uint32) */
  uint16_t match_len;  /* Seg[0].MatchLen: the probe length; the first segment's length in a two-segment hit */
  uint8_t strand;      /* Seg[0].Strand: '+' or '-' */
  uint8_t mismatches;  /* Seg[0].Mismatches == Seg[0].TrimMismatches */
  uint32_t ext;        /* bits 0-11 Seg[0].TrimLeft, 12-23 Seg[0].TrimRight, then K4_EXT_* */
} k4_hit;
#define K4_EXT_CHIMERIC (1u << 24)  /* FlgChimeric: flank-trimmed alignment (`-c`, SfxArray.cpp:6064-6189) */
#define K4_EXT_INDEL (1u << 25)     /* FlgInDel: two segments around a microInDel (`-a`, :7526) */
#define K4_EXT_INSERT (1u << 26)    /* FlgInsert: the InDel is an insertion into the read */
#define K4_EXT_SPLICE (1u << 27)    /* FlgSplice: two segments around a splice junction (`-A`, :7208) */
#define K4_EXT_NONORPHAN (1u << 28) /* FlgNonOrphan: another read shares the junction (KAligner.cpp:2456-2465) */
#define K4_HIT_TRIM_LEFT(h) ((h).ext & 0xFFFu)
#define K4_HIT_TRIM_RIGHT(h) (((h).ext >> 12) & 0xFFFu)

/* Seg[1] of a two-segment hit and tsHitLoci.Score; one record per READ (LocateInDels / LocateSpliceJuncts report at most
 * one hit).  All zero unless hit slot 0 carries K4_EXT_INDEL or K4_EXT_SPLICE.  16 bytes. */
typedef struct {
  uint32_t chrom_id;   /* Seg[1].ChromID */
  uint32_t match_loci; /* Seg[1].MatchLoci */
  uint16_t match_len;  /* Seg[1].MatchLen */
  uint16_t read_ofs;   /* Seg[1].ReadOfs */
  uint8_t mismatches;  /* Seg[1].Mismatches */
  uint8_t reserved;
  uint16_t score;      /* Score */
} k4_seg2;

/* tsSfxEntry (libkit4b/SfxArray.h:98-106) without packing */
typedef struct {
  uint32_t entry_id;
  uint32_t fblock_id;
  char name[81];
  uint16_t name_hash;
  uint32_t seq_len;
  uint64_t start_ofs;
  uint64_t end_ofs;
} k4_entry;

typedef struct {
  uint64_t concat_len;   /* tsSfxBlock.ConcatSeqLen */
  uint64_t tot_seqs_len; /* CSfxArray::GetTotSeqsLen, SfxArray.h:970 */
  uint32_t sfx_el_size;  /* 4 | 5 */
  uint32_t n_entries;    /* CSfxArray::GetNumEntries, SfxArray.h:958 */
  uint32_t kmer_k;       /* length of the direct-address k-mer -> SA-interval table */
  uint32_t n_exc_blocks; /* 64-base blocks holding a non-ACGT symbol (N / EOS) */
  uint64_t device_bytes; /* HBM held by this index */
  int32_t device;        /* HIP device ordinal */
  int32_t max_iter;      /* CSfxArray::GetMaxIter */
  char dataset[81];      /* CSfxArray::GetDatasetName */
} k4_info_t;

/* The arguments of CSfxArray::AlignReads (libkit4b/SfxArray.h:614-634). */
typedef struct {
  int32_t tot_mm;          /* TotMM */
  int32_t core_len;        /* CoreLen */
  int32_t core_delta;      /* CoreDelta */
  int32_t max_core_slides; /* MaxNumCoreSlides */
  int32_t min_core_len;    /* MinCoreLen (used by the chimeric phase only, SfxArray.cpp:7925) */
  int32_t mm_delta;        /* MMDelta */
  int32_t strand;          /* Align2Strand: K4_STRAND_* */
  int32_t max_hits;        /* MaxHits */
  /* the optional phases a read enters when the ones above found nothing (SfxArray.cpp:7894-7930); 0 = off */
  int32_t min_chimeric_len;     /* MinChimericLen: 15..99, % of the read that must survive flank trimming */
  int32_t micro_indel_len;      /* microInDelLen: 1..20; needs the *_ext entry points (k4_seg2 output) */
  int32_t max_splice_junct_len; /* MaxSpliceJunctLen: 25..100000; needs the *_ext entry points */
} k4_align_params;

/* What CKAligner::AlignRead derives per read and how it classifies (ngskit4b/KAligner.cpp:9583-10105) */
typedef struct {
  int32_t max_subs;       /* -s per 100 bp (pPars->MaxSubs) */
  int32_t min_edit_dist;  /* -e (pPars->MinEditDist) 1|2 */
  int32_t max_ns;         /* -n (m_MaxNs), default 1 */
  int32_t pmode;          /* -m 0 default,1 more,2 ultra,3 less sensitive (KAligner.cpp:9377-9393) */
  int32_t strand;         /* K4_STRAND_* */
  int32_t max_ml;         /* max(m_MaxMLmatches, PE ? cMaxMLPEmatches : 0) (KAligner.cpp:9604) */
  int32_t pe_mode;        /* classification of a read with hits (KAligner.cpp:9907-10023): 0 SE, default MLMode (accepted,
                           * first instance); 1 PE (accepted iff one instance, else multi-aligned with NumHits = instances);
                           * 2 SE, MLMode eMLall (`-r5 -R<max_ml>`): accepted with NumHits = instances, every one reported;
                           * 3 = 2 with reads over the limit treated as if exactly max_ml matched (`-X`, :9856-9861);
                           * 4 = 3 through CSfxArray::LocateBestMatches (`-N`, SfxArray.cpp:6836-7205) instead of AlignReads */
  int32_t min_core_len;   /* 0: derive as LocateCoredApprox does (KAligner.cpp:9367-9393) */
  int32_t max_num_slides; /* 0: derive from pmode */
  int32_t min_chimeric_len;     /* -c (pPars->MinChimericLen, KAligner.cpp:9801): 0 or 15..99 */
  int32_t micro_indel_len;      /* -a (pPars->microInDelLen): 0..20; SE only; needs the *_ext entry points */
  int32_t max_splice_junct_len; /* -A (pPars->SpliceJunctLen): 0 or 25..100000; SE only; needs the *_ext entry points */
} k4_kalign_params;

typedef struct {
  int32_t hit_rslt; /* tHRslt from AlignReads (K4_HR_SEQERRS when the read has too many Ns) */
  int32_t inst;     /* tsReadHit.LowHitInstances */
  int32_t low_mm;   /* tsReadHit.LowMMCnt */
  int32_t nxt_mm;   /* tsReadHit.NxtLowMMCnt */
  int32_t nar;      /* tsReadHit.NAR */
  int32_t num_hits; /* tsReadHit.NumHits */
} k4_read_result;

/* run-time tallies for the roofline accounting; cumulative until k4_reset_counters */
typedef struct {
  uint64_t n_reads;    /* reads processed */
  uint64_t n_lookup;   /* seed lookups (one per LocateFirstExact the reference would have made) */
  uint64_t n_probe;    /* suffix-array probes actually issued (SA element + packed-reference window) */
  uint64_t n_cand;     /* candidates that reached the Hamming extension */
  uint64_t n_slow;     /* reads routed to the exact general kernel (N, separators, many candidates) */
  uint64_t n_bases;    /* read bases in */
} k4_counters;

/* ---- index life cycle ------------------------------------------------------------------------------
 * k4_open               <- CSfxArray::Open(path) + SetTargBlock(1)   SfxArray.h:528,957 (SfxArray.cpp:969,1982)
 * k4_open_host          <- same, from an in-memory block (CSfxArray::Open(bBisulfite,bColorspace) in-memory mode, SfxArray.h:533)
 * k4_open_device        <- same, adopting a 1-byte/base sequence and a suffix array already resident in HBM
 * k4_close              <- CSfxArray::Close/Reset                    SfxArray.h:527,548
 * kmer_k: 0 = choose from the index size; device: HIP ordinal. */
int k4_open(const char* sfx_path, int device, int kmer_k, k4_index** out);
/* k4_open in two halves (the reading of the reads can run beside the index load, as kalign's loader runs beside its aligner threads,
 * KAligner.cpp:4786-4866): k4_open_async returns once header and entry table are read; k4_info / k4_get_entry / k4_min_core_len /
 * k4_set_max_iter / k4_set_fastq_quality and a pipeline's ingest side (k4_pipeline_open, acquire / submit) may be used at once, a
 * library thread uploads the arrays and builds the device structures meanwhile.  k4_open_wait joins it and returns the load's
 * result; the pipeline waits by itself before its first alignment batch, every other entry point needs k4_open_wait first.
 * On an error from k4_open_wait the handle is still to be released with k4_close. */
int k4_open_async(const char* sfx_path, int device, int kmer_k, k4_index** out);
int k4_open_wait(k4_index* ix);
double k4_open_seconds(const k4_index* ix);  /* wall seconds the background load took (after k4_open_wait) */
int k4_open_host(uint64_t concat_len, uint32_t sfx_el_size, const uint8_t* seq, const uint8_t* sa,
                 uint32_t n_entries, const k4_entry* entries, const char* dataset, int device, int kmer_k,
                 k4_index** out);
int k4_open_device(uint64_t concat_len, uint32_t sfx_el_size, const void* d_seq, void* d_sa, int adopt_sa,
                   uint32_t n_entries, const k4_entry* entries, const char* dataset, int device, int kmer_k,
                   k4_index** out);
void k4_close(k4_index* ix);
/* k4_sfx_map / k4_sfx_unmap <- CSfxArray::Disk2Hdr / Disk2Entries (SfxArray.cpp:629-825): the .sfx file mapped read-only with
 * its tables decoded, for callers that place the index themselves (k4_open_host, or libk4comm's broadcast over xGMI) */
typedef struct {
  const void* map; size_t map_len;      /* the mapping (owned; k4_sfx_unmap releases it and `entries`) */
  uint64_t concat_len; uint32_t sfx_el_size, n_entries;
  k4_entry* entries;
  const uint8_t* seq;                   /* concat_len bytes, one etSeqBase per base, EOS separators */
  const uint8_t* sa;                    /* concat_len * sfx_el_size bytes */
  const uint8_t* header;                /* tsSfxHeaderV3, 1224 bytes */
  char dataset[81];
} k4_sfx_file;
int k4_sfx_map(const char* sfx_path, k4_sfx_file* out);
void k4_sfx_unmap(k4_sfx_file* f);
int k4_set_raw_header(k4_index* ix, const void* hdr_1224); /* the header an index opened from parts reports / writes */
const char* k4_last_error(const k4_index* ix);            /* <- CErrorCodes::GetErrMsg, ErrorCodes.h:99-113 */
const char* k4_global_error(void);                        /* error text when no index handle exists yet */
int k4_info(const k4_index* ix, k4_info_t* out);          /* <- GetNumEntries/GetTotSeqsLen/GetSfxHeader */
int k4_get_entry(const k4_index* ix, uint32_t entry_id, k4_entry* out); /* <- GetIdentName/GetSeqLen, SfxArray.h:967-969 */
int k4_get_ident(const k4_index* ix, const char* name);   /* <- CSfxArray::GetIdent, SfxArray.h:968 */
int k4_set_max_iter(k4_index* ix, int max_iter);          /* <- CSfxArray::SetMaxIter, SfxArray.h:556 */
/* k4_set_fastq_quality <- kalign -g<n> (etFQMethod, KAlignerCL.cpp:241,499): how the quality line of a FASTQ record is read --
 * 0 Sanger (Illumina 1.8+), 1 Illumina 1.3+, 2 Solexa / Illumina before 1.3, 3 ignore (the default).  With 0..2 k4_parse_fastx_dev
 * (and the pipeline) scale every base's score to 4 bits as LoadRawReads does (KAligner.cpp:12096-12158: phred clamped to 40,
 * (phred + 2) * 15 / 40) into bits 4..7 of the read byte, and the SAM / BAM writers report it as ReportBAMread does
 * (KAligner.cpp:6120-6145: '!' + score * 40 / 15, reversed for a Crick alignment, `*` / 0xff when every score is zero). */
int k4_set_fastq_quality(k4_index* ix, int method);
int k4_get_seq(const k4_index* ix, uint32_t entry_id, uint32_t loci, uint8_t* out, uint32_t len); /* <- GetSeq, SfxArray.h:996 */
int k4_write_sfx(const k4_index* ix, const char* sfx_path); /* <- CSfxArray::Finalise/Flush2Disk, SfxArray.cpp:892 */
int k4_set_description(k4_index* ix, const char* description, const char* title); /* <- SetDescription / SetTitle, SfxArray.h:552-553 */
int k4_get_sfx_header(const k4_index* ix, void* out_1224);  /* <- CSfxArray::GetSfxHeader: tsSfxHeaderV3, SfxArray.h:194-207 */

/* ---- suffix-array construction on the GPU -----------------------------------
 * <- CSfxArray::AddEntry + Finalise -> QSortSeq (SfxArray.cpp:1518,1758,9739): same suffix order
 * (A<C<G<T<N<EOS, compare stops after the first EOS), ties broken by offset.
 * d_seq: concat_len bytes in HBM (bases + one EOS per entry); d_sa: concat_len*sfx_el_size bytes in HBM, filled. */
int k4_build_sa_device(uint64_t concat_len, uint32_t sfx_el_size, const void* d_seq, void* d_sa, int device);

/* ---- the hot path -------------------------------------------------------------------------------------
 * k4_align_reads_batch  <- CSfxArray::AlignReads for n_reads fresh reads (In/Out ints start at 0 as in
 *                          KAligner.cpp:9609-9611,9678-9680), uniform parameters.   SfxArray.h:614 (SfxArray.cpp:7838)
 * k4_kalign_batch       <- CKAligner::AlignRead for n_reads reads.                  KAligner.cpp:9583
 * reads: concatenated etSeqBase bytes (A0 C1 G2 T3 N4; bits 3..7 ignored); read i = reads[offs[i] .. offs[i]+lens[i]).
 * hits: n_reads * max_hits records; slots beyond min(inst,max_hits) are zero.
 * *_dev variants take DEVICE pointers for every array and a hipStream_t (as void*); they enqueue only
 * (no host synchronisation) once the workspace has been sized by k4_reserve(). */
int k4_reserve(k4_index* ix, int64_t max_reads, int32_t max_read_len, int32_t max_hits);
int k4_align_reads_batch(k4_index* ix, const k4_align_params* p, int64_t n_reads, const uint8_t* reads,
                         const uint64_t* offs, const uint32_t* lens, int32_t* rslt, int32_t* inst, int32_t* low,
                         int32_t* nxt, k4_hit* hits);
int k4_align_reads_batch_dev(k4_index* ix, const k4_align_params* p, int64_t n_reads, int32_t max_read_len,
                             const void* d_reads, const void* d_offs, const void* d_lens, void* d_rslt, void* d_inst,
                             void* d_low, void* d_nxt, void* d_hits, void* stream);
int k4_kalign_batch(k4_index* ix, const k4_kalign_params* p, int64_t n_reads, const uint8_t* reads,
                    const uint64_t* offs, const uint32_t* lens, k4_read_result* out, k4_hit* hits);
int k4_kalign_batch_dev(k4_index* ix, const k4_kalign_params* p, int64_t n_reads, int32_t max_read_len,
                        const void* d_reads, const void* d_offs, const void* d_lens, void* d_out, void* d_hits,
                        void* stream);
int k4_min_core_len(const k4_index* ix, int pmode, int* max_num_slides); /* <- LocateCoredApprox, KAligner.cpp:9367-9393 */
/* The same two calls with the k4_seg2 output the microInDel / splice phases need: seg2 holds n_reads
 * records.  <- CSfxArray::AlignReads incl. LocateInDels (SfxArray.cpp:7526), LocateSpliceJuncts (:7208) and the chimeric
 * LocateCoreMultiples pass (:6064-6189, AdaptiveTrim :5561); <- CKAligner::AlignRead with -c / -a / -A. */
int k4_align_reads_ext_batch(k4_index* ix, const k4_align_params* p, int64_t n_reads, const uint8_t* reads,
                             const uint64_t* offs, const uint32_t* lens, int32_t* rslt, int32_t* inst, int32_t* low,
                             int32_t* nxt, k4_hit* hits, k4_seg2* seg2);
int k4_align_reads_ext_batch_dev(k4_index* ix, const k4_align_params* p, int64_t n_reads, int32_t max_read_len,
                                 const void* d_reads, const void* d_offs, const void* d_lens, void* d_rslt, void* d_inst,
                                 void* d_low, void* d_nxt, void* d_hits, void* d_seg2, void* stream);
int k4_kalign_ext_batch(k4_index* ix, const k4_kalign_params* p, int64_t n_reads, const uint8_t* reads,
                        const uint64_t* offs, const uint32_t* lens, k4_read_result* out, k4_hit* hits, k4_seg2* seg2);
int k4_kalign_ext_batch_dev(k4_index* ix, const k4_kalign_params* p, int64_t n_reads, int32_t max_read_len,
                            const void* d_reads, const void* d_offs, const void* d_lens, void* d_out, void* d_hits,
                            void* d_seg2, void* stream);
/* The post-alignment stages `kalign -A / -a / -x` run on the SE results (device arrays as k4_kalign_ext_batch_dev left them):
 * k4_auto_trim_flanks_dev     <- CKAligner::AutoTrimFlanks (KAligner.cpp:1714-1917): trims the flanks of one-segment hits
 *                                back to min_flank_exacts exactly matching bases (TrimLeft / TrimRight in k4_hit.ext); a read
 *                                that cannot keep half its length becomes K4_NAR_TRIM.  pe: the PE variant (no elimination).
 * k4_remove_orphan_juncts_dev <- RemoveOrphanSpliceJuncts / RemoveOrphanMicroInDels (KAligner.cpp:2406-2594), which =
 *                                K4_EXT_SPLICE or K4_EXT_INDEL: a junction no other read shares (within 3 bp at both ends)
 *                                loses its alignment (K4_NAR_SPLICEJCTN / K4_NAR_MICROINDEL).
 * Both wait for `stream` and return counts. */
int k4_auto_trim_flanks_dev(k4_index* ix, int32_t min_flank_exacts, int pe, int64_t n_reads, int32_t max_ml, void* d_rr,
                            void* d_hits, const void* d_reads, const void* d_offs, const void* d_lens, int64_t* n_eliminated,
                            void* stream);
int k4_remove_orphan_juncts_dev(k4_index* ix, uint32_t which, int64_t n_reads, int32_t max_ml, void* d_rr, void* d_hits,
                                const void* d_seg2, int64_t* n_removed, void* stream);
/* k4_best_matches_batch <- CSfxArray::LocateBestMatches (SfxArray.h:793, SfxArray.cpp:6836-7205; CKAligner's `-N`) for
 * n_reads reads: at most p->max_hits alignments with no more than p->tot_mm mismatches, sorted by mismatches; rslt = the
 * call's return value (0 none, 1..max_hits, max_hits+1 when further matches were sloughed), inst = alignments in the
 * read's hit slots.  mm_delta and min_core_len of *p are not used; MaxIter is the index's (k4_set_max_iter). */
int k4_best_matches_batch(k4_index* ix, const k4_align_params* p, int64_t n_reads, const uint8_t* reads, const uint64_t* offs,
                          const uint32_t* lens, int32_t* rslt, int32_t* inst, k4_hit* hits);
int k4_best_matches_batch_dev(k4_index* ix, const k4_align_params* p, int64_t n_reads, int32_t max_read_len,
                              const void* d_reads, const void* d_offs, const void* d_lens, void* d_rslt, void* d_inst,
                              void* d_hits, void* stream);

/* ---- paired ends -----------------------------------------------------------------------------------------
 * k4_mate_rescue_batch <- CSfxArray::AlignPairedRead (SfxArray.h:880, SfxArray.cpp:8571-8767; MinChimericLen 0, insert
 *                         window < 1000 => the linear-scan branch :8731-8766 with AdaptiveTrim's full-length rule :5612-5638)
 * k4_kalign_pe_batch   <- the PE flow of CKAligner: ProcCoredApprox (KAligner.cpp:10160-10239: both ends aligned as SE
 *                         with MaxHits 10, multi x multi resolution) + ProcessPairedEnds (:3159-3596: AcceptProvPE,
 *                         PEInsertSize, orphan rescue, NAR reassignment).  out[2i] = PE1 of pair i, out[2i+1] = PE2. */
enum { K4_NAR_CHROMFILT = 11, K4_NAR_PEINSERTMIN = 13, K4_NAR_PEINSERTMAX = 14, K4_NAR_PENOHIT = 15, K4_NAR_PESTRAND = 16,
       K4_NAR_PECHROM = 17, K4_NAR_PEUNALIGN = 18 };  /* eNAR, KAligner.h:136-158 */
typedef struct {
  int32_t pe_mode;      /* etPEproc (KAligner.h:278-282): 1 orphan recovery, 2 unique only, 3 orphanSE, 4 uniqueSE */
  int32_t pair_min_len; /* -d */
  int32_t pair_max_len; /* -D */
  int32_t pair_strand;  /* -E */
} k4_pe_params;
typedef struct {        /* the tsReadHit fields that matter after ProcessPairedEnds */
  int32_t nar;
  int32_t num_hits;
  int32_t inst;
  int32_t low_mm;
  int32_t pe_aligned;   /* FlgPEAligned */
  int32_t rescued;      /* 1 when the hit came from the mate rescue */
  k4_hit hit;
} k4_pe_read;
typedef struct {        /* one AlignPairedRead call */
  uint32_t chrom_id;    /* ChromID of the anchored mate */
  uint32_t start_loci;  /* StartLoci / EndLoci of the anchored mate */
  uint32_t end_loci;
  uint32_t read_len;    /* ReadLen of the mate to place */
  uint64_t read_off;    /* its bases: reads[read_off .. read_off+read_len) (etSeqBase bytes, sense as sequenced) */
  int32_t b3prime_extend;
  int32_t antisense;
  int32_t min_insert;
  int32_t max_insert;
  int32_t max_allowed_mm;
  uint32_t chimeric;    /* 0, or MinChimericLen (15..99) | CoreLen << 8 | CoreDelta << 20: AlignPairedRead's chimeric mode */
} k4_rescue_task;
int k4_mate_rescue_batch(k4_index* ix, int64_t n_tasks, const k4_rescue_task* tasks, const uint8_t* reads,
                         uint64_t reads_bytes, int32_t* rslt, k4_hit* hits);
int k4_kalign_pe_batch(k4_index* ix, const k4_kalign_params* p, const k4_pe_params* pe, int64_t n_pairs,
                       const uint8_t* reads1, const uint64_t* offs1, const uint32_t* lens1, const uint8_t* reads2,
                       const uint64_t* offs2, const uint32_t* lens2, k4_pe_read* out);
/* Device-resident form: reads interleaved (read 2i = PE1 of pair i, read 2i+1 = its PE2), every array in HBM; the SE
 * pass, the pairing kernel (AcceptProvPE / PEInsertSize / NAR reassignment, one thread per pair) and the orphan kernel
 * (AlignPairedRead, one wave per orphan pair) are enqueued on `stream`, which is then waited for. */
int k4_kalign_pe_batch_dev(k4_index* ix, const k4_kalign_params* p, const k4_pe_params* pe, int64_t n_pairs,
                           int32_t max_read_len, const void* d_reads, const void* d_offs, const void* d_lens,
                           void* d_out, void* stream);

/* ---- read ingest and SAM emit on the device ----------------------------------------------
 * k4_parse_fastx_dev   <- CKAligner::LoadRawReads (KAligner.cpp:11648-12421) over CFasta: FASTA ('>', sequence over any
 *                         number of lines) or FASTQ ('@', four lines per record) text resident in HBM -> etSeqBase reads
 *                         (a/c/g/t/u either case -> 0..3, '-' -> 6, other letters -> 4, anything else sloughed:
 *                         CFasta::ReadSequence / Ascii2Sense, Fasta.cpp:1172,1657), offsets, lengths, and the span of
 *                         each descriptor up to its first white space (<= 79 bytes; FASTA: behind leading blanks).  A chunk that is not the last one may end inside a
 *                         record: info->consumed tells how many bytes were used; resubmit the rest in front of the next
 *                         chunk.  The bases go to d_reads[reads_base ...) (room for text_bytes + 16 bytes), offs are
 *                         relative to d_reads; name offsets are text_base + offset in this chunk; the per-record arrays
 *                         hold max_records elements.
 * k4_prepare_reads_dev <- the length filter of LoadRawReads (:12024-12060; a pair is dropped when either mate fails) and
 *                         the PE1/PE2 interleave: slots of dropped reads stay in place with length 0.
 * k4_format_sam_dev    <- WriteBAMReadHits (:5718-5914) / ReportBAMread (:5957-6320) / SortHitMatch (:10969) /
 *                         CSAMfile::AddAlignment (SAMfile.cpp:2194-2377): the accepted reads as SAM lines in coordinate
 *                         order (chrom, start, len, strand, mismatches, then load order), in a buffer this call
 *                         allocates (*d_sam, release with k4_free_device); header lines are the caller's.  SE with
 *                         max_ml > 1 (eMLall): a read contributes one line per reported instance (NumHits).  Also the
 *                         ReportAlignStats tallies and which chromosomes received a hit (for the @SQ rule of :5785-5821). */
enum { K4_FASTA = 1, K4_FASTQ = 2 };
typedef struct {
  uint64_t n_records;
  uint64_t consumed;     /* text bytes that belonged to the records returned */
  uint64_t n_bases;
  uint32_t max_len;
  uint32_t format;       /* K4_FASTA | K4_FASTQ */
} k4_parse_info;
typedef struct {          /* where the QNAMEs live: [0] the reads / PE1 file, [1] the PE2 file (PE only) */
  const void* d_text[2];
  const void* d_name_off[2]; /* uint64 per record: byte offset into d_text */
  const void* d_name_len[2]; /* uint32 per record */
} k4_sam_names;
typedef struct {
  uint64_t nar[20];      /* reads per eNAR value (slots of dropped reads are not counted) */
  uint64_t plus, minus;  /* accepted alignments per strand */
  uint64_t n_lines;      /* SAM lines written */
} k4_sam_stats;
int k4_parse_fastx_dev(k4_index* ix, const void* d_text, uint64_t text_bytes, uint64_t text_base, int final_chunk,
                       int format /* 0: from the first byte */, int64_t max_records, void* d_reads, uint64_t reads_base,
                       void* d_offs, void* d_lens, void* d_name_off, void* d_name_len, k4_parse_info* info, void* stream);
int k4_prepare_reads_dev(k4_index* ix, int pe, int64_t n_units, int32_t min_len, int32_t max_len, const void* d_offs1,
                         const void* d_lens1, const void* d_offs2, const void* d_lens2, uint64_t reads2_base,
                         void* d_offs_out, void* d_lens_out, uint64_t* n_under, uint64_t* n_over, uint32_t* max_read_len,
                         void* stream);
/* ... with kalign's end trims (`-y` / `-Y`: KAligner.cpp:12254-12260, the length tests over the trimmed lengths :12040-12090) and
 * its sampling (`-#<n>`, :11983-11989: of the file's reads / pairs every n-th is loaded, the first included; first_unit = index in
 * the file of this call's first unit; 0 / 1: all) */
int k4_prepare_reads_trim_dev(k4_index* ix, int pe, int64_t n_units, int32_t min_len, int32_t max_len, int32_t trim5, int32_t trim3,
                              int32_t sample_nth, int64_t first_unit, const void* d_offs1, const void* d_lens1, const void* d_offs2, const void* d_lens2, uint64_t reads2_base,
                              void* d_offs_out, void* d_lens_out, uint64_t* n_under, uint64_t* n_over, uint32_t* max_read_len,
                              void* stream);
int k4_format_sam_dev(k4_index* ix, int pe, int64_t n_units, const void* d_rr, const void* d_hits, int32_t max_ml,
                      const void* d_pe, const void* d_reads, const void* d_offs, const void* d_lens,
                      const k4_sam_names* names, void** d_sam, uint64_t* sam_bytes, k4_sam_stats* stats,
                      uint8_t* chrom_hit /* host, n_entries + 1 bytes, or NULL */, void* stream);
/* the same with the second segments (d_seg2: one k4_seg2 per read, or NULL): CIGAR with soft clips for trimmed hits and
 * I / D / N between the segments, MAPQ scaled as ReportBAMread does (KAligner.cpp:6148-6233) */
int k4_format_sam_ext_dev(k4_index* ix, int pe, int64_t n_units, const void* d_rr, const void* d_hits, int32_t max_ml,
                          const void* d_pe, const void* d_seg2, const void* d_reads, const void* d_offs, const void* d_lens,
                          const k4_sam_names* names, void** d_sam, uint64_t* sam_bytes, k4_sam_stats* stats,
                          uint8_t* chrom_hit, void* stream);
/* k4_format_sam_all_dev <- kalign -M1 (eFMsamAll; WriteBAMReadHits KAligner.cpp:5846-5866, the unaligned branch of ReportBAMread
 * :6253-6276): the same body followed by one record per loaded read that was not accepted, grouped by NAR code in ascending order
 * (SortHitMatch sorts on NAR first; within a code the reference's order is undefined, here load order):
 * QNAME FLAG(4; PE: 1|2|64/128|4 and the mate's 8 or 32) * 0 128 <len>M * 0 0 SEQ(as read) * <empty field> YU:Z:<two-letter NAR> */
int k4_format_sam_all_dev(k4_index* ix, int pe, int64_t n_units, const void* d_rr, const void* d_hits, int32_t max_ml,
                          const void* d_pe, const void* d_seg2, const void* d_reads, const void* d_offs, const void* d_lens,
                          const k4_sam_names* names, void** d_sam, uint64_t* sam_bytes, k4_sam_stats* stats,
                          uint8_t* chrom_hit, void* stream);
/* ... as BAM records: refID, pos and the mate fields -1, bin 0, MAPQ 128, one CIGAR operation <len>M, aux YU:Z:<NAR> */
int k4_format_bam_all_dev(k4_index* ix, int pe, int64_t n_units, const void* d_rr, const void* d_hits, int32_t max_ml,
                          const void* d_pe, const void* d_seg2, const void* d_reads, const void* d_offs, const void* d_lens,
                          const k4_sam_names* names, int32_t sq_all, void** d_bam, uint64_t* bam_bytes, k4_sam_stats* stats,
                          uint8_t* chrom_hit, void* stream);
/* k4_unaligned_fasta_dev <- kalign -j / -J (CKAligner::ReportNoneAligned / ReportMultiAlign, KAligner.cpp:3833-4020): the loaded reads
 * whose NAR is EN or NL (which 0) / ML (which 1) as FASTA, `>lcl|na|<ReadID> <name> <ReadID>|1|<len>` (`lcl|ml` for which 1; ReadID
 * 1.. over the loaded reads in load order) and the read as loaded at 70 bases per line, grouped by NAR.  *text: malloc'd host text
 * (k4_free_host). */
int k4_unaligned_fasta_dev(k4_index* ix, int pe, int64_t n_units, const void* d_rr, const void* d_pe, const void* d_reads,
                           const void* d_offs, const void* d_lens, const k4_sam_names* names, int32_t which, char** text,
                           uint64_t* text_bytes, uint64_t* n_listed, void* stream);
/* k4_format_bam_dev <- the same alignments as uncompressed BAM records in coordinate order (CSAMfile::AddAlignment's BAM branch,
 * SAMfile.cpp:2379-2640: block_size, refID, pos, bin<<16|MAPQ<<8|l_read_name, FLAG<<16|n_cigar_op, l_seq, next_refID,
 * next_pos, tlen, read_name, cigar, 4-bit seq -- reverse complemented for a Crick alignment --, qual 0xff).  refID is the
 * index in the header's reference dictionary: every sequence of the index when sq_all, else those that received a hit, in
 * index order (WriteBAMReadHits, KAligner.cpp:5785-5821).  Magic, header text, dictionary, BGZF blocks and the .bai index
 * are the caller's (include/k4_bam.hpp does them on host threads). */
int k4_format_bam_dev(k4_index* ix, int pe, int64_t n_units, const void* d_rr, const void* d_hits, int32_t max_ml,
                      const void* d_pe, const void* d_seg2, const void* d_reads, const void* d_offs, const void* d_lens,
                      const k4_sam_names* names, int32_t sq_all, void** d_bam, uint64_t* bam_bytes, k4_sam_stats* stats,
                      uint8_t* chrom_hit, void* stream);
/* k4_snp_csv_dev <- CKAligner::ProcessSNPs + OutputSNPs (KAligner.cpp:8168-8590, 7098-7760; `kalign -p<n> -P<q> -S<file>`): SNP calls
 * over the accepted alignments of a run, as the text of kalign's main SNP CSV (header line included).  Per chromosome on the
 * device: base counts per locus, the 51-base background window, the coverage / non-reference tests; on the host, for the loci that
 * pass: binomial p-value (CStats::Binomial), Benjamini-Hochberg at `qvalue`, ranks, text.  min_snp_reads: kalign -p (1..),
 * snp_nonref_pcnt: kalign -1 (percent).  *csv is malloc'd: release with k4_free_host.  The files kalign writes beside the CSV
 * (coverage WIG, DiSNPs, TriSNPs, markers) are not produced. */
int k4_snp_csv_dev(k4_index* ix, int pe, int64_t n_units, const void* d_rr, const void* d_hits, int32_t max_ml, const void* d_pe,
                   const void* d_reads, const void* d_offs, const void* d_lens, int32_t min_snp_reads, double qvalue,
                   double snp_nonref_pcnt, char** csv, uint64_t* csv_bytes, uint64_t* n_snps, void* stream);
/* ... as VCF 4.1 (what kalign writes when the SNP file's name ends in .vcf): ALT = the alleles with at least a tenth of the strongest
 * one's count, AF per allele, QUAL = phred of the p-value capped at 100, DP = coverage */
int k4_snp_vcf_dev(k4_index* ix, int pe, int64_t n_units, const void* d_rr, const void* d_hits, int32_t max_ml, const void* d_pe,
                   const void* d_reads, const void* d_offs, const void* d_lens, int32_t min_snp_reads, double qvalue,
                   double snp_nonref_pcnt, char** vcf, uint64_t* vcf_bytes, uint64_t* n_snps, void* stream);
/* both files of a kalign SNP run: the SNP file (vcf == 0: CSV) and the coverage WIG kalign writes beside it (<snp file minus its
 * extension>.covsegs.wig: variableStep spans of roughly equal coverage, AccumWIGCnts / CompleteWIGSpan KAligner.cpp:6993-7085;
 * walked on host threads, one per chromosome, while the device piles up the next ones) */
int k4_snp_files_dev(k4_index* ix, int vcf, int pe, int64_t n_units, const void* d_rr, const void* d_hits, int32_t max_ml,
                     const void* d_pe, const void* d_reads, const void* d_offs, const void* d_lens, int32_t min_snp_reads, double qvalue,
                     double snp_nonref_pcnt, char** snp, uint64_t* snp_bytes, uint64_t* n_snps, char** wig, uint64_t* wig_bytes,
                     void* stream);
/* every file of a kalign SNP run: the two above and the haplotype files kalign writes beside them (<snp file minus its extension>
 * .disnp.csv / .trisnp.csv, KAligner.cpp:4553-4554, header :8252-8330, lines :7767-8101: two / three called loci following each
 * other within min(300, mean aligned length) bases, the alignments covering all of them (IterateReadsOverlapping :10475-10546)
 * and how many show each base combination (AdjAlignSNPBase :1581-1632)).  Each text is malloc'd: release with k4_free_host. */
typedef struct k4_snp_files {
  char* snp;    uint64_t snp_bytes;  uint64_t n_snps;
  char* wig;    uint64_t wig_bytes;
  char* disnp;  uint64_t disnp_bytes;
  char* trisnp; uint64_t trisnp_bytes;
} k4_snp_files;
int k4_snp_run_dev(k4_index* ix, int vcf, int pe, int64_t n_units, const void* d_rr, const void* d_hits, int32_t max_ml,
                   const void* d_pe, const void* d_reads, const void* d_offs, const void* d_lens, int32_t min_snp_reads, double qvalue,
                   double snp_nonref_pcnt, k4_snp_files* out, void* stream);
void k4_free_host(void* p);
/* k4_select_hits_dev <- MLMode eMLrand (`-r2`, KAligner.cpp:9945-9962) after k4_kalign_batch_dev with pe_mode 2: every accepted
 * read keeps ONE instance, hits[choice[i] % NumHits] moved to slot 0, NumHits = 1.  d_choice: uint32 per read, the caller's
 * draws (the reference: rand() once per read within the limit, in load order when it runs one thread). */
int k4_select_hits_dev(k4_index* ix, int64_t n_reads, int32_t max_ml, void* d_rr, void* d_hits, const void* d_choice,
                       void* stream);
/* k4_assign_multi_dev <- CKAligner::AssignMultiMatches (KAligner.cpp:5092-5258; scoring: ProcAssignMultiMatches :4944-5085),
 * MLMode eMLuniq (ml_mode 3, `-r3`) / eMLmulti (4, `-r4`), after k4_kalign_batch_dev with pe_mode 1 over ALL reads of the
 * run (the clustering is global): a multi-aligned read whose best locus clusters well enough with other reads (score >= 50
 * and twice the next locus') becomes accepted with that locus in slot 0 (NumHits 1, LowHitInstances 1).
 * max_reads_len: the longest read loaded (m_MaxReadsLen).  Waits for `stream`. */
int k4_assign_multi_dev(k4_index* ix, int ml_mode, int32_t max_reads_len, int64_t n_reads, int32_t max_ml, void* d_rr,
                        void* d_hits, int64_t* n_assigned, void* stream);
/* ---- the overlapped host <-> device pipeline (kit4b_amd/csrc/k4_pipeline.hip) ---------------------------
 * <- the reference's loader thread running beside its aligner threads (CKAligner::InitiateLoadingReads / ProcLoadReadFiles,
 *    KAligner.cpp:4786-4866,11323-11496; ThreadedIterReads :10370-10438) and its writer (WriteBAMReadHits :5718).
 * Three HIP streams: the caller's reader thread(s) fill pinned ring buffers (k4_pipeline_acquire / _submit) whose contents go
 * up on the copy stream while a worker thread of the library parses, filters and aligns the chunks that have already arrived
 * on the compute stream; k4_pipeline_format then makes ONE coordinate-sorted SAM body over everything (identical to a single
 * batch), which k4_pipeline_next_sam hands out piece by piece from a second pinned ring while the next pieces come down.
 * One acquire, then one submit, per end at a time; `final` marks the last chunk of an end (bytes may be 0). */
typedef struct k4_pipeline k4_pipeline;
typedef struct {
  int32_t paired;               /* 0: one input (SE), 1: two inputs, record i of end 0 pairs with record i of end 1 */
  k4_kalign_params kp;          /* as for k4_kalign_ext_batch_dev / k4_kalign_pe_batch_dev (PE: max_ml 10, pe_mode 1) */
  k4_pe_params pe;
  int32_t min_len, max_len;     /* the length filter of LoadRawReads (k4_prepare_reads_dev) */
  uint32_t n_buffers;           /* pinned buffers per end (0: 3) */
  uint32_t min_batch_units;     /* reads / pairs that must have arrived before an alignment batch is launched (0: 2^22; not
                                 * over the last eighth of an input whose size is known: little is left behind the last upload) */
  uint64_t chunk_bytes;         /* size of one pinned buffer = one upload (0: 256 MiB) */
  uint64_t expect_text_bytes[2];/* text per end, if known (file sizes): the HBM arena is sized once */
} k4_pipeline_params;
typedef struct {                /* where the aligned batch lives in HBM (for the global stages between alignment and report:
                                 * k4_assign_multi_dev, k4_select_hits_dev, k4_auto_trim_flanks_dev, k4_remove_orphan_juncts_dev) */
  int64_t n_units, n_reads;     /* reads (SE) / pairs (PE); reads through the SE pass */
  uint32_t max_read_len;
  int32_t max_ml;
  uint64_t n_under, n_over;     /* sloughed by the length filter */
  void *d_rr, *d_hits, *d_seg2, *d_pe, *d_reads, *d_offs, *d_lens;
  k4_sam_names names;
} k4_pipeline_view;
int k4_pipeline_open(k4_index* ix, const k4_pipeline_params* p, k4_pipeline** out);
int k4_pipeline_set_trims(k4_pipeline* pl, int32_t trim5, int32_t trim3); /* kalign -y / -Y; before the first chunk is submitted */
int k4_pipeline_set_sampling(k4_pipeline* pl, int32_t sample_nth);          /* kalign -#<n> (one input file per end); likewise */
int k4_pipeline_acquire(k4_pipeline* pl, int end, void** buf, uint64_t* cap); /* blocks while every buffer is on its way up */
int k4_pipeline_submit(k4_pipeline* pl, int end, uint64_t bytes, int final_chunk);
/* text in the caller's own memory (pinned for the full PCIe rate); it must stay valid until k4_pipeline_wait_aligned returns */
int k4_pipeline_submit_host(k4_pipeline* pl, int end, const void* text, uint64_t bytes, int final_chunk);
int k4_pipeline_wait_aligned(k4_pipeline* pl, k4_pipeline_view* view); /* after the final chunks: every read is aligned */
int k4_pipeline_format(k4_pipeline* pl, k4_sam_stats* stats, uint8_t* chrom_hit /* host, n_entries + 1, or NULL */, uint64_t* sam_bytes);
/* ... as BAM records (k4_format_bam_dev); the pieces come down through k4_pipeline_next_sam / k4_pipeline_read_sam all the same */
int k4_pipeline_format_bam(k4_pipeline* pl, int32_t sq_all, k4_sam_stats* stats, uint8_t* chrom_hit, uint64_t* bam_bytes);
/* ... as SAM text with every loaded read (k4_format_sam_all_dev, kalign -M1) */
int k4_pipeline_format_all(k4_pipeline* pl, k4_sam_stats* stats, uint8_t* chrom_hit, uint64_t* sam_bytes);
int k4_pipeline_format_bam_all(k4_pipeline* pl, int32_t sq_all, k4_sam_stats* stats, uint8_t* chrom_hit, uint64_t* bam_bytes);
int k4_pipeline_next_sam(k4_pipeline* pl, const void** ptr, uint64_t* bytes); /* valid until the next call; 0 bytes: done */
int k4_pipeline_read_sam(k4_pipeline* pl, void* dst, uint64_t cap, uint64_t* bytes); /* the whole body into caller memory */
void k4_pipeline_close(k4_pipeline* pl);

void k4_free_device(void* p);
/* device memory for host programs that use the *_dev entry points without a HIP runtime of their own */
int k4_alloc_device(k4_index* ix, uint64_t bytes, void** d_ptr);
int k4_copy_to_device(k4_index* ix, void* d_dst, const void* src, uint64_t bytes);
int k4_copy_to_host(k4_index* ix, void* dst, const void* d_src, uint64_t bytes);
/* pageable or file-mapped host memory -> device memory through pinned pieces filled by several host threads (the way k4_open
 * itself uploads the two big arrays of an .sfx file); for callers that place an index themselves, e.g. libk4comm's rank 0 */
int k4_upload_pageable(int device, void* d_dst, const void* h_src, uint64_t bytes);
/* the caller's own host memory (page-aligned address and length, e.g. a mapped file) registered with the HIP runtime in place, so
 * that the device copies of the *_dev / pipeline entry points run from / into it at the full PCIe rate without a staging copy */
int k4_host_register(void* p, uint64_t bytes);
int k4_host_unregister(void* p);

/* kernel timing for roofline measurement: when enabled, every batch brackets the dominant kernel (k4k_align_step, one launch per AlignReads phase) with
 * HIP events on the stream it is launched on; k4_get_kernel_times synchronises, returns the summed duration and the
 * number of launches since the last call, and resets. */
int k4_enable_kernel_timing(k4_index* ix, int on);
int k4_get_kernel_times(k4_index* ix, double* fast_kernel_ms, int32_t* launches);
/* ... both parts of a batch: the step kernels (as above) and the general kernel's passes (k4k_align_slow) that follow them on
 * the same stream; their sum is the device time of the batch's alignment */
int k4_get_kernel_times_split(k4_index* ix, double* step_kernels_ms, double* general_kernel_ms, int32_t* launches);
int k4_get_counters(k4_index* ix, k4_counters* out); /* synchronises the device */
int k4_reset_counters(k4_index* ix);
int k4_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif
