/* oracle/k4oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C) of the kit4b short-read alignment hot path, used as the parity checker in
 * tests/, in __graft_entry__.smoke() and as bench.py's cpu_baseline ("port").  Nothing on the product
 * path (kit4b_amd/, include/) may include, link or call this.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks this restatement against the golden vectors
 * under tests/golden/ that were captured from the real reference library (oracle/_ref/libk4ref.so, built
 * by oracle/Makefile from /root/reference) by tests/golden/make_golden.py; when oracle/_ref is present the
 * same tests also compare against the live reference on fresh random inputs.
 *
 * Every function cites the reference file:line it restates (paths relative to /root/reference).
 */
#ifndef K4ORACLE_H
#define K4ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* etSeqBase values, libkit4b/commdefs.h:76-87 */
enum { K4O_A = 0, K4O_C = 1, K4O_G = 2, K4O_T = 3, K4O_N = 4, K4O_EOS = 7 };

/* tHRslt, libkit4b/SfxArray.h:79-87 */
enum { K4O_HR_NONE = 0, K4O_HR_HITS = 1, K4O_HR_MMDELTA = 2, K4O_HR_HITINSTS = 3, K4O_HR_RMMDELTA = 4,
       K4O_HR_SEQERRS = 5, K4O_HR_FATAL = 6 };

/* eALStrand, libkit4b/SfxArray.h:72-77 */
enum { K4O_STRAND_BOTH = 0, K4O_STRAND_WATSON = 1, K4O_STRAND_CRICK = 2 };

/* eNAR (subset reachable from AlignRead), ngskit4b/KAligner.h:136-158 */
enum { K4O_NAR_UNALIGNED = 0, K4O_NAR_ACCEPTED = 1, K4O_NAR_NS = 2, K4O_NAR_NOHIT = 3, K4O_NAR_MMDELTA = 4,
       K4O_NAR_MULTIALIGN = 5 };

/* the fields of tsHitLoci.Seg[0] that the default path fills (libkit4b/SfxArray.h:239-260,
 * SfxArray.cpp:6264-6307); 16 bytes, same layout as k4_hit in include/k4sfx.h */
typedef struct {
  uint32_t chrom_id;   /* Seg[0].ChromID: 1-based EntryID */
  uint32_t match_loci; /* Seg[0].MatchLoci: 0-based, the reference truncates it to uint32 (SfxArray.cpp:6277) */
  uint16_t match_len;  /* Seg[0].MatchLen = ProbeLen */
  uint8_t strand;      /* '+' or '-' */
  uint8_t mismatches;  /* Seg[0].Mismatches */
  uint32_t ext;        /* 0 for the default path; the optional phases: bits 0-11 Seg[0].TrimLeft, 12-23 Seg[0].TrimRight,
                        * 24 FlgChimeric, 25 FlgInDel, 26 FlgInsert, 27 FlgSplice, 28 FlgNonOrphan (same as k4_hit.ext) */
} k4o_hit;
#define K4O_EXT_CHIMERIC (1u << 24)
#define K4O_EXT_INDEL (1u << 25)
#define K4O_EXT_INSERT (1u << 26)
#define K4O_EXT_SPLICE (1u << 27)
#define K4O_EXT_NONORPHAN (1u << 28)

/* Seg[1] of a two-segment hit (microInDel / splice junction: those phases report at most one hit per read) and the
 * tsHitLoci.Score; 16 bytes, same layout as k4_seg2 in include/k4sfx.h.  All zero for a one-segment result. */
typedef struct {
  uint32_t chrom_id;   /* Seg[1].ChromID */
  uint32_t match_loci; /* Seg[1].MatchLoci */
  uint16_t match_len;  /* Seg[1].MatchLen */
  uint16_t read_ofs;   /* Seg[1].ReadOfs */
  uint8_t mismatches;  /* Seg[1].Mismatches */
  uint8_t reserved;
  uint16_t score;      /* tsHitLoci.Score */
} k4o_seg2;

/* the three arguments of CSfxArray::AlignReads that switch its optional phases on (SfxArray.cpp:7894-7930) */
typedef struct {
  int min_chimeric_len;     /* MinChimericLen: 0, or 15..99 (% of the read that must remain after flank trimming) */
  int micro_indel_len;      /* microInDelLen: 0..20 */
  int max_splice_junct_len; /* MaxSpliceJunctLen: 0, or 25..100000 */
} k4o_ext_params;

typedef struct {
  uint32_t entry_id;   /* tsSfxEntry.EntryID (1..n) */
  uint32_t fblock_id;  /* tsSfxEntry.fBlockID */
  char name[81];       /* tsSfxEntry.szSeqName */
  uint16_t name_hash;  /* CUtility::GenHash16 */
  uint32_t seq_len;
  uint64_t start_ofs;
  uint64_t end_ofs;    /* inclusive, excludes EOS */
} k4o_entry;

typedef struct k4o_index k4o_index;

/* counters used for the roofline accounting of SURVEY.md 8(d) */
typedef struct {
  uint64_t n_lookup; /* LocateFirstExact calls */
  uint64_t n_probe;  /* binary-search probes (SA element + suffix compare) */
  uint64_t n_cand;   /* candidates that reached the Hamming extension */
} k4o_counters;

/* ---- index: .sfx I/O and construction ----------------------------------------------------------- */
k4o_index* k4o_open(const char* sfx_path, char* err, size_t errlen);          /* SfxArray.cpp:629-825,1915-1969 */
k4o_index* k4o_build(int nseq, const char* const* names, const uint8_t* const* seqs, const uint32_t* lens,
                     const char* dataset, int force_el_size, int nthreads);  /* SfxArray.cpp:1518-1753,9739-9834 */
k4o_index* k4o_from_parts(uint64_t n, uint32_t el, uint8_t* seq, uint8_t* sa, uint32_t n_entries,
                          const k4o_entry* entries, const char* dataset);    /* adopts seq/sa (not freed) */
int k4o_write(const k4o_index* ix, const char* sfx_path);                     /* SfxArray.cpp:380-620,892-946 */
void k4o_close(k4o_index* ix);

uint64_t k4o_concat_len(const k4o_index* ix);
uint32_t k4o_el_size(const k4o_index* ix);
const uint8_t* k4o_seq(const k4o_index* ix);
const uint8_t* k4o_sa_bytes(const k4o_index* ix);
uint32_t k4o_num_entries(const k4o_index* ix);
const k4o_entry* k4o_entries(const k4o_index* ix);
uint64_t k4o_tot_seqs_len(const k4o_index* ix);                               /* CSfxArray::GetTotSeqsLen */
int64_t k4o_sa_at(const k4o_index* ix, int64_t i);                            /* SfxOfsToLoci, SfxArray.cpp:49-60 */
void k4o_set_max_iter(k4o_index* ix, int max_iter);                           /* SfxArray.cpp:1501 */

/* ---- search core ---------------------------------------------------------------------------------- */
int64_t k4o_locate_first_exact(const k4o_index* ix, const uint8_t* probe, int probe_len, int64_t lo, int64_t hi,
                               k4o_counters* ctr);                            /* SfxArray.cpp:7938-8058 */
int k4o_locate_core_multiples(const k4o_index* ix, int max_tot_mm, int core_len, int core_delta, int max_slides,
                              int mm_delta, int strand, int* inst, int* low, int* nxt, uint8_t* probe,
                              int probe_len, int max_hits, k4o_hit* hits, k4o_counters* ctr); /* :5806-6369 */
int k4o_locate_best_matches(const k4o_index* ix, int max_tot_mm, int core_len, int core_delta, int max_slides, int strand,
                            uint8_t* probe, int probe_len, int max_hits, int* inst, k4o_hit* hits /* max_hits + 1 */,
                            int cur_max_iter, k4o_counters* ctr);                    /* :6836-7205 */
int k4o_align_reads(const k4o_index* ix, int tot_mm, int core_len, int core_delta, int max_slides, int min_core_len,
                    int mm_delta, int strand, int* inst, int* low, int* nxt, uint8_t* probe, int probe_len,
                    int max_hits, k4o_hit* hits, k4o_counters* ctr);          /* SfxArray.cpp:7838-7933 */

/* ---- the optional phases of AlignReads: oracle/k4oracle_ext.c ------------------------------- */
int k4o_adaptive_trim(uint32_t seq_len, const uint8_t* probe, const uint8_t* targ, uint32_t min_trim_len, uint32_t max_mm,
                      uint32_t min_flank_matches, uint32_t* trim_seq_len, uint32_t* trim_start, uint32_t* trim_end,
                      uint32_t* trim_mms);                                     /* SfxArray.cpp:5561-5795 */
int k4o_locate_core_multiples_chimeric(const k4o_index* ix, int min_chimeric_len, int max_tot_mm, int core_len, int core_delta,
                                       int max_slides, int mm_delta, int strand, int* inst, int* low, int* nxt,
                                       uint8_t* probe, int probe_len, int max_hits, k4o_hit* hits,
                                       k4o_counters* ctr);                     /* SfxArray.cpp:5806-6369 incl. :6064-6189 */
int k4o_locate_indels(const k4o_index* ix, int micro_indel_len, int max_tot_mm, int core_len, int strand, int* inst, int* low,
                      int* nxt, uint8_t* probe, int probe_len, int max_hits, k4o_hit* hit, k4o_seg2* seg2, int* score,
                      k4o_counters* ctr);                                      /* SfxArray.cpp:7526-7832, 9277-9735 */
int k4o_locate_splice_juncts(const k4o_index* ix, int max_splice_junct_len, int max_tot_mm, int core_len, int strand, int* inst,
                             int* low, int* nxt, uint8_t* probe, int probe_len, int max_hits, k4o_hit* hit, k4o_seg2* seg2,
                             int* score, k4o_counters* ctr);                   /* SfxArray.cpp:7208-7523, 8771-9265 */
/* AlignReads with all of its arguments (SfxArray.cpp:7838-7933); seg2: one record (slot 0's second segment) */
int k4o_align_reads_ext(const k4o_index* ix, const k4o_ext_params* ext, int tot_mm, int core_len, int core_delta, int max_slides,
                        int min_core_len, int mm_delta, int strand, int* inst, int* low, int* nxt, uint8_t* probe,
                        int probe_len, int max_hits, k4o_hit* hits, k4o_seg2* seg2, k4o_counters* ctr);
int k4o_align_reads_ext_batch(const k4o_index* ix, const k4o_ext_params* ext, int tot_mm, int core_len, int core_delta,
                              int max_slides, int min_core_len, int mm_delta, int strand, int max_hits, int64_t n_reads,
                              const uint8_t* reads, const uint64_t* offs, const uint32_t* lens, int32_t* rslt, int32_t* inst,
                              int32_t* low, int32_t* nxt, k4o_hit* hits, k4o_seg2* seg2, int nthreads, k4o_counters* ctr);
/* ---- CKAligner::AlignRead level ------------------------------------------------------------------- */
typedef struct {
  int max_subs;      /* -s: allowed substitutions per 100 bp (KAlignerCL.cpp:796) */
  int min_edit_dist; /* -e: 1 or 2 (MMDelta) */
  int max_ns;        /* -n: default 1 */
  int pmode;         /* -m: 0 default,1 more,2 ultra,3 less (KAligner.cpp:9377-9393) */
  int strand;        /* K4O_STRAND_* */
  int max_ml;        /* max(MaxMLmatches, PE ? 10 : 0): MaxHits handed to AlignReads (KAligner.cpp:9604) */
  int pe_mode;       /* 0 SE classification, 1 PE classification (KAligner.cpp:9982-10023) */
  int min_core_len;  /* 0 = derive from the index (KAligner.cpp:9367-9393) */
  int max_num_slides;/* 0 = derive from pmode */
  int min_chimeric_len;     /* -c (pPars->MinChimericLen) */
  int micro_indel_len;      /* -a (pPars->microInDelLen) */
  int max_splice_junct_len; /* -A (pPars->SpliceJunctLen) */
} k4o_kalign_params;

typedef struct {
  int32_t hit_rslt;  /* tHRslt returned by AlignReads (or K4O_HR_SEQERRS) */
  int32_t inst;      /* LowHitInstances after the clamp to MaxML+1 (KAligner.cpp:9854) */
  int32_t low_mm;
  int32_t nxt_mm;
  int32_t nar;       /* eNAR */
  int32_t num_hits;  /* tsReadHit.NumHits */
} k4o_read_result;

int k4o_min_core_len(const k4o_index* ix, int pmode, int* max_num_slides);   /* KAligner.cpp:9367-9393 */
void k4o_read_params(const k4o_kalign_params* kp, int min_core_len, int max_num_slides_per100, int read_len,
                     int* max_tot_mm, int* core_len, int* core_delta, int* max_slides); /* KAligner.cpp:9662-9672 */
int k4o_align_read(const k4o_index* ix, const k4o_kalign_params* kp, const uint8_t* read, int read_len,
                   k4o_read_result* out, k4o_hit* hits, k4o_counters* ctr);  /* KAligner.cpp:9583-10105 */
/* reads: concatenated etSeqBase bytes, read i = reads[offs[i] .. offs[i]+lens[i]); hits: n*max_ml records */
int k4o_align_batch(const k4o_index* ix, const k4o_kalign_params* kp, int64_t n_reads, const uint8_t* reads,
                    const uint64_t* offs, const uint32_t* lens, k4o_read_result* out, k4o_hit* hits,
                    int nthreads, k4o_counters* ctr);                          /* KAligner.cpp:10110-10263 */

/* CKAligner::AlignRead over a batch with kp->min_chimeric_len / micro_indel_len / max_splice_junct_len honoured; seg2: one
 * record per read */
int k4o_align_ext_batch(const k4o_index* ix, const k4o_kalign_params* kp, int64_t n_reads, const uint8_t* reads,
                        const uint64_t* offs, const uint32_t* lens, k4o_read_result* out, k4o_hit* hits, k4o_seg2* seg2,
                        int nthreads, k4o_counters* ctr);
/* the post-alignment stages those options bring with them (SE): eNAR values KAligner.h:136-158 */
enum { K4O_NAR_TRIM = 6, K4O_NAR_SPLICEJCTN = 7, K4O_NAR_MICROINDEL = 8 };
/* CKAligner::AutoTrimFlanks (KAligner.cpp:1714-1917; `-x`, implied by `-A` without `-c`): returns the number of reads
 * eliminated; reads / offs / lens as for the batch calls, hits: max_ml per read (slot 0 is the reported one) */
int64_t k4o_auto_trim_flanks(const k4o_index* ix, int min_flank_exacts, int pe, int64_t n_reads, const uint8_t* reads,
                             const uint64_t* offs, const uint32_t* lens, int max_ml, k4o_read_result* rr, k4o_hit* hits,
                             const k4o_seg2* seg2);
/* CKAligner::RemoveOrphanSpliceJuncts / RemoveOrphanMicroInDels (KAligner.cpp:2406-2594): which = K4O_EXT_SPLICE or
 * K4O_EXT_INDEL; returns the number of orphans removed */
int64_t k4o_remove_orphan_juncts(uint32_t which, int64_t n_reads, int max_ml, k4o_read_result* rr, k4o_hit* hits,
                                 const k4o_seg2* seg2);

/* raw AlignReads over a batch with uniform explicit parameters (the CSfxArray boundary) */
int k4o_align_reads_batch(const k4o_index* ix, int tot_mm, int core_len, int core_delta, int max_slides,
                          int min_core_len, int mm_delta, int strand, int max_hits, int64_t n_reads,
                          const uint8_t* reads, const uint64_t* offs, const uint32_t* lens, int32_t* rslt,
                          int32_t* inst, int32_t* low, int32_t* nxt, k4o_hit* hits, int nthreads,
                          k4o_counters* ctr);

/* ---- paired ends ------------------------------------------------------------------------------------------ */
/* eNAR values the PE pass assigns, ngskit4b/KAligner.h:136-158 */
enum { K4O_NAR_CHROMFILT = 11, K4O_NAR_PEINSERTMIN = 13, K4O_NAR_PEINSERTMAX = 14, K4O_NAR_PENOHIT = 15,
       K4O_NAR_PESTRAND = 16, K4O_NAR_PECHROM = 17, K4O_NAR_PEUNALIGN = 18 };

typedef struct {
  int pe_mode;       /* etPEproc: 1 orphan recovery, 2 unique only, 3 orphanSE, 4 uniqueSE (KAligner.h:278-282) */
  int pair_min_len;  /* -d */
  int pair_max_len;  /* -D */
  int pair_strand;   /* -E */
} k4o_pe_params;

typedef struct {     /* the tsReadHit fields that matter downstream of ProcessPairedEnds */
  int32_t nar;
  int32_t num_hits;
  int32_t inst;
  int32_t low_mm;
  int32_t pe_aligned; /* FlgPEAligned */
  int32_t rescued;    /* 1 when the hit came from AlignPairedRead */
  k4o_hit hit;
} k4o_pe_read;

int k4o_pe_insert_size(int pair_min_len, int pair_max_len, int pair_strand, uint8_t pe1_strand, uint32_t pe1_start,
                       uint32_t pe1_end, uint8_t pe2_strand, uint32_t pe2_start, uint32_t pe2_end); /* KAligner.cpp:2875-2918 */
int k4o_align_paired_read(const k4o_index* ix, int b3prime_extend, int antisense, uint32_t chrom_id,
                          uint32_t start_loci, uint32_t end_loci, int min_insert, int max_insert, int max_allowed_mm,
                          int read_len, const uint8_t* read, k4o_hit* out); /* SfxArray.cpp:8571-8767 + AdaptiveTrim :5561-5639 */
/* ... with MinChimericLen 15..99 (the `-c` mode of paired-end kalign): the placement that keeps the longest flank-trimmed stretch
 * of the mate (AdaptiveTrim's general rule, :5641-5795), ties by fewer mismatches, then first found; windows below 1000 loci are
 * scanned (:8731-8766), wider ones are seeded with exact cores of min(core_len, MinPutLen) bases every core_delta bases through
 * IterateExactsRange (:8685-8726, :3461-3553).  out->ext carries TrimLeft / TrimRight / the chimeric flag. */
int k4o_align_paired_read_x(const k4o_index* ix, int b3prime_extend, int antisense, uint32_t chrom_id, uint32_t start_loci,
                            uint32_t end_loci, int min_insert, int max_insert, int max_allowed_mm, int read_len,
                            int min_chimeric_len, int core_len, int core_delta, const uint8_t* read, k4o_hit* out);
/* CKAligner PE flow for n_pairs pairs: ProcCoredApprox (KAligner.cpp:10160-10239) then ProcessPairedEnds (:3159-3596).
 * out[2*i] = PE1, out[2*i+1] = PE2. kp->pe_mode / max_ml are forced to the PE values (1, 10). */
int k4o_kalign_pe_batch(const k4o_index* ix, const k4o_kalign_params* kp, const k4o_pe_params* pe, int64_t n_pairs,
                        const uint8_t* reads1, const uint64_t* offs1, const uint32_t* lens1, const uint8_t* reads2,
                        const uint64_t* offs2, const uint32_t* lens2, k4o_pe_read* out, int nthreads);

/* kalign's SNP calling, main CSV (CKAligner::ProcessSNPs KAligner.cpp:8168-8590 + OutputSNPs :7098-7760; k4oracle_snp.c):
 * nar / hits[i * hit_stride]: one reported alignment per read.  Returns the CSV text (release with k4o_free). */
char* k4o_snp_csv(const k4o_index* ix, int64_t n_reads, const int32_t* nar, const k4o_hit* hits, int hit_stride, const uint8_t* reads,
                  const uint64_t* offs, const uint32_t* lens, int min_snp_reads, double qvalue, double snp_nonref_pcnt,
                  int64_t* n_snps);
/* vcf != 0: the records of the VCF form instead (no header lines) */
char* k4o_snp_text(const k4o_index* ix, int vcf, int64_t n_reads, const int32_t* nar, const k4o_hit* hits, int hit_stride,
                   const uint8_t* reads, const uint64_t* offs, const uint32_t* lens, int min_snp_reads, double qvalue, double snp_nonref_pcnt,
                   int64_t* n_snps);
/* the coverage WIG kalign writes beside the SNP file (.covsegs.wig; AccumWIGCnts / CompleteWIGSpan, KAligner.cpp:6993-7085) */
char* k4o_snp_wig(const k4o_index* ix, int64_t n_reads, const int32_t* nar, const k4o_hit* hits, int hit_stride, const uint8_t* reads,
                  const uint64_t* offs, const uint32_t* lens, int min_snp_reads, double qvalue, double snp_nonref_pcnt);
/* the haplotype files kalign writes beside the SNP file (n_loci 2: .disnp.csv, 3: .trisnp.csv; KAligner.cpp:7767-8101) */
char* k4o_snp_haplotypes(const k4o_index* ix, int n_loci, int64_t n_reads, const int32_t* nar, const k4o_hit* hits, int hit_stride,
                         const uint8_t* reads, const uint64_t* offs, const uint32_t* lens, int min_snp_reads, double qvalue,
                         double snp_nonref_pcnt);
void k4o_free(void* p);

/* CKAligner::AssignMultiMatches (KAligner.cpp:5092-5258) with ProcAssignMultiMatches (:4944-5085) over the results of
 * k4o_align_batch run with pe_mode 1 (a read within the instance limit keeps its inst loci; unique ones are accepted):
 * ml_mode 3 = eMLuniq (`-r3`, cluster with uniquely aligned reads only), 4 = eMLmulti (`-r4`).  A multi-aligned read that
 * wins a locus becomes accepted with that locus in slot 0 (NumHits 1, LowHitInstances 1).  nthreads only shapes the
 * blocks GetClusterStartEnd (:4913-4940) hands out, which bound the reference's copy-the-previous-score shortcut.
 * Returns the number of reads assigned. */
int64_t k4o_assign_multi_matches(int ml_mode, int max_reads_len, int64_t n_reads, int max_ml, k4o_read_result* rr,
                                 k4o_hit* hits, int nthreads);

void k4o_revcomp(uint8_t* seq, int len);                                      /* SeqTrans.cpp:497-545 */

#ifdef __cplusplus
}
#endif
#endif
