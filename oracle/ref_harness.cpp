// oracle/ref_harness.cpp -- TEST INFRASTRUCTURE ONLY (never shipped, never on the product path).
//
// Thin extern "C" shim over the *real* reference library (kit4b libkit4b), compiled by
// oracle/Makefile from the sources where they lie under /root/reference into oracle/_ref/libk4ref.so.
// It is used (a) to generate the golden vectors under tests/golden/ and (b) to validate the CPU
// restatement in oracle/k4oracle.c.  No reference source is copied into this repository; this file
// only *calls* the reference's public CSfxArray API:
//   CSfxArray::Open            libkit4b/SfxArray.h:528   (SfxArray.cpp:969)
//   CSfxArray::AddEntry        libkit4b/SfxArray.cpp:1518
//   CSfxArray::Finalise        libkit4b/SfxArray.cpp:1758
//   CSfxArray::SetTargBlock    libkit4b/SfxArray.cpp:1982
//   CSfxArray::SetMaxIter      libkit4b/SfxArray.cpp:1501
//   CSfxArray::InitialiseCoreKMers  libkit4b/SfxArray.cpp:8062
//   CSfxArray::AlignReads      libkit4b/SfxArray.h:614   (SfxArray.cpp:7838)
//   CSfxArray::AlignPairedRead libkit4b/SfxArray.h:880   (SfxArray.cpp:8571)
// The call sequence mirrors ngskit4b/kit4bax.cpp:760-906 (index build) and
// ngskit4b/KAligner.cpp:342-388,9353-9397 (open for alignment).
#include "libkit4b/commhdrs.h"

// globals every libkit4b host program must define (ngskit4b/ngskit4b.cpp does the same)
CDiagnostics gDiagnostics;
char gszProcName[_MAX_FNAME] = "k4ref";
CStopWatch gStopWatch;

extern "C" {

struct k4ref_hit {            // flat copy of tsHitLoci.Seg[0] (libkit4b/SfxArray.h:239-260)
  uint32_t chrom_id;          // Seg[0].ChromID   (1-based EntryID)
  uint64_t match_loci;        // Seg[0].MatchLoci (0-based in chrom)
  uint16_t match_len;         // Seg[0].MatchLen
  uint8_t strand;             // Seg[0].Strand '+'/'-'
  uint8_t mismatches;         // Seg[0].Mismatches
};

struct k4ref_handle {
  CSfxArray* sfx;
  tsIdentNode* nodes;
};

// Build a .sfx file from in-memory sequences (etSeqBase bytes: A=0,C=1,G=2,T=3,N=4).
int k4ref_build_sfx(const char* path, const char* dataset, int nseq, const char** names,
                    const uint8_t** seqs, const uint32_t* lens, int threads, int max_base_cmp_len) {
  CSfxArray* p = new CSfxArray;
  p->SetMaxQSortThreads(threads);
  int rslt = p->Open((char*)path, true, false, false);
  if (rslt != eBSFSuccess) { delete p; return rslt; }
  p->SetDescription((char*)"k4ref harness");
  p->SetTitle((char*)"k4ref");
  p->SetDatasetName((char*)dataset);
  uint64_t tot = 0;
  for (int i = 0; i < nseq; i++) tot += lens[i] + 1;
  p->SetInitalSfxAllocEls(tot);
  p->SetMaxBaseCmpLen(max_base_cmp_len > 0 ? max_base_cmp_len : 100000);  // kit4bax.cpp:859, default -k 100000
  for (int i = 0; i < nseq && rslt >= 0; i++)
    rslt = p->AddEntry((char*)names[i], (etSeqBase*)seqs[i], lens[i]);
  if (rslt >= 0) {
    rslt = p->Finalise();
    if (rslt < 0) p->Close();
  } else
    p->Close(false);
  delete p;
  return rslt;
}

void* k4ref_open(const char* path, int max_iter, int core_kmer_len) {
  CSfxArray* p = new CSfxArray;
  if (p->Open((char*)path, false, false, false) != eBSFSuccess) { delete p; return NULL; }
  if (p->SetTargBlock(1) < 0) { delete p; return NULL; }
  p->SetMaxIter(max_iter);
  if (core_kmer_len > 0) p->InitialiseCoreKMers(core_kmer_len);
  k4ref_handle* h = new k4ref_handle;
  h->sfx = p;
  h->nodes = new tsIdentNode[cMaxNumIdentNodes];
  return h;
}

void k4ref_close(void* vh) {
  k4ref_handle* h = (k4ref_handle*)vh;
  if (!h) return;
  delete[] h->nodes;
  delete h->sfx;
  delete h;
}

uint64_t k4ref_tot_seqs_len(void* vh) { return ((k4ref_handle*)vh)->sfx->GetTotSeqsLen(); }
int k4ref_num_entries(void* vh) { return ((k4ref_handle*)vh)->sfx->GetNumEntries(); }

// One CSfxArray::AlignReads call.  inst/low/nxt are In/Out exactly as in the reference
// (a fresh read starts them at 0, KAligner.cpp:9609-9611).  probe is mutated and restored by the callee.
int k4ref_align_reads(void* vh, int tot_mm, int core_len, int core_delta, int max_slides, int min_core_len,
                      int mm_delta, int strand, uint8_t* probe, int probe_len, int max_hits, int* inst,
                      int* low, int* nxt, k4ref_hit* out_hits) {
  k4ref_handle* h = (k4ref_handle*)vh;
  tsHitLoci* hits = new tsHitLoci[max_hits + 1];
  memset(hits, 0, sizeof(tsHitLoci) * (max_hits + 1));
  int rslt = h->sfx->AlignReads(0, 1, 0, tot_mm, core_len, core_delta, max_slides, min_core_len, mm_delta,
                                (eALStrand)strand, 0, 0, inst, low, nxt, (etSeqBase*)probe, probe_len,
                                max_hits, hits, cMaxNumIdentNodes, h->nodes);
  for (int i = 0; i < max_hits; i++) {
    out_hits[i].chrom_id = hits[i].Seg[0].ChromID;
    out_hits[i].match_loci = hits[i].Seg[0].MatchLoci;
    out_hits[i].match_len = hits[i].Seg[0].MatchLen;
    out_hits[i].strand = hits[i].Seg[0].Strand;
    out_hits[i].mismatches = hits[i].Seg[0].Mismatches;
  }
  delete[] hits;
  return rslt;
}

// One CSfxArray::AlignPairedRead call (mate rescue; KAligner.cpp:3372-3386 shows the caller's arguments).
int k4ref_align_paired_read(void* vh, int b3prime_extend, int antisense, uint32_t chrom_id, uint32_t start_loci,
                            uint32_t end_loci, int min_insert, int max_insert, int max_allowed_mm,
                            int min_hamming, int read_len, int min_chimeric_len, int core_len, int core_delta,
                            int max_slides, uint8_t* read, k4ref_hit* out_hit) {
  k4ref_handle* h = (k4ref_handle*)vh;
  tsHitLoci hit;
  memset(&hit, 0, sizeof(hit));
  int rslt = h->sfx->AlignPairedRead(b3prime_extend != 0, antisense != 0, chrom_id, start_loci, end_loci,
                                     min_insert, max_insert, max_allowed_mm, min_hamming, read_len,
                                     min_chimeric_len, core_len, core_delta, max_slides, (etSeqBase*)read, &hit);
  out_hit->chrom_id = hit.Seg[0].ChromID;
  out_hit->match_loci = hit.Seg[0].MatchLoci;
  out_hit->match_len = hit.Seg[0].MatchLen;
  out_hit->strand = hit.Seg[0].Strand;
  out_hit->mismatches = hit.Seg[0].Mismatches;
  return rslt;
}

// ---- the optional phases of AlignReads --------------------------------------------------------------------
struct k4ref_xhit {           // flat copy of one tsHitLoci (libkit4b/SfxArray.h:239-260), both segments
  uint32_t chrom_id;          // Seg[0]
  uint64_t match_loci;
  uint16_t match_len;
  uint8_t strand;
  uint8_t mismatches;
  uint16_t trim_left;
  uint16_t trim_right;
  uint8_t flags;              // bit 0 FlgChimeric, 1 FlgInDel, 2 FlgInsert, 3 FlgSplice, 4 FlgNonOrphan
  uint8_t seg1_mismatches;    // Seg[1]
  uint16_t score;
  uint32_t seg1_chrom_id;
  uint64_t seg1_match_loci;
  uint16_t seg1_match_len;
  uint16_t seg1_read_ofs;
};

// CSfxArray::AlignReads with every argument (SfxArray.cpp:7838): MinChimericLen, microInDelLen and MaxSpliceJunctLen included.
int k4ref_align_reads_ext(void* vh, int min_chimeric_len, int micro_indel_len, int max_splice_junct_len, int tot_mm,
                          int core_len, int core_delta, int max_slides, int min_core_len, int mm_delta, int strand,
                          uint8_t* probe, int probe_len, int max_hits, int* inst, int* low, int* nxt, k4ref_xhit* out_hits) {
  k4ref_handle* h = (k4ref_handle*)vh;
  tsHitLoci* hits = new tsHitLoci[max_hits + 2];
  memset(hits, 0, sizeof(tsHitLoci) * (max_hits + 2));
  int rslt = h->sfx->AlignReads(0, 1, min_chimeric_len, tot_mm, core_len, core_delta, max_slides, min_core_len, mm_delta,
                                (eALStrand)strand, micro_indel_len, max_splice_junct_len, inst, low, nxt, (etSeqBase*)probe,
                                probe_len, max_hits, hits, cMaxNumIdentNodes, h->nodes);
  for (int i = 0; i < max_hits; i++) {
    k4ref_xhit& o = out_hits[i];
    memset(&o, 0, sizeof(o));
    o.chrom_id = hits[i].Seg[0].ChromID;
    o.match_loci = hits[i].Seg[0].MatchLoci;
    o.match_len = hits[i].Seg[0].MatchLen;
    o.strand = hits[i].Seg[0].Strand;
    o.mismatches = hits[i].Seg[0].Mismatches;
    o.trim_left = hits[i].Seg[0].TrimLeft;
    o.trim_right = hits[i].Seg[0].TrimRight;
    o.flags = (uint8_t)(hits[i].FlgChimeric | (hits[i].FlgInDel << 1) | (hits[i].FlgInsert << 2) | (hits[i].FlgSplice << 3) |
                        (hits[i].FlgNonOrphan << 4));
    o.score = hits[i].Score;
    o.seg1_chrom_id = hits[i].Seg[1].ChromID;
    o.seg1_match_loci = hits[i].Seg[1].MatchLoci;
    o.seg1_match_len = hits[i].Seg[1].MatchLen;
    o.seg1_read_ofs = hits[i].Seg[1].ReadOfs;
    o.seg1_mismatches = hits[i].Seg[1].Mismatches;
  }
  delete[] hits;
  return rslt;
}

// CSfxArray::AlignPairedRead with the whole tsHitLoci back (trims and flags of a chimeric placement)
int k4ref_align_paired_read_x(void* vh, int b3prime_extend, int antisense, uint32_t chrom_id, uint32_t start_loci,
                              uint32_t end_loci, int min_insert, int max_insert, int max_allowed_mm, int min_hamming, int read_len,
                              int min_chimeric_len, int core_len, int core_delta, int max_slides, uint8_t* read, k4ref_xhit* o) {
  k4ref_handle* h = (k4ref_handle*)vh;
  tsHitLoci hit;
  memset(&hit, 0, sizeof(hit));
  int rslt = h->sfx->AlignPairedRead(b3prime_extend != 0, antisense != 0, chrom_id, start_loci, end_loci, min_insert, max_insert,
                                     max_allowed_mm, min_hamming, read_len, min_chimeric_len, core_len, core_delta, max_slides,
                                     (etSeqBase*)read, &hit);
  memset(o, 0, sizeof(*o));
  o->chrom_id = hit.Seg[0].ChromID;
  o->match_loci = hit.Seg[0].MatchLoci;
  o->match_len = hit.Seg[0].MatchLen;
  o->strand = hit.Seg[0].Strand;
  o->mismatches = hit.Seg[0].Mismatches;
  o->trim_left = hit.Seg[0].TrimLeft;
  o->trim_right = hit.Seg[0].TrimRight;
  o->flags = (uint8_t)(hit.FlgChimeric | (hit.FlgInDel << 1) | (hit.FlgInsert << 2) | (hit.FlgSplice << 3) | (hit.FlgNonOrphan << 4));
  return rslt;
}

// CSfxArray::AdaptiveTrim (SfxArray.cpp:5561) on caller-supplied sequences; out = {TrimSeqLen, TrimStart, TrimEnd, TrimMMs}
int k4ref_adaptive_trim(void* vh, uint32_t seq_len, uint8_t* probe, uint8_t* targ, uint32_t min_trim_len, uint32_t max_mm,
                        uint32_t min_flank, uint32_t* out4) {
  k4ref_handle* h = (k4ref_handle*)vh;
  return h->sfx->AdaptiveTrim(seq_len, (etSeqBase*)probe, (etSeqBase*)targ, 1, seq_len, min_trim_len, max_mm, min_flank, &out4[0],
                              &out4[1], &out4[2], &out4[3]);
}

}  // extern "C"
