// kit4b_amd/csrc/k4_internal.h -- shared between the translation units of libk4sfx.so (not installed).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <thread>
#include <vector>
#include "../../include/k4sfx.h"

// ---- HBM layout of the index --------------------------------------------------------------------------
// ref2      2-bit packed reference, MSB-first inside 32-bit words: base i lives in bits (30 - 2*(i & 15)) of
//           word (i >> 4); non-ACGT symbols (N=4, EOS=7) are stored as 0 and described by the exception data.
//           K4_PAD_WORDS zero words sit in front of base 0 and behind the last base, so a window that starts up
//           to K4_PAD_BASES before 0 or runs past the end can be fetched without bounds tests.
// excbm     one bit per 256-base block (K4_EXC_SHIFT): set when the block holds a non-ACGT symbol.  1.5 MB for 3 Gbp,
//           so it lives in every XCD's L2; a read window spans at most a few blocks -> one 8-byte fetch per probe.
// excblk    sorted list of the flagged block numbers; excnib holds, for flagged block r, its 256 symbols as
//           nibbles (32 words, symbol j in bits 4*(j&7) of word r*32 + (j>>3)) -- the exact 4-bit reference.
// sa        the .sfx suffix array unchanged: 4- or 5-byte little-endian elements.
// ktab      direct-address table over the first k bases, 4^k + 1 entries {lb, pos0, sig} (12 bytes; 16 bytes when
//           concat_len >= 2^32: 64-bit lb, then 40-bit pos0 and a 12-base sig sharing a word): lb[c] = number of suffixes that sort before the k-mer with code c
//           (first base most significant), so the bucket of c is SA[lb[c] .. lb[c+1]); pos0[c] = SA[lb[c]], the
//           offset of the bucket's first suffix, which saves the dependent SA fetch for the first probe of a lookup;
//           sig[c] = the 16 bases after the k-mer in that suffix: a core that disagrees with it cannot match a
//           single-suffix bucket, so no probe is issued.
// entries   start/end offsets and ids of the chromosomes (tsSfxEntry), sorted by start.
#define K4_EXC_SHIFT 8            // log2 of the exception-bitmap block size in bases
#define K4_EXC_BLOCK (1 << K4_EXC_SHIFT)
#define K4_SUP_WORDS 512           // LDS-resident coarse exception bitmap: 16 Kbit, one bit per 2^sup_shift bases
#define K4_LDS_ENTRIES 128         // chromosome tables up to this size are searched in LDS
#define K4_PAD_BASES 2048
#define K4_PAD_WORDS (K4_PAD_BASES / 16)
#define K4_MAX_FAST_READ_LEN 512   // longer reads run in the general kernel
#define K4_MAX_READ_LEN 4096       // cMaxSeqLen is 2000 (KAligner.h:115)
#define K4_PROF_SLOTS 32          // u64 slots behind k4_counters for builds with -DK4_SLOW_PROF (tools/slow_prof.py)
#ifndef K4_DEEP_BUCKET
#define K4_DEEP_BUCKET 24          // k-mer table buckets deeper than this mark repeat families (k4_align.hip: K4_DEFER_BUCKET)
#endif
#define K4_DEDUP_CAP 6             // distinct candidates per strand pass kept by the fast kernel
#define K4_MAX_IDENT_NODES 1024000 // cMaxNumIdentNodes, libkit4b/SfxArray.h:15

struct K4DevIndex {
  const uint32_t* ref2;   // points at the word holding base 0
  const uint32_t* excbm;
  const uint32_t* excsup;  // K4_SUP_WORDS words: bit b set when bases [b << sup_shift, (b+1) << sup_shift) hold an exception
  const uint32_t* excblk;
  const uint32_t* excnib;
  const uint8_t* sa;
  const void* ktab;
  const uint64_t* ent_start;
  const uint64_t* ent_end;
  const uint32_t* ent_id;
  uint64_t n;        // ConcatSeqLen
  uint32_t n_exc;
  uint32_t n_entries;
  uint32_t el;       // 4 | 5
  uint32_t k;        // k-mer table length
  uint32_t ktab64;   // table entries are 64-bit
  uint32_t sup_shift; // log2 bases per bit of excsup (== K4_EXC_SHIFT when the whole fine bitmap fits)
  int32_t max_iter;
};

struct K4Workspace {
  int64_t cap_reads = 0;
  int32_t cap_len = 0;
  int32_t cap_hits = 0;
  uint32_t* ids[2] = {nullptr, nullptr};   // survivors of a step: read ids ...
  uint64_t* rows[2] = {nullptr, nullptr};  // ... and their packed rows (forward + reverse-complement words)
  uint32_t* slow_list = nullptr; // read ids for the general kernel
  uint8_t* slow_step = nullptr;  // and the phase ordinal at which each left the fast path
  uint32_t* huge_list = nullptr; // reads that outgrew the small dedupe tables of the first general pass
  uint8_t* huge_step = nullptr;
  uint32_t* ctl = nullptr;       // [0] slow count, [1] slow head, [2+t] survivor count of step t
  uint8_t* slow_probe = nullptr; // per slow lane: probe bytes scratch
  uint64_t* slow_hash = nullptr; // per slow lane: open-addressing table of (generation<<32 | TargSeqID)
  uint32_t slow_lanes = 0;
  uint32_t slow_hash_cap = 0;    // entries per lane (power of two)
  // staging for the host-pointer entry points
  uint8_t* d_reads = nullptr; size_t d_reads_cap = 0;
  uint64_t* d_offs = nullptr; uint32_t* d_lens = nullptr;
  int32_t* d_out4 = nullptr;     // rslt/inst/low/nxt or k4_read_result
  k4_hit* d_hits = nullptr;
  int64_t stage_reads = 0; int32_t stage_hits = 0;
  // small host-pointer batches (the facade's AlignReads groups): one pinned block up, one down.  [0, K4_SMALL_STAGE/2):
  // offs | lens | reads; [K4_SMALL_STAGE/2, K4_SMALL_STAGE): results | hits -- same layout in h_small and d_small
  uint8_t* h_small = nullptr; uint8_t* d_small = nullptr;
  // where the current host-pointer batch lives on the device (d_small or the d_* buffers above)
  const uint8_t* c_reads = nullptr; const uint64_t* c_offs = nullptr; const uint32_t* c_lens = nullptr;
  int32_t* c_out = nullptr; k4_hit* c_hits = nullptr; bool c_small = false;
};

struct k4_index {
  int device = 0;
  K4DevIndex d{};
  // owning pointers
  uint32_t* ref2_alloc = nullptr;
  uint32_t* excbm = nullptr;
  uint32_t* excsup = nullptr;
  uint32_t* excblk = nullptr;
  uint32_t* excnib = nullptr;
  uint8_t* sa = nullptr;
  bool owns_sa = true;
  void* ktab = nullptr;
  uint64_t* ent_start = nullptr;
  uint64_t* ent_end = nullptr;
  uint32_t* ent_id = nullptr;
  uint64_t* counters = nullptr; // device k4_counters
  // FASTQ qualities (kalign -g, etFQMethod): 3 = ignored (the default); 0 Sanger / 1 Illumina 1.3+ / 2 Solexa: the parser puts the
  // scaled 4-bit score of every base into bits 4..7 of its read byte (the reference's own in-memory form), the SAM / BAM writers emit it
  int q_method = 3;
  uint8_t* d_qlut = nullptr;    // 256 bytes: quality character -> 4-bit score of the chosen method
  double deep_bucket_frac = 0;  // share of the suffixes that sit in k-mer buckets deeper than K4_DEEP_BUCKET (k4_index.hip)
  std::vector<k4_entry> entries;
  std::string dataset;
  std::string description, title;  // header text for k4_write_sfx (k4_set_description)
  std::vector<uint8_t> raw_header;  // the file's tsSfxHeaderV3 when the index was opened from a .sfx
  uint64_t tot_seqs_len = 0;
  uint64_t device_bytes = 0;
  std::string err;
  K4Workspace ws;
  // paired-end pass (k4_pe.hip): both ends' SE results, their hit slots, the orphan list, {orphan count, error flag}
  k4_read_result* pe_rr = nullptr;
  k4_hit* pe_hits = nullptr;
  uint32_t* pe_list = nullptr;
  uint32_t* pe_ctl = nullptr;
  int64_t pe_cap_pairs = 0;
  int32_t pe_cap_hits = 0;
  // staging of k4_mate_rescue_batch (grow-only)
  void *rs_tasks = nullptr, *rs_reads = nullptr, *rs_res = nullptr, *rs_hits = nullptr;
  size_t rs_cap_tasks = 0, rs_cap_reads = 0;
  // k4_open_async: the arrays are uploaded and the device structures built by this thread; k4_open_wait joins it
  std::thread loader;
  int load_rc = 0;
  double load_seconds = 0;  // what that thread took
  hipStream_t stream = nullptr; // internal stream for the host-pointer entry points
  bool timing = false;          // bracket k4k_align_fast with events
  std::vector<hipEvent_t> ev0, ev1, ev2;  // before the step kernels, behind them, behind the general kernel's passes
  size_t ev_used = 0;
};

void k4_set_global_error(const char* fmt, ...);
int k4_fail(k4_index* ix, int code, const char* fmt, ...);
int k4_check_hip(k4_index* ix, hipError_t e, const char* what);

// k4_index.hip
int k4i_build_device_structures(k4_index* ix, const void* d_seq_bytes, int kmer_k);
// k4_align.hip
#define K4_SMALL_STAGE (4u << 20)
#define K4_SMALL_READS 4096
int k4i_kalign_batch_dev(k4_index* ix, const k4_kalign_params* p, int64_t n, int32_t max_len, const void* d_reads,
                         const void* d_offs, const void* d_lens, void* d_out, void* d_hits, void* stream, int sparse_hits);
// k4_sabuild.hip
int k4i_build_sa(uint64_t n, uint32_t el, const uint8_t* d_seq, uint8_t* d_sa, int device, std::string* err);

#define K4_HIP(ix, call)                                   \
  do {                                                     \
    int _rc = k4_check_hip((ix), (call), #call);           \
    if (_rc != K4_OK) return _rc;                          \
  } while (0)
