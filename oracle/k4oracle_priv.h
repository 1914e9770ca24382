/* oracle/k4oracle_priv.h -- TEST INFRASTRUCTURE ONLY: what k4oracle.c and k4oracle_ext.c share. */
#ifndef K4ORACLE_PRIV_H
#define K4ORACLE_PRIV_H
#include "k4oracle.h"

#define K4O_HDR_SIZE 1224       /* sizeof(tsSfxHeaderV3), pack(4) */
#define K4O_BLOCK_HDR 20        /* tsSfxBlock up to SeqSuffix[0], pack(1) */
#define K4O_ENTRY_SIZE 111      /* sizeof(tsSfxEntry), pack(1) */
#define K4O_MAX_IDENT_NODES 1024000 /* cMaxNumIdentNodes, SfxArray.h:15 */
#define K4O_DFLT_MAX_ITER 50000 /* cDfltMaxIter, SfxArray.h:12 */

struct k4o_index {
  uint64_t n;  /* ConcatSeqLen: bases + one EOS per entry == number of SA elements */
  uint32_t el; /* SfxElSize 4|5 */
  uint8_t* seq;
  uint8_t* sa;
  uint32_t n_entries;
  k4o_entry* entries;
  char dataset[81];
  int max_iter;
  void* map; /* non-NULL when seq/sa point into an mmap of the file */
  size_t map_len;
  int owns;  /* seq/sa malloc'd by us */
};


typedef struct { uint64_t* slot; uint32_t cap; uint32_t* used; uint32_t n_used; } k4oi_idset; /* slot = id+1, 0 = empty */
void k4oi_idset_init(k4oi_idset* s);
void k4oi_idset_clear(k4oi_idset* s);
void k4oi_idset_free(k4oi_idset* s);
int k4oi_idset_insert(k4oi_idset* s, uint32_t id); /* 1 if new */
const k4o_entry* k4oi_map_chunk_hit2entry(const k4o_index* ix, uint64_t ofs); /* SfxArray.cpp:2609-2654 */
int k4oi_cmp_probe_targ(const uint8_t* probe, const uint8_t* targ, int len);   /* SfxArray.cpp:2508-2525 */
/* CKAligner::AlignRead for one read with kp's optional-phase arguments honoured (k4oracle_ext.c); returns the tHRslt */
int k4oi_align_read_ext(const k4o_index* ix, const k4o_kalign_params* kp, int mcl, int spm, const uint8_t* read, int read_len,
                        uint8_t* scratch, k4o_read_result* out, k4o_hit* hits);
#endif
