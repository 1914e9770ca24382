/* oracle/k4oracle.c -- TEST INFRASTRUCTURE ONLY (see k4oracle.h).
 *
 * Plain-C, byte-per-base restatement of the reference CPU algorithm for the kalign hot path:
 *   .sfx container            libkit4b/SfxArray.h:95-123,191-223, SfxArray.cpp:380-620,629-825
 *   suffix order              libkit4b/SfxArray.cpp:9779-9834 (QSortSeqCmp32/40)
 *   SfxOfsToLoci              libkit4b/SfxArray.cpp:49-60
 *   MapChunkHit2Entry         libkit4b/SfxArray.cpp:2609-2654
 *   LocateFirstExact          libkit4b/SfxArray.cpp:7938-8058
 *   LocateCoreMultiples       libkit4b/SfxArray.cpp:5806-6369 (default, non-chimeric, base-space branch)
 *   AlignReads                libkit4b/SfxArray.cpp:7838-7933 (phases; InDel/splice/chimeric phases are off)
 *   CKAligner::AlignRead      ngskit4b/KAligner.cpp:9583-10105 (parameter derivation + NAR classification)
 * It is deliberately simple and sequential per read; threads only split reads.
 * Not restated (out of scope): bisulfite, colourspace, chimeric trimming, microInDel, splice.
 */
#define _GNU_SOURCE
#include "k4oracle_priv.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ctype.h>
#include <pthread.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <fcntl.h>
#include <unistd.h>

/* ------------------------------------------------------------------------------------------------ */
static uint64_t rd_u64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }
static uint32_t rd_u32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
static uint16_t rd_u16(const uint8_t* p) { uint16_t v; memcpy(&v, p, 2); return v; }

/* SfxOfsToLoci, SfxArray.cpp:49-60 */
int64_t k4o_sa_at(const k4o_index* ix, int64_t i) {
  const uint8_t* p = ix->sa + (uint64_t)i * ix->el;
  uint64_t v = rd_u32(p);
  if (ix->el == 5) v |= (uint64_t)p[4] << 32;
  return (int64_t)v;
}

static void sa_put(uint8_t* sa, uint32_t el, uint64_t i, uint64_t v) {
  uint8_t* p = sa + i * el;
  uint32_t lo = (uint32_t)v;
  memcpy(p, &lo, 4);
  if (el == 5) p[4] = (uint8_t)(v >> 32);
}

uint64_t k4o_concat_len(const k4o_index* ix) { return ix->n; }
uint32_t k4o_el_size(const k4o_index* ix) { return ix->el; }
const uint8_t* k4o_seq(const k4o_index* ix) { return ix->seq; }
const uint8_t* k4o_sa_bytes(const k4o_index* ix) { return ix->sa; }
uint32_t k4o_num_entries(const k4o_index* ix) { return ix->n_entries; }
const k4o_entry* k4o_entries(const k4o_index* ix) { return ix->entries; }
void k4o_set_max_iter(k4o_index* ix, int max_iter) { ix->max_iter = max_iter > 0 ? max_iter : 0; }

uint64_t k4o_tot_seqs_len(const k4o_index* ix) {
  uint64_t t = 0;
  for (uint32_t i = 0; i < ix->n_entries; i++) t += ix->entries[i].seq_len;
  return t;
}

/* CUtility::GenHash16, libkit4b/Utility.cpp:402-420 */
static uint16_t gen_hash16(const char* name) {
  int h = 19937;
  if (!name || !name[0]) return 0;
  for (; *name; name++) {
    h = (h ^ (int)tolower((unsigned char)*name)) * 3119;
    h ^= (h >> 13);
    h &= 0xffff;
  }
  if (h == 0) h = 19937;
  return (uint16_t)h;
}

/* ---- .sfx reader: Disk2Hdr :629, Disk2Entries :714, Disk2SfxBlock :1915 ------------------------- */
k4o_index* k4o_open(const char* path, char* err, size_t errlen) {
#define FAIL(...) do { if (err) snprintf(err, errlen, __VA_ARGS__); if (map && map != MAP_FAILED) munmap(map, len); if (fd >= 0) close(fd); free(ix); return NULL; } while (0)
  k4o_index* ix = NULL;
  void* map = NULL;
  size_t len = 0;
  int fd = open(path, O_RDONLY);
  if (fd < 0) FAIL("unable to open %s", path);
  struct stat st;
  if (fstat(fd, &st) != 0) FAIL("unable to stat %s", path);
  len = (size_t)st.st_size;
  if (len < K4O_HDR_SIZE) FAIL("%s too short for a .sfx header", path);
  map = mmap(NULL, len, PROT_READ, MAP_PRIVATE, fd, 0);
  if (map == MAP_FAILED) FAIL("mmap failed on %s", path);
  const uint8_t* f = (const uint8_t*)map;
  if (tolower(f[0]) != 's' || tolower(f[1]) != 'f' || tolower(f[2]) != 'x' || f[3] < '3' || f[3] > '5')
    FAIL("%s: bad magic, not a kit4b suffix array file", path);
  uint32_t ver = rd_u32(f + 4);
  if (ver < 4 || ver > 5) FAIL("%s: structure version %u not supported by the oracle (4..5)", path, ver);
  uint32_t attr = rd_u32(f + 8);
  if (attr & 3) FAIL("%s: bisulfite/colourspace indexes are out of scope", path);
  uint64_t entries_ofs = rd_u64(f + 20);
  uint32_t entries_size = rd_u32(f + 28);
  uint32_t n_blocks = rd_u32(f + 32);
  uint64_t block_ofs = rd_u64(f + 44);
  if (n_blocks != 1) FAIL("%s: NumSfxBlocks=%u (expected 1)", path, n_blocks);
  if (block_ofs + K4O_BLOCK_HDR > len) FAIL("%s: block offset beyond file", path);
  ix = (k4o_index*)calloc(1, sizeof(*ix));
  memcpy(ix->dataset, f + 52, 80);
  const uint8_t* b = f + block_ofs;
  ix->n = rd_u64(b + 8);
  ix->el = rd_u32(b + 16);
  if (ix->el != 4 && ix->el != 5) FAIL("%s: SfxElSize %u", path, ix->el);
  if (block_ofs + K4O_BLOCK_HDR + ix->n + ix->n * ix->el > len) FAIL("%s: block truncated", path);
  ix->seq = (uint8_t*)(b + K4O_BLOCK_HDR);
  ix->sa = (uint8_t*)(b + K4O_BLOCK_HDR + ix->n);
  if (entries_ofs == 0 || entries_ofs + entries_size > len || entries_size < 8) FAIL("%s: bad entries block", path);
  const uint8_t* e = f + entries_ofs;
  ix->n_entries = rd_u32(e);
  if (8 + (uint64_t)ix->n_entries * K4O_ENTRY_SIZE > entries_size) FAIL("%s: entries block truncated", path);
  ix->entries = (k4o_entry*)calloc(ix->n_entries ? ix->n_entries : 1, sizeof(k4o_entry));
  for (uint32_t i = 0; i < ix->n_entries; i++) {
    const uint8_t* p = e + 8 + (size_t)i * K4O_ENTRY_SIZE;
    k4o_entry* d = &ix->entries[i];
    d->entry_id = rd_u32(p);
    d->fblock_id = rd_u32(p + 4);
    memcpy(d->name, p + 8, 81);
    d->name[80] = 0;
    d->name_hash = rd_u16(p + 89);
    d->seq_len = rd_u32(p + 91);
    d->start_ofs = rd_u64(p + 95);
    d->end_ofs = rd_u64(p + 103);
  }
  ix->max_iter = K4O_DFLT_MAX_ITER;
  ix->map = map;
  ix->map_len = len;
  close(fd);
  return ix;
#undef FAIL
}

void k4o_close(k4o_index* ix) {
  if (!ix) return;
  if (ix->map) munmap(ix->map, ix->map_len);
  if (ix->owns) { free(ix->seq); free(ix->sa); }
  free(ix->entries);
  free(ix);
}

k4o_index* k4o_from_parts(uint64_t n, uint32_t el, uint8_t* seq, uint8_t* sa, uint32_t n_entries,
                          const k4o_entry* entries, const char* dataset) {
  k4o_index* ix = (k4o_index*)calloc(1, sizeof(*ix));
  ix->n = n; ix->el = el; ix->seq = seq; ix->sa = sa; ix->n_entries = n_entries;
  ix->entries = (k4o_entry*)calloc(n_entries ? n_entries : 1, sizeof(k4o_entry));
  memcpy(ix->entries, entries, sizeof(k4o_entry) * n_entries);
  if (dataset) strncpy(ix->dataset, dataset, 80);
  ix->max_iter = K4O_DFLT_MAX_ITER;
  return ix;
}

/* ---- suffix sort: order of QSortSeqCmp32/40, SfxArray.cpp:9779-9834 ------------------------------
 * bytewise on the low nibble, A<C<G<T<N<EOS(7); stops after the first EOS (both reach it together => equal).
 * The reference's parallel quicksort leaves tied suffixes in arbitrary order; we break ties by offset. */
static const uint8_t* g_sort_seq; /* set before sorting; sort workers only read it */

static int sfx_cmp(uint64_t a, uint64_t b) {
  const uint8_t* p = g_sort_seq + a;
  const uint8_t* q = g_sort_seq + b;
  for (;;) {
    uint8_t x = *p++ & 0x0f, y = *q++ & 0x0f;
    if (x != y) return x < y ? -1 : 1;
    if (x == K4O_EOS) break;
  }
  return a < b ? -1 : (a > b ? 1 : 0);
}
static int cmp_u64_sfx(const void* a, const void* b) { return sfx_cmp(*(const uint64_t*)a, *(const uint64_t*)b); }

typedef struct { uint64_t* idx; uint64_t* bstart; int nb; int next; pthread_mutex_t mtx; } sort_job;

static void* sort_worker(void* arg) {
  sort_job* j = (sort_job*)arg;
  for (;;) {
    pthread_mutex_lock(&j->mtx);
    int b = j->next++;
    pthread_mutex_unlock(&j->mtx);
    if (b >= j->nb) break;
    uint64_t cnt = j->bstart[b + 1] - j->bstart[b];
    if (cnt > 1) qsort(j->idx + j->bstart[b], cnt, sizeof(uint64_t), cmp_u64_sfx);
  }
  return NULL;
}

static uint32_t bucket_key(const uint8_t* s) { /* first 3 symbols, constant after an EOS */
  uint32_t k = 0;
  int eos = 0;
  for (int i = 0; i < 3; i++) {
    uint32_t c = eos ? 0 : (uint32_t)(s[i] & 0x0f);
    if (c == K4O_EOS) eos = 1;
    k = (k << 3) | (c & 7);
  }
  return k;
}

/* AddEntry :1518-1753 + Finalise/QSortSeq :1758,9739 */
k4o_index* k4o_build(int nseq, const char* const* names, const uint8_t* const* seqs, const uint32_t* lens,
                     const char* dataset, int force_el, int nthreads) {
  uint64_t n = 0;
  for (int i = 0; i < nseq; i++) n += (uint64_t)lens[i] + 1;
  k4o_index* ix = (k4o_index*)calloc(1, sizeof(*ix));
  ix->n = n;
  ix->el = force_el ? (uint32_t)force_el : (n < 4000000000ULL ? 4 : 5); /* cThres8ByteSfxEls, SfxArray.cpp:906-909 */
  ix->seq = (uint8_t*)malloc(n + 16);
  memset(ix->seq + n, K4O_EOS, 16); /* slack so bucket_key may read 2 bytes past the final EOS */
  ix->sa = (uint8_t*)malloc(n * ix->el + 8);
  ix->owns = 1;
  ix->n_entries = (uint32_t)nseq;
  ix->entries = (k4o_entry*)calloc(nseq ? nseq : 1, sizeof(k4o_entry));
  ix->max_iter = K4O_DFLT_MAX_ITER;
  if (dataset) strncpy(ix->dataset, dataset, 80);
  uint64_t ofs = 0;
  for (int i = 0; i < nseq; i++) {
    k4o_entry* e = &ix->entries[i];
    e->entry_id = (uint32_t)i + 1;
    e->fblock_id = 1;
    strncpy(e->name, names[i], 80);
    e->name_hash = gen_hash16(names[i]);
    e->seq_len = lens[i];
    e->start_ofs = ofs;
    e->end_ofs = ofs + lens[i] - 1;
    for (uint32_t k = 0; k < lens[i]; k++) ix->seq[ofs + k] = seqs[i][k] & ~0x08; /* strip cRptMskFlg */
    ix->seq[ofs + lens[i]] = K4O_EOS;
    ofs += (uint64_t)lens[i] + 1;
  }
  /* bucket by the first 3 symbols, then comparison-sort each bucket */
  enum { NB = 512 };
  uint64_t* bstart = (uint64_t*)calloc(NB + 1, sizeof(uint64_t));
  uint64_t* idx = (uint64_t*)malloc(sizeof(uint64_t) * (n ? n : 1));
  for (uint64_t i = 0; i < n; i++) bstart[bucket_key(ix->seq + i) + 1]++;
  for (int b = 0; b < NB; b++) bstart[b + 1] += bstart[b];
  uint64_t* fill = (uint64_t*)malloc(sizeof(uint64_t) * NB);
  memcpy(fill, bstart, sizeof(uint64_t) * NB);
  for (uint64_t i = 0; i < n; i++) idx[fill[bucket_key(ix->seq + i)]++] = i;
  free(fill);
  g_sort_seq = ix->seq;
  sort_job job = { idx, bstart, NB, 0, PTHREAD_MUTEX_INITIALIZER };
  if (nthreads < 1) nthreads = 1;
  if (nthreads > 64) nthreads = 64;
  pthread_t th[64];
  for (int t = 0; t < nthreads; t++) pthread_create(&th[t], NULL, sort_worker, &job);
  for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
  for (uint64_t i = 0; i < n; i++) sa_put(ix->sa, ix->el, i, idx[i]);
  free(idx);
  free(bstart);
  return ix;
}

/* ---- .sfx writer: Hdr2Disk :380, SfxBlock2Disk :499, Entries2Disk :583 --------------------------- */
int k4o_write(const k4o_index* ix, const char* path) {
  FILE* fp = fopen(path, "wb");
  if (!fp) return -1;
  uint8_t hdr[K4O_HDR_SIZE];
  memset(hdr, 0, sizeof(hdr));
  memcpy(hdr, "sfx5", 4);
  uint32_t ver = 5, attr = 0, nblocks = 1;
  uint64_t block_ofs = K4O_HDR_SIZE;
  uint64_t block_size = K4O_BLOCK_HDR + ix->n + ix->n * ix->el;
  uint64_t entries_ofs = block_ofs + block_size;
  uint32_t entries_size = 8 + K4O_ENTRY_SIZE * ix->n_entries;
  uint64_t file_len = entries_ofs + entries_size;
  memcpy(hdr + 4, &ver, 4);
  memcpy(hdr + 8, &attr, 4);
  memcpy(hdr + 12, &file_len, 8);
  memcpy(hdr + 20, &entries_ofs, 8);
  memcpy(hdr + 28, &entries_size, 4);
  memcpy(hdr + 32, &nblocks, 4);
  memcpy(hdr + 36, &block_size, 8);
  memcpy(hdr + 44, &block_ofs, 8);
  strncpy((char*)hdr + 52, ix->dataset, 80);
  strncpy((char*)hdr + 133, "k4oracle", 1023);
  strncpy((char*)hdr + 1157, "k4oracle", 63);
  fwrite(hdr, 1, sizeof(hdr), fp);
  uint8_t bh[K4O_BLOCK_HDR];
  uint32_t bid = 1;
  memcpy(bh, &bid, 4);
  memcpy(bh + 4, &ix->n_entries, 4);
  memcpy(bh + 8, &ix->n, 8);
  memcpy(bh + 16, &ix->el, 4);
  fwrite(bh, 1, sizeof(bh), fp);
  fwrite(ix->seq, 1, ix->n, fp);
  fwrite(ix->sa, 1, ix->n * ix->el, fp);
  uint32_t ne[2] = { ix->n_entries, ix->n_entries };
  fwrite(ne, 4, 2, fp);
  for (uint32_t i = 0; i < ix->n_entries; i++) {
    uint8_t e[K4O_ENTRY_SIZE];
    const k4o_entry* s = &ix->entries[i];
    memset(e, 0, sizeof(e));
    memcpy(e, &s->entry_id, 4);
    memcpy(e + 4, &s->fblock_id, 4);
    strncpy((char*)e + 8, s->name, 80);
    memcpy(e + 89, &s->name_hash, 2);
    memcpy(e + 91, &s->seq_len, 4);
    memcpy(e + 95, &s->start_ofs, 8);
    memcpy(e + 103, &s->end_ofs, 8);
    fwrite(e, 1, sizeof(e), fp);
  }
  int bad = ferror(fp);
  fclose(fp);
  return bad ? -1 : 0;
}

/* ---- MapChunkHit2Entry, SfxArray.cpp:2609-2654 ---------------------------------------------------- */
#define map_chunk_hit2entry k4oi_map_chunk_hit2entry
const k4o_entry* k4oi_map_chunk_hit2entry(const k4o_index* ix, uint64_t ofs) {
  int64_t lo = 0, hi = (int64_t)ix->n_entries - 1;
  while (hi >= lo) {
    int64_t mid = (hi + lo) / 2;
    const k4o_entry* e = &ix->entries[mid];
    uint32_t blk = e->fblock_id & 0xff;
    if (blk > 1) { hi = mid - 1; continue; }
    if (blk < 1) { lo = mid + 1; continue; }
    if (e->start_ofs <= ofs && e->end_ofs >= ofs) return e;
    if (e->start_ofs > ofs) hi = mid - 1;
    else lo = mid + 1;
  }
  return NULL; /* ofs sits on an EOS separator */
}

/* ---- ReverseComplement, SeqTrans.cpp:497-545 (N/InDel/Undef stay; anything else stops complementing) */
void k4o_revcomp(uint8_t* s, int len) {
  if (len < 1) return;
  for (int i = 0; i < len; i++) {
    uint8_t b = s[i], flg = b & 0x18;
    b &= ~0x18;
    if (b <= K4O_T) s[i] = (uint8_t)((3 - b) | flg);
    else if (b == K4O_N || b == 5 || b == 6) continue;
    else break;
  }
  for (int i = 0, j = len - 1; i < j; i++, j--) { uint8_t t = s[i]; s[i] = s[j]; s[j] = t; }
}

/* probe vs suffix over len bases: target EOS => probe < target.  CmpProbeTarg, SfxArray.cpp:2508-2525 */
#define cmp_probe_targ k4oi_cmp_probe_targ
int k4oi_cmp_probe_targ(const uint8_t* probe, const uint8_t* targ, int len) {
  for (int i = 0; i < len; i++) {
    uint8_t t = targ[i] & 0x0f;
    if (t == K4O_EOS) return -1;
    uint8_t p = probe[i] & 0x0f;
    if (p > t) return 1;
    if (p < t) return -1;
  }
  return 0;
}

/* ---- LocateFirstExact, SfxArray.cpp:7938-8058: returns index+1 of the lowest matching suffix, 0 if none.
 * (The k-mer memo fast path :7959-7964 returns the same value by construction, :8123-8224.) */
int64_t k4o_locate_first_exact(const k4o_index* ix, const uint8_t* probe, int probe_len, int64_t lo, int64_t hi,
                               k4o_counters* ctr) {
  if (ctr) ctr->n_lookup++;
  do {
    int64_t mid = (lo + hi) / 2;
    int c = cmp_probe_targ(probe, ix->seq + k4o_sa_at(ix, mid), probe_len);
    if (ctr) ctr->n_probe++;
    if (c == 0) {
      if (mid == 0 || lo == mid) return mid + 1;
      int64_t mark = mid;
      for (;;) { /* narrow to the lowest matching index */
        if (c == 0) {
          mark = mid;
          if (mark == 0) return 1;
          hi = mid - 1;
        }
        mid = (lo + hi) / 2;
        c = cmp_probe_targ(probe, ix->seq + k4o_sa_at(ix, mid), probe_len);
        if (ctr) ctr->n_probe++;
        if (c == 0) continue;
        lo = mid + 1;
        if (lo == mark) return mark + 1;
      }
    }
    if (c < 0) {
      if (mid == 0) break;
      hi = mid - 1;
    } else
      lo = mid + 1;
  } while (hi >= lo);
  return 0;
}

/* ---- dedupe set for TargSeqIDs: per strand pass, SfxArray.cpp:5845,5946,6037-6055 ---------------- */
typedef k4oi_idset idset;
void k4oi_idset_init(k4oi_idset* s) {
  s->cap = 1u << 12;
  s->slot = (uint64_t*)calloc(s->cap, sizeof(uint64_t));
  s->used = (uint32_t*)malloc(sizeof(uint32_t) * s->cap);
  s->n_used = 0;
}
void k4oi_idset_clear(k4oi_idset* s) {
  for (uint32_t i = 0; i < s->n_used; i++) s->slot[s->used[i]] = 0;
  s->n_used = 0;
}
void k4oi_idset_free(k4oi_idset* s) { free(s->slot); free(s->used); }
#define idset_init k4oi_idset_init
#define idset_clear k4oi_idset_clear
#define idset_free k4oi_idset_free
#define idset_insert k4oi_idset_insert
int k4oi_idset_insert(k4oi_idset* s, uint32_t id) { /* 1 if new */
  if (s->n_used * 2 >= s->cap) { /* grow */
    uint32_t ncap = s->cap * 2;
    uint64_t* nslot = (uint64_t*)calloc(ncap, sizeof(uint64_t));
    uint32_t* nused = (uint32_t*)malloc(sizeof(uint32_t) * ncap);
    uint32_t nn = 0;
    for (uint32_t i = 0; i < s->n_used; i++) {
      uint64_t v = s->slot[s->used[i]];
      uint32_t h = ((uint32_t)v * 2654435761u) & (ncap - 1);
      while (nslot[h]) h = (h + 1) & (ncap - 1);
      nslot[h] = v;
      nused[nn++] = h;
    }
    free(s->slot); free(s->used);
    s->slot = nslot; s->used = nused; s->cap = ncap; s->n_used = nn;
  }
  uint64_t v = (uint64_t)id + 1;
  uint32_t h = ((uint32_t)v * 2654435761u) & (s->cap - 1);
  while (s->slot[h]) {
    if (s->slot[h] == v) return 0;
    h = (h + 1) & (s->cap - 1);
  }
  s->slot[h] = v;
  s->used[s->n_used++] = h;
  return 1;
}

static void store_hit(k4o_hit* h, const k4o_entry* e, int64_t left, char strand, int probe_len, int mm) {
  memset(h, 0, sizeof(*h));
  h->chrom_id = e->entry_id;
  h->match_loci = (uint32_t)((uint64_t)left - e->start_ofs);
  h->match_len = (uint16_t)probe_len;
  h->strand = (uint8_t)strand;
  h->mismatches = (uint8_t)mm;
}

/* ---- LocateCoreMultiples, SfxArray.cpp:5806-6369 (MinChimericLen == 0 branch) --------------------- */
int k4o_locate_core_multiples(const k4o_index* ix, int max_tot_mm, int core_len, int core_delta, int max_slides,
                              int mm_delta, int strand, int* p_inst, int* p_low, int* p_nxt, uint8_t* probe,
                              int probe_len, int max_hits, k4o_hit* hits, k4o_counters* ctr) {
  if (ix->n == 0) return -1;
  /* :5889-5895 carried-in state that cannot improve */
  if (*p_inst > max_hits && *p_low == 0) return K4O_HR_HITINSTS;
  if (*p_inst >= 1 && *p_low == 0 && (*p_nxt - *p_low) < mm_delta) return K4O_HR_MMDELTA;

  int inst, low, nxt;
  if (*p_inst <= 0 || *p_low < 0 || *p_nxt < 0) { /* :5902-5907 */
    inst = *p_inst = 0;
    low = *p_low = max_tot_mm + mm_delta + 1;
    nxt = *p_nxt = low;
  } else {
    inst = *p_inst; low = *p_low; nxt = *p_nxt;
  }
  int cur_hit = inst < max_hits ? inst : -1; /* pCurHit, :5915-5918 */
  const int max_iter = ix->max_iter;
  const int64_t n = (int64_t)ix->n;
  char cur_strand = '+';
  if (strand == K4O_STRAND_CRICK) { k4o_revcomp(probe, probe_len); cur_strand = '-'; }
  idset ids;
  idset_init(&ids);

  do {
    int cur_delta = core_delta;
    int slides = 0;
    uint32_t n_nodes = 0;
    idset_clear(&ids);
    for (int ofs = 0; slides < max_slides && ofs <= probe_len - core_len && cur_delta > core_len / 3 &&
                      n_nodes < K4O_MAX_IDENT_NODES;
         slides++, ofs += cur_delta) {
      if (ofs + core_len + cur_delta > probe_len) cur_delta = probe_len - (ofs + core_len); /* :5956 */
      int64_t t = k4o_locate_first_exact(ix, probe + ofs, core_len, 0, n - 1, ctr);
      if (t == 0) continue;
      t -= 1;
      int iter = 0, first = 1;
      while (!max_iter || iter < max_iter) {
        if (n_nodes >= K4O_MAX_IDENT_NODES) break;
        if (!first) { /* :5978-6019 step to the next suffix while it still starts with the core */
          if (t + 1 >= n || k4o_sa_at(ix, t + 1) + core_len > n) break;
          if (cmp_probe_targ(probe + ofs, ix->seq + k4o_sa_at(ix, t + 1), core_len) != 0) break;
          t += 1;
        }
        first = 0;
        int64_t pos = k4o_sa_at(ix, t);
        if (pos < (int64_t)(uint32_t)ofs) continue; /* :6023 */
        int64_t left = pos - ofs;
        const k4o_entry* e = map_chunk_hit2entry(ix, (uint64_t)left);
        if (e == NULL || (uint64_t)left + (uint32_t)probe_len - 1 > e->end_ofs) continue; /* :6033 */
        uint32_t targ_id = (uint32_t)(1 + pos - (uint32_t)ofs); /* :6037, truncation is deliberate (Q7) */
        if (!idset_insert(&ids, targ_id)) continue;
        n_nodes++;
        iter++;
        /* :6190-6261 full-read Hamming extension with the two early-outs */
        if (ctr) ctr->n_cand++;
        const uint8_t* tb = ix->seq + left;
        int mm = 0, i;
        for (i = 0; i < probe_len; i++) {
          uint8_t tv = tb[i] & 0x0f, pv = probe[i] & 0x0f;
          if (tv == K4O_EOS) break;
          if (pv == tv) continue;
          if (++mm > max_tot_mm) break;
          if (mm >= nxt) break;
        }
        if (i != probe_len) continue;
        if (mm < low) { /* :6264-6284 new best */
          cur_hit = 0;
          inst = 1;
          nxt = low;
          low = mm;
          store_hit(&hits[0], e, left, cur_strand, probe_len, mm);
        } else if (mm == low) { /* :6286-6306 another instance of the best */
          inst += 1;
          if (cur_hit != -1 && inst <= max_hits) {
            cur_hit += 1;
            if (cur_hit < max_hits) store_hit(&hits[cur_hit], e, left, cur_strand, probe_len, mm);
          }
        } else if (mm < nxt) /* :6308-6311 */
          nxt = mm;
        if (inst > max_hits && low == 0) break; /* :6313 */
      }
      if (inst > max_hits && low == 0) { strand = 3; break; } /* :6316-6320 eALSnone */
    }
    if (cur_strand == '+' && strand == K4O_STRAND_BOTH) { /* :6323-6334 */
      k4o_revcomp(probe, probe_len);
      cur_strand = '-';
      strand = K4O_STRAND_CRICK;
    } else
      strand = 3;
  } while (!(inst > max_hits && low == 0) && strand != 3);
  idset_free(&ids);
  if (cur_strand == '-') k4o_revcomp(probe, probe_len); /* :6338-6342 restore */

  /* :6345-6368 */
  if (*p_low == low && *p_inst == inst) {
    if (*p_nxt > nxt) {
      *p_nxt = nxt;
      if (nxt - *p_low < mm_delta) return K4O_HR_MMDELTA;
      return K4O_HR_RMMDELTA;
    }
    return K4O_HR_NONE;
  }
  *p_low = low; *p_inst = inst; *p_nxt = nxt;
  if (inst >= 1 && (nxt - low) < mm_delta) return K4O_HR_MMDELTA;
  if (inst > max_hits) return K4O_HR_HITINSTS;
  return K4O_HR_HITS;
}

/* ---- LocateBestMatches, SfxArray.cpp:6836-7205: at most max_hits alignments with no more than max_tot_mm mismatches,
 * kept sorted by mismatches; returns 0 (none), 1..max_hits, or max_hits+1 when further matches were sloughed;
 * *p_inst = alignments in hits[].  hits needs room for max_hits + 1 records (the reference's own memmove spills one). */
static int64_t locate_last_exact(const k4o_index* ix, const uint8_t* probe, int probe_len, int64_t first_idx) {
  /* LocateLastExact (:8226-8340) as the reference calls it here: index+1 of the last suffix that starts with the probe;
   * first_idx is a suffix known to match */
  int64_t lo = first_idx, hi = (int64_t)ix->n - 1;
  while (lo < hi) { /* largest index whose suffix still equals the probe */
    int64_t mid = lo + (hi - lo + 1) / 2;
    if (cmp_probe_targ(probe, ix->seq + k4o_sa_at(ix, mid), probe_len) == 0) lo = mid; else hi = mid - 1;
  }
  return lo + 1;
}

int k4o_locate_best_matches(const k4o_index* ix, int max_tot_mm, int core_len, int core_delta, int max_slides, int strand,
                            uint8_t* probe, int probe_len, int max_hits, int* p_inst, k4o_hit* hits, int cur_max_iter,
                            k4o_counters* ctr) {
  if (ix->n == 0) return -1;
  const int64_t n = (int64_t)ix->n;
  int inst = 0, sloughed = 0;
  if (p_inst) *p_inst = 0;
  char cur_strand = '+';
  if (strand == K4O_STRAND_CRICK) { k4o_revcomp(probe, probe_len); cur_strand = '-'; }
  idset ids;
  idset_init(&ids);
  do {
    int cur_delta = core_delta, slides = 0;
    uint32_t n_nodes = 0;
    idset_clear(&ids);
    for (int ofs = 0; slides < max_slides && ofs <= probe_len - core_len && cur_delta > core_len / 3 &&
                      n_nodes < K4O_MAX_IDENT_NODES;
         slides++, ofs += cur_delta) {
      if (ofs + core_len + cur_delta > probe_len) cur_delta = probe_len - (ofs + core_len);
      int64_t t = k4o_locate_first_exact(ix, probe + ofs, core_len, 0, n - 1, ctr);
      if (t == 0) continue;
      t -= 1;
      int iter = 0, first = 1;
      uint32_t num_copies = 0;
      while (!cur_max_iter || iter < cur_max_iter) {
        if (n_nodes >= K4O_MAX_IDENT_NODES) break;
        if (!first) {
          if (t + 1 >= n || k4o_sa_at(ix, t + 1) + core_len > n) break;
          if (iter == 100 && !num_copies) { /* :6969-6976 too many copies of this core: give it up */
            int64_t last = locate_last_exact(ix, probe + ofs, core_len, t);
            num_copies = last > 0 ? (uint32_t)(1 + last - t) : 0;
            if (cur_max_iter && num_copies > (uint32_t)cur_max_iter) break;
          }
          if (cmp_probe_targ(probe + ofs, ix->seq + k4o_sa_at(ix, t + 1), core_len) != 0) break;
          t += 1;
        }
        first = 0;
        int64_t pos = k4o_sa_at(ix, t);
        if (pos < (int64_t)(uint32_t)ofs) continue;
        int64_t left = pos - ofs;
        if ((uint64_t)left + (uint32_t)probe_len > ix->n) continue; /* :7034 */
        uint32_t targ_id = (uint32_t)(1 + pos - (uint32_t)ofs);
        if (!idset_insert(&ids, targ_id)) continue;
        n_nodes++;
        iter++;
        if (ctr) ctr->n_cand++;
        const uint8_t* tb = ix->seq + left;
        int mm = 0, i;
        for (i = 0; i < probe_len; i++) { /* :7060-7125 */
          uint8_t tv = tb[i] & 0x0f, pv = probe[i] & 0x0f;
          if (tv == K4O_EOS) break;
          if (pv == tv) continue;
          if (++mm > max_tot_mm) break;
        }
        if (i != probe_len) continue;
        /* :7129-7176 keep the hits sorted by mismatches, at most max_hits of them */
        int cur = -1;
        if (inst) {
          if (inst == max_hits) sloughed = 1;
          int b;
          for (b = 0; b < inst; b++)
            if (hits[b].mismatches > mm) {
              cur = b;
              if (b + 1 < max_hits) memmove(&hits[b + 1], &hits[b], sizeof(k4o_hit) * (size_t)(inst - b));
              break;
            }
          if (b == inst && inst < max_hits) cur = inst;
        } else
          cur = 0;
        if (cur >= 0) {
          const k4o_entry* e = map_chunk_hit2entry(ix, (uint64_t)left);
          if (e == NULL) continue; /* (cannot happen: the window holds no separator) */
          store_hit(&hits[cur], e, left, cur_strand, probe_len, mm);
          if (inst < max_hits) inst += 1;
          else max_tot_mm = hits[inst - 1].mismatches; /* :7171-7175 only better ones from now on */
        }
      }
      if (inst == max_hits && max_tot_mm == 0 && !sloughed) { strand = 3; break; }
    }
    if (cur_strand == '+' && strand == K4O_STRAND_BOTH) {
      k4o_revcomp(probe, probe_len);
      cur_strand = '-';
      strand = K4O_STRAND_CRICK;
    } else
      strand = 3;
  } while (!(inst == max_hits && max_tot_mm == 0 && !sloughed) && strand != 3);
  idset_free(&ids);
  if (cur_strand == '-') k4o_revcomp(probe, probe_len);
  if (p_inst) *p_inst = inst;
  if (inst == 0) return 0;
  return sloughed ? inst + 1 : inst;
}

/* ---- AlignReads, SfxArray.cpp:7838-7933 (microInDelLen = MaxSpliceJunctLen = MinChimericLen = 0) -- */
int k4o_align_reads(const k4o_index* ix, int tot_mm, int core_len, int core_delta, int max_slides, int min_core_len,
                    int mm_delta, int strand, int* inst, int* low, int* nxt, uint8_t* probe, int probe_len,
                    int max_hits, k4o_hit* hits, k4o_counters* ctr) {
  (void)min_core_len; /* only used by the chimeric phase */
  int rslt = 0, allow = 0;
  if (tot_mm > 0) {
    for (allow = 0; allow <= tot_mm; allow++) {
      int cl = probe_len / (allow + mm_delta);
      if (cl <= core_len) break;
      rslt = k4o_locate_core_multiples(ix, allow, cl, cl, max_slides, mm_delta, strand, inst, low, nxt, probe,
                                       probe_len, max_hits, hits, ctr);
      if (rslt != 0) return rslt;
    }
  }
  if (allow <= tot_mm) {
    rslt = k4o_locate_core_multiples(ix, tot_mm, core_len, core_delta, max_slides, mm_delta, strand, inst, low, nxt,
                                     probe, probe_len, max_hits, hits, ctr);
    if (rslt != 0) return rslt;
  }
  return 0;
}

/* ---- CKAligner::LocateCoredApprox parameter derivation, KAligner.cpp:9367-9393 -------------------- */
int k4o_min_core_len(const k4o_index* ix, int pmode, int* max_num_slides) {
  uint64_t tot = k4o_tot_seqs_len(ix);
  int autolen = 1;
  while (tot >>= 2) autolen++;
  autolen -= 1;
  int mcl = autolen > 4 ? autolen : 4; /* cKAMinCoreLen, KAligner.h:39 */
  int slides;
  switch (pmode) {
    case 2: mcl -= 2; slides = 9; break; /* ePMUltraSens */
    case 1: mcl -= 1; slides = 8; break; /* ePMMoreSens */
    case 0: slides = 8; break;           /* ePMdefault */
    default: mcl += 2; slides = 6; break;
  }
  if (max_num_slides) *max_num_slides = slides;
  return mcl;
}

/* KAligner.cpp:9662-9672 */
void k4o_read_params(const k4o_kalign_params* kp, int min_core_len, int slides_per100, int read_len, int* max_tot_mm,
                     int* core_len, int* core_delta, int* max_slides) {
  int mm = kp->max_subs == 0 ? 0 : (int)(0.5 + (read_len * kp->max_subs) / 100.0);
  if (kp->max_subs != 0 && mm < 1) mm = 1;
  if (mm > 63) mm = 63; /* cMaxTotAllowedSubs */
  int cl = read_len / (kp->min_edit_dist == 1 ? mm + 1 : mm + 2);
  if (cl < min_core_len) cl = min_core_len;
  int sl = (slides_per100 * read_len + 99) / 100;
  if (sl < 1) sl = 1;
  int cd = read_len / sl - 1;
  if (cd < cl) cd = cl;
  *max_tot_mm = mm; *core_len = cl; *core_delta = cd; *max_slides = sl;
}

/* ---- CKAligner::AlignRead, KAligner.cpp:9583-10105 (base space, no priority regions, MLMode default) */
static int align_read_with(const k4o_index* ix, const k4o_kalign_params* kp, int min_core_len, int slides_per100,
                           const uint8_t* read, int read_len, uint8_t* scratch, k4o_read_result* out, k4o_hit* hits,
                           k4o_counters* ctr) {
  memset(out, 0, sizeof(*out));
  out->nar = K4O_NAR_NOHIT;
  /* :9618-9640 unpack, count Ns */
  int max_ns_seq = 0, ns = 0, i;
  if (kp->max_ns) {
    max_ns_seq = (read_len * kp->max_ns) / 100;
    if (max_ns_seq < kp->max_ns) max_ns_seq = kp->max_ns;
  }
  for (i = 0; i < read_len; i++) {
    uint8_t b = read[i] & 0x07;
    scratch[i] = b;
    if (b > K4O_N) break;
    if (b == K4O_N && ++ns > max_ns_seq) break;
  }
  if (i != read_len) {
    out->nar = K4O_NAR_NS;
    out->hit_rslt = K4O_HR_SEQERRS;
    return K4O_HR_SEQERRS;
  }
  int tot_mm, core_len, core_delta, slides;
  k4o_read_params(kp, min_core_len, slides_per100, read_len, &tot_mm, &core_len, &core_delta, &slides);
  int inst = 0, low = 0, nxt = 0;
  int max_ml = kp->max_ml < 1 ? 1 : kp->max_ml;
  memset(hits, 0, sizeof(k4o_hit) * (size_t)max_ml);
  int r;
  if (kp->pe_mode == 4) { /* -N: LocateBestMatches instead of AlignReads, :9776-9796 */
    k4o_hit* tmp = (k4o_hit*)calloc((size_t)max_ml + 1, sizeof(k4o_hit));
    r = k4o_locate_best_matches(ix, tot_mm, core_len, core_delta, slides, kp->strand, scratch, read_len, max_ml, &inst, tmp,
                                ix->max_iter, ctr);
    memcpy(hits, tmp, sizeof(k4o_hit) * (size_t)max_ml);
    for (int q = inst; q < max_ml; q++) memset(&hits[q], 0, sizeof(k4o_hit));
    free(tmp);
    r = r == 0 ? K4O_HR_NONE : (r >= 1 ? K4O_HR_HITS : r);
  } else
    r = k4o_align_reads(ix, tot_mm, core_len, core_delta, slides, min_core_len, kp->min_edit_dist, kp->strand, &inst,
                        &low, &nxt, scratch, read_len, max_ml, hits, ctr);
  if (inst > max_ml) inst = max_ml + 1; /* :9854 */
  if (kp->pe_mode >= 3 && r == K4O_HR_HITINSTS) { inst = max_ml; r = K4O_HR_HITS; } /* -X / -N clamp, :9856-9861 */
  out->hit_rslt = r;
  out->inst = inst; out->low_mm = low; out->nxt_mm = nxt;
  switch (r) {
    case K4O_HR_NONE: /* :9891-9905 */
      out->nar = K4O_NAR_NOHIT; out->low_mm = 0; out->inst = 0; out->nxt_mm = 0; /* tsReadHit fields as AlignRead leaves them */
      break;
    case K4O_HR_HITS: /* :9907-10025 */
      if (kp->pe_mode >= 2) { out->nar = K4O_NAR_ACCEPTED; out->num_hits = inst < max_ml ? inst : max_ml; } /* eMLall :9913-9931 */
      else if (!kp->pe_mode || inst == 1) { out->nar = K4O_NAR_ACCEPTED; out->num_hits = 1; }
      else { out->nar = K4O_NAR_MULTIALIGN; out->num_hits = inst; }
      break;
    case K4O_HR_MMDELTA: out->nar = K4O_NAR_MMDELTA; break;        /* :10027-10039 */
    case K4O_HR_HITINSTS: out->nar = K4O_NAR_MULTIALIGN; break;    /* :10041-10051,10068-10079 */
    default: break;                                                 /* eHRRMMDelta: NAR stays NL */
  }
  return r;
}

int k4o_align_read(const k4o_index* ix, const k4o_kalign_params* kp, const uint8_t* read, int read_len,
                   k4o_read_result* out, k4o_hit* hits, k4o_counters* ctr) {
  int spm = kp->max_num_slides;
  int mcl = kp->min_core_len;
  if (mcl <= 0 || spm <= 0) {
    int s2;
    int m2 = k4o_min_core_len(ix, kp->pmode, &s2);
    if (mcl <= 0) mcl = m2;
    if (spm <= 0) spm = s2;
  }
  uint8_t* scratch = (uint8_t*)malloc((size_t)read_len + 8);
  int r = align_read_with(ix, kp, mcl, spm, read, read_len, scratch, out, hits, ctr);
  free(scratch);
  return r;
}

/* ---- batches over pthreads (reads are independent; KAligner.cpp:10110-10263,10370-10438) ---------- */
typedef struct {
  const k4o_index* ix;
  const k4o_kalign_params* kp;
  int mcl, spm;
  int64_t n;
  const uint8_t* reads; const uint64_t* offs; const uint32_t* lens;
  k4o_read_result* out; k4o_hit* hits;
  /* raw mode */
  int raw, tot_mm, core_len, core_delta, slides, mm_delta, strand, max_hits;
  int32_t *rslt, *inst, *low, *nxt;
  int64_t next; pthread_mutex_t mtx;
  k4o_counters ctr;
} batch_job;

static void* batch_worker(void* arg) {
  batch_job* j = (batch_job*)arg;
  k4o_counters c = { 0, 0, 0 };
  uint8_t* scratch = (uint8_t*)malloc(1 << 16);
  const int64_t chunk = 256;
  for (;;) {
    pthread_mutex_lock(&j->mtx);
    int64_t b = j->next;
    j->next += chunk;
    pthread_mutex_unlock(&j->mtx);
    if (b >= j->n) break;
    int64_t e = b + chunk < j->n ? b + chunk : j->n;
    for (int64_t i = b; i < e; i++) {
      const uint8_t* rd = j->reads + j->offs[i];
      int len = (int)j->lens[i];
      if (j->raw) {
        int mh = j->max_hits;
        k4o_hit* h = j->hits + (size_t)i * mh;
        memset(h, 0, sizeof(k4o_hit) * (size_t)mh);
        memcpy(scratch, rd, (size_t)len);
        int in = 0, lo = 0, nx = 0;
        j->rslt[i] = k4o_align_reads(j->ix, j->tot_mm, j->core_len, j->core_delta, j->slides, 0, j->mm_delta,
                                     j->strand, &in, &lo, &nx, scratch, len, mh, h, &c);
        j->inst[i] = in; j->low[i] = lo; j->nxt[i] = nx;
      } else {
        int mh = j->kp->max_ml < 1 ? 1 : j->kp->max_ml;
        align_read_with(j->ix, j->kp, j->mcl, j->spm, rd, len, scratch, &j->out[i], j->hits + (size_t)i * mh, &c);
      }
    }
  }
  free(scratch);
  pthread_mutex_lock(&j->mtx);
  j->ctr.n_lookup += c.n_lookup; j->ctr.n_probe += c.n_probe; j->ctr.n_cand += c.n_cand;
  pthread_mutex_unlock(&j->mtx);
  return NULL;
}

static void run_batch(batch_job* j, int nthreads, k4o_counters* ctr) {
  pthread_mutex_init(&j->mtx, NULL);
  if (nthreads < 1) nthreads = 1;
  if (nthreads > 256) nthreads = 256;
  pthread_t th[256];
  for (int t = 0; t < nthreads; t++) pthread_create(&th[t], NULL, batch_worker, j);
  for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
  pthread_mutex_destroy(&j->mtx);
  if (ctr) *ctr = j->ctr;
}

int k4o_align_batch(const k4o_index* ix, const k4o_kalign_params* kp, int64_t n_reads, const uint8_t* reads,
                    const uint64_t* offs, const uint32_t* lens, k4o_read_result* out, k4o_hit* hits, int nthreads,
                    k4o_counters* ctr) {
  batch_job j;
  memset(&j, 0, sizeof(j));
  j.ix = ix; j.kp = kp; j.n = n_reads; j.reads = reads; j.offs = offs; j.lens = lens; j.out = out; j.hits = hits;
  j.mcl = kp->min_core_len; j.spm = kp->max_num_slides;
  if (j.mcl <= 0 || j.spm <= 0) {
    int s2, m2 = k4o_min_core_len(ix, kp->pmode, &s2);
    if (j.mcl <= 0) j.mcl = m2;
    if (j.spm <= 0) j.spm = s2;
  }
  run_batch(&j, nthreads, ctr);
  return 0;
}

int k4o_align_reads_batch(const k4o_index* ix, int tot_mm, int core_len, int core_delta, int max_slides,
                          int min_core_len, int mm_delta, int strand, int max_hits, int64_t n_reads,
                          const uint8_t* reads, const uint64_t* offs, const uint32_t* lens, int32_t* rslt,
                          int32_t* inst, int32_t* low, int32_t* nxt, k4o_hit* hits, int nthreads, k4o_counters* ctr) {
  (void)min_core_len;
  batch_job j;
  memset(&j, 0, sizeof(j));
  j.ix = ix; j.n = n_reads; j.reads = reads; j.offs = offs; j.lens = lens; j.hits = hits;
  j.raw = 1; j.tot_mm = tot_mm; j.core_len = core_len; j.core_delta = core_delta; j.slides = max_slides;
  j.mm_delta = mm_delta; j.strand = strand; j.max_hits = max_hits;
  j.rslt = rslt; j.inst = inst; j.low = low; j.nxt = nxt;
  run_batch(&j, nthreads, ctr);
  return 0;
}


/* ==== paired ends ============================================================================================= */
/* PEInsertSize, KAligner.cpp:2875-2918 (m_bPEcircularised false) */
int k4o_pe_insert_size(int pair_min_len, int pair_max_len, int pair_strand, uint8_t s1, uint32_t st1, uint32_t en1,
                       uint8_t s2, uint32_t st2, uint32_t en2) {
  if ((pair_strand && s1 != s2) || (!pair_strand && s1 == s2)) return -1;
  uint32_t mx = en1 > en2 ? en1 : en2, mn = st1 < st2 ? st1 : st2;
  int frag = (int)(1 + mx - mn);
  if (frag < 0) return -1;
  if (frag < pair_min_len) return -6;
  if (frag > pair_max_len) return -7;
  return frag;
}

/* AcceptProvPE, KAligner.cpp:2799-2861 (no chromosome filters; hits are untrimmed full-length: Adj* are identities) */
/* AdjStartLoci / AdjEndLoci (KAligner.cpp:1633-1650) on the flat hit: trims live in ext (TrimLeft | TrimRight << 12) */
static uint32_t hit_adj_start(const k4o_hit* h) { return h->match_loci + (h->strand == '+' ? (h->ext & 0xFFF) : ((h->ext >> 12) & 0xFFF)); }
static uint32_t hit_adj_end(const k4o_hit* h) {
  return h->match_loci + (h->match_len - (h->strand == '+' ? ((h->ext >> 12) & 0xFFF) : (h->ext & 0xFFF)) - 1);
}
static int accept_prov_pe(const k4o_pe_params* pe, int nh1, const k4o_hit* h1, int nh2, const k4o_hit* h2) { /* :2799-2861 */
  if (!(nh1 == 1 && nh2 == 1)) return 0;
  if (h1->chrom_id != h2->chrom_id) return -2;
  return k4o_pe_insert_size(pe->pair_min_len, pe->pair_max_len, pe->pair_strand, h1->strand, hit_adj_start(h1), hit_adj_end(h1),
                            h2->strand, hit_adj_start(h2), hit_adj_end(h2));
}

/* AdaptiveTrim with MinTrimLen == SeqLen, SfxArray.cpp:5561-5639: mismatch count with the end-flank rule; returns
 * SeqLen and *mms when accepted, 0 when not, <0 on parameter errors */
static int adaptive_trim_full(int seq_len, const uint8_t* probe, const uint8_t* targ, uint32_t max_mm, uint32_t min_flank,
                              uint32_t* mms) {
  *mms = 0;
  if (seq_len < 25 || seq_len > 2048 || seq_len < 15 || max_mm > (uint32_t)((15 * seq_len + 99) / 100) || min_flank > 10)
    return -100;
  if (min_flank == 0) min_flank = 1;
  uint32_t max_allowed = ((uint32_t)seq_len * max_mm + 99) / 100, n = 0;
  for (int o = 0; o < seq_len; o++) {
    if ((probe[o] & 0x0f) != (targ[o] & 0x0f)) {
      if (++n > max_allowed) break;
      if ((uint32_t)o < min_flank || (uint32_t)(seq_len - o) < min_flank) { n = max_allowed + 1; break; }
    }
  }
  if (n <= max_allowed) { *mms = n; return seq_len; }
  return 0;
}

/* AlignPairedRead, SfxArray.cpp:8571-8767, MinChimericLen == 0: the linear scan of :8731-8766.
 * Returns 1 with *out filled, 0 no match, -1 bad arguments.
 * Windows of 1000 loci or more: the reference takes its seed branch there (:8685-8726) with CoreLen forced to 0 (:8616-8620),
 * and IterateExactsRange (:3461-3553) with a zero-length probe never sees a mismatch, so its do/while walks past the end of
 * the suffix array: `ngskit4b kalign -U1 -d100 -D1500` on tests/golden/g1.sfx ends with SIGSEGV at the first rescue.  There
 * is nothing to restate; the linear-scan rule is applied to windows of any size (parity UNPINNED for that case: no reference
 * output can exist). */
int k4o_align_paired_read(const k4o_index* ix, int b3prime, int antisense, uint32_t chrom_id, uint32_t start_loci,
                          uint32_t end_loci, int min_insert, int max_insert, int max_allowed_mm, int read_len,
                          const uint8_t* read, k4o_hit* out) {
  memset(out, 0, sizeof(*out));
  if (chrom_id < 1 || chrom_id > ix->n_entries) return -1;
  const k4o_entry* e = &ix->entries[chrom_id - 1];
  uint32_t chrom_len = e->seq_len;
  if (chrom_len == 0) return -1;
  if (start_loci >= end_loci || end_loci >= chrom_len) return -1;
  if (min_insert > max_insert) return 0;
  if (min_insert < read_len) { max_insert += read_len - min_insert; min_insert = read_len; }
  uint32_t sp, ep;
  if (b3prime) {
    if ((uint32_t)(start_loci + min_insert) >= chrom_len) return 0;
    sp = start_loci + min_insert - read_len;
    uint32_t a = chrom_len - read_len, b = (uint32_t)(start_loci + max_insert - read_len);
    ep = a < b ? a : b;
  } else {
    if (end_loci < (uint32_t)min_insert) return 0;
    sp = end_loci <= (uint32_t)max_insert ? 0 : end_loci - max_insert;
    ep = end_loci - min_insert;
  }
  uint8_t* rs = (uint8_t*)malloc((size_t)read_len + 1);
  memcpy(rs, read, (size_t)read_len);
  if (antisense) k4o_revcomp(rs, read_len);
  const uint8_t* chrom = ix->seq + e->start_ofs;
  int min_put_len = read_len;
  uint32_t prev_best = (uint32_t)max_allowed_mm + 1;
  for (uint32_t loci = sp; loci <= ep; loci++) {
    uint32_t mms;
    int r = adaptive_trim_full(read_len, rs, chrom + loci, (uint32_t)max_allowed_mm, 3, &mms);
    if (r > min_put_len || (r == min_put_len && mms < prev_best)) {
      prev_best = mms;
      min_put_len = r;
      out->chrom_id = chrom_id;
      out->match_loci = loci;
      out->match_len = (uint16_t)read_len;
      out->strand = antisense ? '-' : '+';
      out->mismatches = (uint8_t)mms;
      if (mms == 0) break;
    }
    if (ep == 0xFFFFFFFFu) break;
  }
  free(rs);
  return prev_best <= (uint32_t)max_allowed_mm ? 1 : 0;
}

typedef struct {
  const k4o_index* ix; k4o_kalign_params kp; const k4o_pe_params* pe; int mcl, spm;
  int64_t n; const uint8_t *r1, *r2; const uint64_t *o1, *o2; const uint32_t *l1, *l2; k4o_pe_read* out;
  int64_t next; pthread_mutex_t mtx; int bad;
} pe_job;

static void classify_pe_se(const k4o_read_result* r, const k4o_hit* hits, k4o_pe_read* o) { /* AlignRead, PE branch */
  memset(o, 0, sizeof(*o));
  o->nar = r->nar; o->num_hits = r->num_hits; o->inst = r->inst; o->low_mm = r->low_mm;
  if (r->nar == K4O_NAR_ACCEPTED) o->hit = hits[0];
}

static int try_rescue(pe_job* j, const k4o_pe_read* anchor, int anchor_is_pe1, const uint8_t* mate, int mate_len,
                      k4o_hit* hit, int* frag) {
  const k4o_pe_params* pe = j->pe;
  int plus = anchor->hit.strand == '+';
  int b3, anti;
  if (anchor_is_pe1) { /* KAligner.cpp:3329-3335 */
    b3 = plus;
    anti = pe->pair_strand ? !plus : plus;
  } else {             /* :3438-3444 */
    b3 = plus; anti = plus;
    if (pe->pair_strand) { b3 = !b3; anti = !anti; }
  }
  uint32_t st = hit_adj_start(&anchor->hit), en = hit_adj_end(&anchor->hit); /* OrphStartLoci / OrphEndLoci, :3354-3355 */
  uint8_t* seq = (uint8_t*)malloc((size_t)mate_len + 1);
  for (int i = 0; i < mate_len; i++) seq[i] = mate[i] & 0x07;
  int r;
  if (j->kp.min_chimeric_len > 0) { /* :3357-3386 the window core the caller derives for this mate */
    int tot_mm = j->kp.max_subs == 0 ? 0 : (int)(0.5 + ((mate_len - 1) * j->kp.max_subs) / 100.0);
    if (j->kp.max_subs != 0 && tot_mm < 1) tot_mm = 1;
    if (tot_mm > 63) tot_mm = 63; /* cMaxTotAllowedSubs */
    int core_len = mate_len / (j->kp.min_edit_dist == 1 ? tot_mm + 1 : tot_mm + 2);
    if (core_len < j->mcl) core_len = j->mcl;
    int core_delta = mate_len / j->spm - 1;
    if (core_delta < core_len) core_delta = core_len;
    r = k4o_align_paired_read_x(j->ix, b3, anti, anchor->hit.chrom_id, st, en, pe->pair_min_len, pe->pair_max_len, j->kp.max_subs,
                                mate_len, j->kp.min_chimeric_len, core_len, core_delta, seq, hit);
  } else
    r = k4o_align_paired_read(j->ix, b3, anti, anchor->hit.chrom_id, st, en, pe->pair_min_len, pe->pair_max_len,
                              j->kp.max_subs, mate_len, seq, hit);
  free(seq);
  if (r == 1) {
    uint32_t hs = hit_adj_start(hit), he = hit_adj_end(hit); /* :3390, :3492 */
    *frag = anchor_is_pe1 ? k4o_pe_insert_size(pe->pair_min_len, pe->pair_max_len, pe->pair_strand, anchor->hit.strand, st, en, hit->strand, hs, he)
                          : k4o_pe_insert_size(pe->pair_min_len, pe->pair_max_len, pe->pair_strand, hit->strand, hs, he, anchor->hit.strand, st, en);
    if (*frag <= 0) r = 0;
  }
  return r == 1;
}

static void process_pair(pe_job* j, int64_t i, uint8_t* scratch) {
  const k4o_pe_params* pe = j->pe;
  k4o_read_result r1, r2;
  k4o_hit h1[10], h2[10];
  const uint8_t* rd1 = j->r1 + j->o1[i];
  const uint8_t* rd2 = j->r2 + j->o2[i];
  int len1 = (int)j->l1[i], len2 = (int)j->l2[i];
  int hr1, hr2;
  if (j->kp.min_chimeric_len > 0) { /* AlignRead with the chimeric pass of AlignReads behind the standard phases */
    hr1 = k4oi_align_read_ext(j->ix, &j->kp, j->mcl, j->spm, rd1, len1, scratch, &r1, h1);
    hr2 = k4oi_align_read_ext(j->ix, &j->kp, j->mcl, j->spm, rd2, len2, scratch, &r2, h2);
  } else {
    hr1 = align_read_with(j->ix, &j->kp, j->mcl, j->spm, rd1, len1, scratch, &r1, h1, NULL);
    hr2 = align_read_with(j->ix, &j->kp, j->mcl, j->spm, rd2, len2, scratch, &r2, h2, NULL);
  }
  k4o_pe_read* f = &j->out[2 * i];
  k4o_pe_read* r = &j->out[2 * i + 1];
  classify_pe_se(&r1, h1, f);
  classify_pe_se(&r2, h2, r);
  /* ProcCoredApprox: both multi-hit (fewer than cMaxMLPEmatches each): accept iff exactly one consistent combination */
  if (hr1 == K4O_HR_HITS && hr2 == K4O_HR_HITS && !(f->inst == 1 && r->inst == 1) && f->inst < 10 && r->inst < 10) {
    int multi = 0, accepted = 0;
    k4o_hit p1, p2;
    for (int a = 0; !(multi && !accepted) && a < f->inst; a++)
      for (int b = 0; b < r->inst; b++)
        if (accept_prov_pe(pe, 1, &h1[a], 1, &h2[b]) > 0) {
          if (!multi) { p1 = h1[a]; p2 = h2[b]; multi = 1; accepted = 1; }
          else { accepted = 0; break; }
        }
    if (accepted) {
      f->hit = p1; f->nar = K4O_NAR_ACCEPTED; f->num_hits = 1;
      r->hit = p2; r->nar = K4O_NAR_ACCEPTED; r->num_hits = 1;
    }
  }
  /* ProcessPairedEnds, KAligner.cpp:3207-3585 */
  int f_unal = f->nar == K4O_NAR_NS || f->nar == K4O_NAR_NOHIT || f->nar == K4O_NAR_UNALIGNED;
  int r_unal = r->nar == K4O_NAR_NS || r->nar == K4O_NAR_NOHIT || r->nar == K4O_NAR_UNALIGNED;
  if (!(f->nar == K4O_NAR_ACCEPTED || r->nar == K4O_NAR_ACCEPTED)) return;
  const int unique_only = pe->pe_mode == 2;
  if (unique_only && (f_unal || r_unal)) goto no_pe_strict;
  if (f->nar == K4O_NAR_ACCEPTED && r->nar == K4O_NAR_ACCEPTED) {
    int frag = accept_prov_pe(pe, f->num_hits, &f->hit, r->num_hits, &r->hit);
    if (frag > 0) { f->pe_aligned = r->pe_aligned = 1; return; }
    switch (frag) {
      case -1: f->nar = r->nar = K4O_NAR_PESTRAND; break;
      case -2: f->nar = r->nar = K4O_NAR_PECHROM; break;
      case -6: f->nar = r->nar = K4O_NAR_PEINSERTMIN; break;
      case -7: f->nar = r->nar = K4O_NAR_PEINSERTMAX; break;
      default: break;
    }
    if (unique_only) goto no_pe_strict;
  }
  if (pe->pe_mode == 1 || pe->pe_mode == 3) {
    k4o_hit hit;
    int frag;
    if (f->num_hits == 1 && !r_unal && try_rescue(j, f, 1, rd2, len2, &hit, &frag)) {
      r->hit = hit; r->num_hits = 1; r->low_mm = hit.mismatches; r->inst = 1; r->rescued = 1;
      f->pe_aligned = r->pe_aligned = 1;
      f->nar = r->nar = K4O_NAR_ACCEPTED;
      return;
    }
    if (r->num_hits == 1 && !f_unal && try_rescue(j, r, 0, rd1, len1, &hit, &frag)) {
      f->hit = hit; f->low_mm = hit.mismatches; f->num_hits = 1; f->inst = 1; f->rescued = 1;
      f->pe_aligned = r->pe_aligned = 1;
      f->nar = r->nar = K4O_NAR_ACCEPTED;
      return;
    }
  }
  if (!(pe->pe_mode == 3 || pe->pe_mode == 4)) {
    f->num_hits = 0; f->inst = 0; r->num_hits = 0; r->inst = 0;
    if (f->nar == K4O_NAR_ACCEPTED) f->nar = K4O_NAR_PENOHIT;
    if (r->nar == K4O_NAR_ACCEPTED) r->nar = K4O_NAR_PENOHIT;
    return;
  }
  /* allowed to accept as SE if uniquely aligned (:3545-3584; no chromosome filter => bChromFilt == true when NumHits == 1) */
  if (f->num_hits != 1) { f->num_hits = 0; f->inst = 0; if (f->nar == K4O_NAR_ACCEPTED) f->nar = K4O_NAR_PEUNALIGN; }
  else f->nar = K4O_NAR_ACCEPTED;
  if (r->num_hits != 1) { r->num_hits = 0; r->inst = 0; if (r->nar == K4O_NAR_ACCEPTED) r->nar = K4O_NAR_PEUNALIGN; }
  else r->nar = K4O_NAR_ACCEPTED;
  return;
no_pe_strict:
  f->num_hits = 0; r->num_hits = 0; f->inst = 0; r->inst = 0;
  if (f->nar == K4O_NAR_ACCEPTED) f->nar = K4O_NAR_PENOHIT;
  if (r->nar == K4O_NAR_ACCEPTED) r->nar = K4O_NAR_PENOHIT;
}

static void* pe_worker(void* arg) {
  pe_job* j = (pe_job*)arg;
  uint8_t* scratch = (uint8_t*)malloc(1 << 16);
  for (;;) {
    pthread_mutex_lock(&j->mtx);
    int64_t b = j->next;
    j->next += 64;
    pthread_mutex_unlock(&j->mtx);
    if (b >= j->n) break;
    int64_t e = b + 64 < j->n ? b + 64 : j->n;
    for (int64_t i = b; i < e; i++) process_pair(j, i, scratch);
  }
  free(scratch);
  return NULL;
}

int k4o_kalign_pe_batch(const k4o_index* ix, const k4o_kalign_params* kp, const k4o_pe_params* pe, int64_t n_pairs,
                        const uint8_t* reads1, const uint64_t* offs1, const uint32_t* lens1, const uint8_t* reads2,
                        const uint64_t* offs2, const uint32_t* lens2, k4o_pe_read* out, int nthreads) {
  pe_job j;
  memset(&j, 0, sizeof(j));
  j.ix = ix; j.kp = *kp; j.pe = pe; j.n = n_pairs;
  j.kp.pe_mode = 1;
  j.kp.max_ml = kp->max_ml > 10 ? kp->max_ml : 10; /* max(m_MaxMLmatches, cMaxMLPEmatches), KAligner.cpp:9604 */
  j.r1 = reads1; j.o1 = offs1; j.l1 = lens1; j.r2 = reads2; j.o2 = offs2; j.l2 = lens2; j.out = out;
  j.mcl = kp->min_core_len; j.spm = kp->max_num_slides;
  if (j.mcl <= 0 || j.spm <= 0) {
    int s2, m2 = k4o_min_core_len(ix, kp->pmode, &s2);
    if (j.mcl <= 0) j.mcl = m2;
    if (j.spm <= 0) j.spm = s2;
  }
  pthread_mutex_init(&j.mtx, NULL);
  if (nthreads < 1) nthreads = 1;
  if (nthreads > 256) nthreads = 256;
  pthread_t th[256];
  for (int t = 0; t < nthreads; t++) pthread_create(&th[t], NULL, pe_worker, &j);
  for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
  pthread_mutex_destroy(&j.mtx);
  return j.bad ? -3 : 0;
}

/* ---- AssignMultiMatches: KAligner.cpp:4913-5258 ------------------------------------------------------------------ */
typedef struct {
  uint32_t read_id; /* load order + 1 */
  uint32_t slot;
  k4o_hit h;
  uint16_t score;
  uint8_t mh, mha, hl; /* FlagMH, FlagMHA, FlagHL (4 clustunique, 5 clustany) */
} mm_ent;

#define MM_UNIQ 0x8000u /* cUniqueClustFlg, KAligner.h:96-101 */
enum { MM_OVERLAP = 10, MM_USCORE = 5, MM_MSCORE = 1, MM_SCALE = 10, MM_MINSCORE = 50 };

static int mm_cmp_loci(const void* a, const void* b) { /* SortMultiHits, KAligner.cpp:11019-11053 */
  const mm_ent *x = (const mm_ent*)a, *y = (const mm_ent*)b;
  if (x->h.chrom_id != y->h.chrom_id) return x->h.chrom_id < y->h.chrom_id ? -1 : 1;
  if (x->h.match_loci != y->h.match_loci) return x->h.match_loci < y->h.match_loci ? -1 : 1;
  if (x->h.match_len != y->h.match_len) return x->h.match_len < y->h.match_len ? -1 : 1;
  if (x->h.mismatches != y->h.mismatches) return x->h.mismatches < y->h.mismatches ? -1 : 1;
  if (x->h.strand != y->h.strand) return x->h.strand < y->h.strand ? -1 : 1;
  if (x->read_id != y->read_id) return x->read_id < y->read_id ? -1 : 1;
  return 0;
}

static int mm_cmp_read(const void* a, const void* b) { /* SortMultiHitReadIDs, KAligner.cpp:11058-11098 */
  const mm_ent *x = (const mm_ent*)a, *y = (const mm_ent*)b;
  if (x->read_id != y->read_id) return x->read_id < y->read_id ? -1 : 1;
  if (x->score != y->score) return x->score > y->score ? -1 : 1;
  if (x->h.chrom_id != y->h.chrom_id) return x->h.chrom_id < y->h.chrom_id ? -1 : 1;
  if (x->h.match_len != y->h.match_len) return x->h.match_len < y->h.match_len ? -1 : 1;
  if (x->h.mismatches != y->h.mismatches) return x->h.mismatches < y->h.mismatches ? -1 : 1;
  if (x->h.match_loci != y->h.match_loci) return x->h.match_loci < y->h.match_loci ? -1 : 1;
  if (x->h.strand != y->h.strand) return x->h.strand < y->h.strand ? -1 : 1;
  return 0;
}

static void mm_score_one(mm_ent* e, int64_t n, int64_t i, int ml_mode, uint32_t max_reads_len) { /* :4975-5081 */
  mm_ent* cur = &e[i];
  const uint32_t cs = cur->h.match_loci, clen = cur->h.match_len, cend = cs + clen - 1;
  uint32_t score;
  cur->score = 0;
  for (int64_t j = i - 1; j >= 0; j--) { /* upstream */
    const mm_ent* c = &e[j];
    if (c->h.chrom_id != cur->h.chrom_id) break;
    if (cs - c->h.match_loci >= max_reads_len) break;
    const uint32_t ce = c->h.match_loci + c->h.match_len - 1;
    if (ce < cs + MM_OVERLAP) continue;
    uint32_t ov = ce - cs;
    if (clen < ov) ov = clen;
    if ((ml_mode == 3 && c->mh) || ((cur->score & MM_UNIQ) && (cur->score & ~MM_UNIQ) >= 0x1fff)) continue;
    if (c->h.strand != cur->h.strand || c->read_id == cur->read_id) continue;
    if (!c->mh) {
      score = 1 + (ov * MM_USCORE) / MM_SCALE;
      if (cur->score & MM_UNIQ) score += cur->score & ~MM_UNIQ;
      if (score > 0x1fff) score = 0x1fff;
      cur->score = (uint16_t)(score | MM_UNIQ);
      if (score == 0x1fff) break;
    } else if (!(cur->score & MM_UNIQ)) {
      score = 1 + (ov * MM_MSCORE) / MM_SCALE;
      score += cur->score & ~MM_UNIQ;
      if (score > 0x1fff) score = 0x1fff;
      cur->score = (uint16_t)score;
    }
  }
  for (int64_t j = i + 1; j < n; j++) { /* downstream */
    const mm_ent* c = &e[j];
    if (c->h.chrom_id != cur->h.chrom_id) break;
    if (c->h.match_loci > cend - (uint32_t)MM_OVERLAP) break;
    uint32_t ov = cend - c->h.match_loci;
    if (c->h.match_len < ov) ov = c->h.match_len;
    if ((ml_mode == 3 && c->mh) || ((cur->score & MM_UNIQ) && (cur->score & ~MM_UNIQ) >= 0x3fff)) continue;
    if (c->h.strand != cur->h.strand || c->read_id == cur->read_id) continue;
    if (!c->mh) {
      score = 1 + (ov * MM_USCORE) / MM_SCALE;
      if (cur->score & MM_UNIQ) score += cur->score & ~MM_UNIQ;
      if (score > 0x3fff) score = 0x3fff;
      cur->score = (uint16_t)(score | MM_UNIQ);
      if (score == 0x3fff) break;
    } else if (!(cur->score & MM_UNIQ)) {
      score = 1 + (ov * MM_MSCORE) / MM_SCALE;
      score += cur->score & ~MM_UNIQ;
      if (score > 0x3fff) score = 0x3fff;
      cur->score = (uint16_t)score;
    }
  }
}

int64_t k4o_assign_multi_matches(int ml_mode, int max_reads_len, int64_t n_reads, int max_ml, k4o_read_result* rr,
                                 k4o_hit* hits, int nthreads) {
  if (ml_mode != 3 && ml_mode != 4) return -1;
  if (nthreads < 1) nthreads = 1;
  int64_t n = 0;
  for (int64_t r = 0; r < n_reads; r++)
    if (rr[r].hit_rslt == K4O_HR_HITS) n += rr[r].inst; /* AddMHitReads, KAligner.cpp:10002-10022 */
  mm_ent* e = (mm_ent*)calloc((size_t)n + 2, sizeof(mm_ent));
  int64_t k = 0;
  for (int64_t r = 0; r < n_reads; r++) {
    if (rr[r].hit_rslt != K4O_HR_HITS) continue;
    for (int q = 0; q < rr[r].inst; q++) {
      e[k].read_id = (uint32_t)(r + 1);
      e[k].slot = (uint32_t)q;
      e[k].h = hits[r * max_ml + q];
      e[k].mh = rr[r].inst > 1;
      k++;
    }
  }
  qsort(e, (size_t)n, sizeof(mm_ent), mm_cmp_loci);
  /* ProcAssignMultiMatches over the blocks GetClusterStartEnd hands out (:4913-4940) */
  int64_t from = 0;
  while (from < n) {
    const uint32_t left = (uint32_t)(n - from);
    uint32_t take = left;
    if (left >= 100) {
      take = (uint32_t)nthreads + left / (uint32_t)nthreads;
      if (take > 2000u) take = 2000u;
      if (take > left) take = left;
    }
    const mm_ent* prev = NULL;
    for (int64_t i = from; i < from + take; i++) {
      if (!e[i].mh) continue;
      if (prev && prev->h.match_loci == e[i].h.match_loci && prev->h.match_len == e[i].h.match_len &&
          prev->h.strand == e[i].h.strand && prev->h.chrom_id == e[i].h.chrom_id) {
        e[i].score = prev->score;
        continue;
      }
      mm_score_one(e, n, i, ml_mode, (uint32_t)max_reads_len);
      prev = &e[i];
    }
    from += take;
  }
  /* the best-scoring locus of each multi-aligned read (:5119-5163) */
  qsort(e, (size_t)n, sizeof(mm_ent), mm_cmp_read);
  uint32_t cur_id = 0;
  for (int64_t i = 0; i < n; i++) {
    if (!e[i].mh || cur_id == e[i].read_id) continue;
    cur_id = e[i].read_id;
    const uint32_t best = e[i].score & ~MM_UNIQ;
    if (best < MM_MINSCORE) continue;
    if ((e[i].score & MM_UNIQ) == (e[i + 1].score & MM_UNIQ) && best < 2u * (e[i + 1].score & ~MM_UNIQ)) continue;
    e[i].mha = 1;
    e[i].hl = (e[i].score & MM_UNIQ) ? 4 : 5;
  }
  /* orphans: a locus won by clustering with other multi-aligned reads needs a neighbour that is still in play (:5168-5251) */
  qsort(e, (size_t)n, sizeof(mm_ent), mm_cmp_loci);
  int64_t assigned = 0;
  for (int64_t i = 0; i < n; i++) {
    if (!e[i].mha) continue;
    int accept = 1;
    if (e[i].hl == 5) {
      accept = 0;
      for (int64_t j = i - 1; j >= 0; j--) {
        const uint32_t dist = e[i].h.match_loci - e[j].h.match_loci;
        if (dist > (uint32_t)(MM_OVERLAP + (int)e[j].h.match_len)) break;
        if (e[j].h.chrom_id != e[i].h.chrom_id) break;
        if (!e[j].mh || e[j].mha == 1) { accept = 1; break; }
      }
      if (!accept)
        for (int64_t j = i + 1; j < n; j++) {
          const uint32_t dist = e[j].h.match_loci - e[i].h.match_loci;
          if (dist > (uint32_t)(MM_OVERLAP + (int)e[i].h.match_len)) break;
          if (e[j].h.chrom_id != e[i].h.chrom_id) break;
          if (!e[j].mh || e[j].mha == 1) { accept = 1; break; }
        }
      if (!accept) e[i].mha = 0;
    }
    if (accept) {
      const int64_t r = (int64_t)e[i].read_id - 1;
      hits[r * max_ml] = e[i].h;
      rr[r].num_hits = 1;
      rr[r].nar = K4O_NAR_ACCEPTED;
      rr[r].inst = 1;
      assigned++;
    }
  }
  free(e);
  return assigned;
}
