/* include/k4comm.h -- C ABI of libk4comm.so: the multi-GPU side of the kalign hot path on one node.
 *
 * One process per GPU; RCCL over xGMI for exactly what north_star names and nothing on the data path:
 *   k4_comm_open_index      rank 0 reads the .sfx ONCE; the 1-byte sequence and the suffix array go to every peer over
 *                           xGMI -- root scatters 1/N of the buffer to each peer over its own direct link, then every rank
 *                           sends its piece to every other rank (all 7 links of every GPU busy; a ring would be bound by one
 *                           link, MI355X_MICROARCH.md) -- and every rank derives its packed reference, exception tables and
 *                           k-mer table locally (k4_open_device).  Replaces N x (CSfxArray::Open + SetTargBlock),
 *                           libkit4b/SfxArray.h:528,957.
 *   k4_comm_allreduce_sum_u64   the final aligned-read count/merge: the per-rank NAR tallies of ReportAlignStats
 *                           (ngskit4b/KAligner.cpp:3600-3830) summed with ncclAllReduce.
 * Reads are independent units: each rank aligns its own contiguous slice, no collective in between.
 * The library links RCCL; libk4sfx.so itself stays free of it.  Plain pointers and sizes only.
 */
#ifndef K4COMM_H
#define K4COMM_H
#include <stddef.h>
#include <stdint.h>
#include "k4sfx.h"

#ifdef __cplusplus
extern "C" {
#endif

#define K4_COMM_ID_BYTES 128 /* ncclUniqueId */
typedef struct k4_comm k4_comm;

/* The exchange k4_comm_open_index runs for each of the two big arrays, as data: who sends which byte range to whom in which of
 * its two phases (0: the root deals piece r to rank r, every direct link of the root busy; 1: every rank passes its own piece to
 * each peer that lacks it).  A pure function of (n_ranks, bytes) -- the RCCL calls are issued from exactly this list, and a CPU
 * test checks it for every rank count: all bytes reach all ranks, nobody sends what it does not hold yet, no (src, dst) link
 * carries more than one piece per phase.  Returns the number of transfers (<= 2 * n_ranks * n_ranks); cap too small: the count
 * needed, nothing written beyond cap. */
typedef struct { int32_t phase, src, dst; uint64_t off, len; } k4_comm_xfer;
int k4_comm_bcast_schedule(int n_ranks, uint64_t bytes, k4_comm_xfer* out, int cap);

int k4_comm_unique_id(uint8_t id[K4_COMM_ID_BYTES]);  /* one rank calls it and hands the bytes to the others (pipe, file, shared memory) */
/* Returns at once: ncclCommInitRank (seconds) runs on a thread of the library and is joined by the first call that talks to the
 * peers (k4_comm_open_index's first broadcast, k4_comm_allreduce_sum_u64, k4_comm_close) -- a communicator that could not be
 * formed is reported there.  Rank 0 maps and uploads the index file meanwhile. */
int k4_comm_init(int device, int rank, int n_ranks, const uint8_t id[K4_COMM_ID_BYTES], k4_comm** out);
int k4_comm_rank(const k4_comm* c);
int k4_comm_size(const k4_comm* c);
/* sfx_path is read by rank 0 only (the others may pass NULL); kmer_k as for k4_open */
int k4_comm_open_index(k4_comm* c, const char* sfx_path, int kmer_k, k4_index** out);
int k4_comm_allreduce_sum_u64(k4_comm* c, uint64_t* vals, int n); /* host array in, summed over the ranks out */
int k4_comm_barrier(k4_comm* c);
const char* k4_comm_last_error(const k4_comm* c);
void k4_comm_close(k4_comm* c);

#ifdef __cplusplus
}
#endif
#endif
