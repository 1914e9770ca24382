// kit4b_amd/csrc/k4_stages.h -- what the overlapped pipeline (k4_pipeline.hip) uses of the emit stage (k4_io.hip) beyond the C ABI.
#pragma once
#include <stdint.h>
#include <vector>
#include <hip/hip_runtime.h>
#include "k4_internal.h"
#include "k4_pool.h"

// The body is written in slices of consecutive lines so that its way down to the host can start behind the first slice instead
// of behind the last: end[k] = byte offset behind slice k, ev[k] = event behind the kernel that wrote it (on the format stream).
// The events belong to the caller (hipEventDestroy).
struct K4SamSlices {
  std::vector<uint64_t> end;
  std::vector<hipEvent_t> ev;
  void clear() {
    for (hipEvent_t e : ev) (void)hipEventDestroy(e);
    ev.clear(); end.clear();
  }
};

// k4_format_sam_ext_dev (bam 0) / k4_format_bam_dev (bam 1) / k4_format_sam_all_dev (bam 2) / k4_format_bam_all_dev (bam 3) (include/k4sfx.h) with two additions: `slices` != nullptr -- the call returns with the
// writing kernels still running (wait for the events, or synchronise the stream); `out_buf` != nullptr -- the body is placed in
// a block of the device's pool held by *out_buf (and *d_sam points into it) instead of a hipMalloc'd block of the caller's.
int k4i_format_records(k4_index* ix, int bam, int sq_all, int pe, int64_t n_units, const void* d_rr, const void* d_hits, int32_t max_ml,
                       const void* d_pe, const void* d_seg2, const void* d_reads, const void* d_offs, const void* d_lens,
                       const k4_sam_names* names, void** d_sam, uint64_t* sam_bytes, k4_sam_stats* stats, uint8_t* chrom_hit, void* stream,
                       K4SamSlices* slices, K4PoolBuf* out_buf);
