/* include/k4_bam.hpp -- host side of BAM output: BGZF blocks and the .bai index around the uncompressed alignment records
 * that k4_format_bam_dev / k4_pipeline_format_bam produce on the device.
 *
 * What it stands in for (libkit4b): CSAMfile::Create / AddRefSeq / StartAlignments / AddAlignment (BAM branch) / UpdateSAIIndex /
 * Close (SAMfile.cpp:1477-1700, 1901-2120, 2379-2654) over bgzf.cpp.  Same file layout -- magic, header text, reference
 * dictionary, records, BGZF end-of-file block; `<name>.bai` beside it -- decoded records are identical to the reference's;
 * the compressed bytes are not (block boundaries and deflate settings are free, SAM specification 4.1).
 *
 * The records arrive as an in-order byte stream in pieces of any size (the pipeline's pinned buffers).  The stream is cut
 * into blocks of at most 0xff00 bytes, each deflated (raw, zlib) by one of `threads` workers, written in order; the index
 * is built from the record headers as they pass (bin from the record, end from its CIGAR), virtual offsets from the block
 * table.  Plain C++11 + zlib: no HIP here. */
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <zlib.h>
#include <algorithm>
#include <condition_variable>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace k4bam {

static const size_t kBlock = 0xff00;  // uncompressed bytes per BGZF block (bgzf.cpp BGZF_BLOCK_SIZE)

// one BGZF member holding src[0, n): 18-byte header with the BC subfield, raw deflate, crc32 + isize
inline bool bgzf_block(const uint8_t* src, size_t n, int level, std::vector<uint8_t>& out) {
  out.resize(18 + compressBound((uLong)n) + 8 + 64);
  z_stream zs;
  memset(&zs, 0, sizeof(zs));
  if (deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return false;
  zs.next_in = const_cast<Bytef*>(src);
  zs.avail_in = (uInt)n;
  zs.next_out = out.data() + 18;
  zs.avail_out = (uInt)(out.size() - 18 - 8);
  const int rc = deflate(&zs, Z_FINISH);
  const size_t clen = zs.total_out;
  deflateEnd(&zs);
  if (rc != Z_STREAM_END || 18 + clen + 8 > 65536) return false;
  static const uint8_t head[12] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0};
  memcpy(out.data(), head, 12);
  out[12] = 'B'; out[13] = 'C'; out[14] = 2; out[15] = 0;
  const uint32_t bsize = (uint32_t)(18 + clen + 8 - 1);
  out[16] = (uint8_t)bsize; out[17] = (uint8_t)(bsize >> 8);
  const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), src, (uInt)n);
  uint8_t* t = out.data() + 18 + clen;
  for (int k = 0; k < 4; k++) { t[k] = (uint8_t)(crc >> (8 * k)); t[4 + k] = (uint8_t)((uint32_t)n >> (8 * k)); }
  out.resize(18 + clen + 8);
  return true;
}

inline int reg2bin(int64_t beg, int64_t end) {  // SAM specification 5.3 (CSAMfile::BAIreg2bin, SAMfile.cpp:2124)
  --end;
  if (beg >> 14 == end >> 14) return ((1 << 15) - 1) / 7 + (int)(beg >> 14);
  if (beg >> 17 == end >> 17) return ((1 << 12) - 1) / 7 + (int)(beg >> 17);
  if (beg >> 20 == end >> 20) return ((1 << 9) - 1) / 7 + (int)(beg >> 20);
  if (beg >> 23 == end >> 23) return ((1 << 6) - 1) / 7 + (int)(beg >> 23);
  if (beg >> 26 == end >> 26) return ((1 << 3) - 1) / 7 + (int)(beg >> 26);
  return 0;
}

struct RefSeq { std::string name; uint32_t len; };

class Writer {
 public:
  Writer() {}
  ~Writer() { if (fp_) fclose(fp_); }
  const std::string& error() const { return err_; }

  // header_text: the SAM header ("@HD...@SQ...@PG...\n"), refs: the dictionary in header order.  level 0..9 (kalign: 6).
  bool open(const std::string& path, const std::string& header_text, const std::vector<RefSeq>& refs, int level, int threads) {
    path_ = path; level_ = level; threads_ = std::max(1, threads); refs_ = refs;
    fp_ = fopen(path.c_str(), "wb");
    if (!fp_) return fail("cannot create " + path);
    idx_.assign(refs.size(), RefIndex());
    // BAI bins address 2^29 bases per sequence (the reference switches to a CSI index above that, SAMfile.cpp:1697-1712: not
    // built here): such a dictionary gets the BAM without an index file
    index_ok_ = true;
    for (const RefSeq& r : refs)
      if (r.len >= (1u << 29)) index_ok_ = false;
    std::vector<uint8_t> h;
    const char magic[4] = {'B', 'A', 'M', 1};
    h.insert(h.end(), magic, magic + 4);
    put32(h, (uint32_t)header_text.size());
    h.insert(h.end(), header_text.begin(), header_text.end());
    put32(h, (uint32_t)refs.size());
    for (const RefSeq& r : refs) {
      put32(h, (uint32_t)r.name.size() + 1);
      h.insert(h.end(), r.name.begin(), r.name.end());
      h.push_back(0);
      put32(h, r.len);
    }
    header_bytes_ = h.size();
    return append(h.data(), h.size(), false);
  }

  // the next piece of the record stream (records may straddle pieces)
  bool write(const void* p, size_t n) { return append((const uint8_t*)p, n, true); }

  // flush, end-of-file block, index file
  bool close() {
    if (!fp_) return true;
    if (!flush(true)) return false;
    static const uint8_t eof[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (fwrite(eof, 1, 28, fp_) != 28) return fail("write failed: " + path_);
    if (fclose(fp_) != 0) { fp_ = nullptr; return fail("close failed: " + path_); }
    fp_ = nullptr;
    if (carry_.size() || need_) return fail("truncated record at the end of the BAM stream");
    return index_ok_ ? write_index() : true;
  }
  uint64_t n_records() const { return n_rec_; }
  bool indexed() const { return index_ok_; }  // false: a sequence of 512 Mbp or more (no .bai written)
  const std::string& path() const { return path_; }
  uint64_t compressed_bytes() const { return coff_; }

 private:
  struct Chunk { uint64_t beg, end; };
  struct RefIndex {
    std::map<uint32_t, std::vector<Chunk>> bins;
    std::vector<uint64_t> linear;  // 16 kb windows: virtual offset of the first alignment overlapping (0: none yet)
    uint64_t n_mapped = 0, off_beg = 0, off_end = 0;
  };
  struct Rec { int32_t ref; int32_t pos, end; uint32_t bin; uint64_t ubeg, uend; };  // uncompressed offsets of the record

  static void put32(std::vector<uint8_t>& v, uint32_t x) { for (int k = 0; k < 4; k++) v.push_back((uint8_t)(x >> (8 * k))); }
  static uint32_t get32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
  bool fail(const std::string& m) { err_ = m; return false; }

  bool append(const uint8_t* p, size_t n, bool records) {
    if (records) scan_records(p, n);
    const size_t batch = kBlock * (size_t)threads_ * 64;  // blocks deflated per flush: 64 per thread (4 MB each)
    while (n) {
      const size_t take = std::min(n, batch - pend_.size());
      pend_.insert(pend_.end(), p, p + take);
      p += take; n -= take;
      if (pend_.size() >= batch && !flush(false)) return false;
    }
    return true;
  }

  // record headers out of the stream: refID, pos, bin and the reference span of the CIGAR
  void scan_records(const uint8_t* p, size_t n) {
    size_t i = 0;
    while (i < n) {
      if (need_ == 0) {  // at a record boundary (possibly with part of the 4-byte length carried over)
        while (carry_.size() < 4 && i < n) carry_.push_back(p[i++]);
        if (carry_.size() < 4) break;
        need_ = get32(carry_.data());
        rec_ubeg_ = header_bytes_ + consumed_;  // the record starts at its block_size field
        consumed_ += 4;
        carry_.clear();
        body_.clear();
      }
      const size_t take = std::min<size_t>(need_ - body_.size(), n - i);
      if (body_.empty() && take == need_) {  // the whole record lies in this piece: indexed in place
        consumed_ += take;
        index_record(p + i, need_, rec_ubeg_, header_bytes_ + consumed_);
        i += take;
        need_ = 0;
        continue;
      }
      body_.insert(body_.end(), p + i, p + i + take);
      i += take;
      consumed_ += take;
      if (body_.size() == need_) {
        index_record(body_.data(), need_, rec_ubeg_, header_bytes_ + consumed_);
        need_ = 0;
      }
    }
  }

  void index_record(const uint8_t* b, size_t len, uint64_t ubeg, uint64_t uend) {
    n_rec_++;
    if (len < 32) return;
    Rec r;
    r.ref = (int32_t)get32(b);
    r.pos = (int32_t)get32(b + 4);
    const uint32_t bmn = get32(b + 8), fnc = get32(b + 12);
    const uint32_t l_name = bmn & 0xff, n_ops = fnc & 0xffff;
    int64_t span = 0;
    const uint8_t* c = b + 32 + l_name;
    for (uint32_t q = 0; q < n_ops && (size_t)(c - b) + 4 <= len; q++, c += 4) {
      const uint32_t op = get32(c);
      const uint32_t code = op & 0xf;
      if (code == 0 || code == 2 || code == 3 || code == 7 || code == 8) span += op >> 4;  // M D N = X consume the reference
    }
    r.end = r.pos + (int32_t)std::max<int64_t>(span, 1);
    // the bin of the INDEX comes from the span on the reference, as indexers compute it; the record's own bin field (kept as
    // the reference writes it) is reg2bin over the aligned bases only, which for a spliced read can be a lower-level bin
    r.bin = (uint32_t)reg2bin(r.pos, r.end);
    r.ubeg = ubeg; r.uend = uend;
    recs_.push_back(r);
  }

  // deflate what is pending (whole blocks; all of it when final), write, and turn the records that are now fully on disk
  // into index entries
  bool flush(bool final) {
    const size_t nb_full = pend_.size() / kBlock;
    const size_t nb = final ? (pend_.size() + kBlock - 1) / kBlock : nb_full;
    if (nb) {
      std::vector<std::vector<uint8_t>> out(nb);
      std::vector<char> ok(nb, 1);
      const size_t total = final ? pend_.size() : nb * kBlock;
      auto work = [&](size_t t) {
        for (size_t k = t; k < nb; k += (size_t)threads_) {
          const size_t o = k * kBlock, len = std::min(kBlock, total - o);
          ok[k] = bgzf_block(pend_.data() + o, len, level_, out[k]) ? 1 : 0;
        }
      };
      std::vector<std::thread> th;
      for (int t = 1; t < threads_ && (size_t)t < nb; t++) th.emplace_back(work, (size_t)t);
      work(0);
      for (std::thread& x : th) x.join();
      for (size_t k = 0; k < nb; k++) {
        if (!ok[k]) return fail("deflate failed");
        block_uoff_.push_back(uwritten_);
        block_coff_.push_back(coff_);
        if (fwrite(out[k].data(), 1, out[k].size(), fp_) != out[k].size()) return fail("write failed: " + path_);
        coff_ += out[k].size();
        uwritten_ += std::min(kBlock, total - k * kBlock);
      }
      pend_.erase(pend_.begin(), pend_.begin() + (ptrdiff_t)total);
    }
    // records that end inside written blocks
    size_t done = 0;
    for (; done < recs_.size() && recs_[done].uend <= uwritten_; done++) add_to_index(recs_[done]);
    recs_.erase(recs_.begin(), recs_.begin() + (ptrdiff_t)done);
    return true;
  }

  uint64_t voffset(uint64_t u) const {  // virtual file offset of uncompressed offset u (u <= uwritten_)
    // the block that holds u: the last one starting at or before it; an offset at the very end of a block is the start of the next
    // (the end of the last written block is addressed as that block's length: readers carry on into the next block)
    const size_t k = (size_t)(std::upper_bound(block_uoff_.begin(), block_uoff_.end(), u) - block_uoff_.begin()) - 1;
    return (block_coff_[k] << 16) | (u - block_uoff_[k]);
  }

  void add_to_index(const Rec& r) {
    if (!index_ok_) return;
    if (r.ref < 0) n_no_coor_++;  // (-M1: the reads that were not accepted follow the alignments)
    if (r.ref < 0 || (size_t)r.ref >= idx_.size() || r.pos < 0) return;
    RefIndex& x = idx_[(size_t)r.ref];
    const uint64_t vb = voffset(r.ubeg), ve = voffset(r.uend);
    std::vector<Chunk>& cs = x.bins[r.bin];
    if (!cs.empty() && cs.back().end >> 16 == vb >> 16) cs.back().end = ve;  // adjacent in the same block: one chunk
    else cs.push_back({vb, ve});
    const size_t w0 = (size_t)(r.pos >> 14), w1 = (size_t)((r.end - 1) >> 14);
    if (x.linear.size() <= w1) x.linear.resize(w1 + 1, 0);
    for (size_t w = w0; w <= w1; w++)
      if (x.linear[w] == 0) x.linear[w] = vb;
    if (x.n_mapped == 0) x.off_beg = vb;
    x.off_end = ve;
    x.n_mapped++;
  }

  bool write_index() {
    std::vector<uint8_t> o;
    const char magic[4] = {'B', 'A', 'I', 1};
    o.insert(o.end(), magic, magic + 4);
    put32(o, (uint32_t)idx_.size());
    for (RefIndex& x : idx_) {
      put32(o, (uint32_t)x.bins.size() + (x.n_mapped ? 1u : 0u));
      for (auto& kv : x.bins) {
        put32(o, kv.first);
        put32(o, (uint32_t)kv.second.size());
        for (const Chunk& c : kv.second) { put64(o, c.beg); put64(o, c.end); }
      }
      if (x.n_mapped) {  // the metadata pseudo-bin samtools writes (37450: offsets of the reference's records, mapped / unmapped counts)
        put32(o, 37450); put32(o, 2);
        put64(o, x.off_beg); put64(o, x.off_end); put64(o, x.n_mapped); put64(o, 0);
      }
      // linear index: a window without its own alignment inherits the next lower one's offset (samtools does the same fill)
      uint64_t last = 0;
      for (uint64_t& v : x.linear) { if (v == 0) v = last; else last = v; }
      put32(o, (uint32_t)x.linear.size());
      for (uint64_t v : x.linear) put64(o, v);
    }
    put64(o, n_no_coor_);  // records without coordinates (only with -M1)
    const std::string ip = path_ + ".bai";
    FILE* f = fopen(ip.c_str(), "wb");
    if (!f) return fail("cannot create " + ip);
    const bool w = fwrite(o.data(), 1, o.size(), f) == o.size();
    if (fclose(f) != 0 || !w) return fail("write failed: " + ip);
    return true;
  }
  static void put64(std::vector<uint8_t>& v, uint64_t x) { for (int k = 0; k < 8; k++) v.push_back((uint8_t)(x >> (8 * k))); }

  std::string path_, err_;
  FILE* fp_ = nullptr;
  int level_ = 6, threads_ = 1;
  bool index_ok_ = true;
  std::vector<RefSeq> refs_;
  std::vector<RefIndex> idx_;
  std::vector<uint8_t> pend_;                 // uncompressed bytes not yet in a block
  std::vector<uint64_t> block_uoff_, block_coff_;
  uint64_t uwritten_ = 0, coff_ = 0, consumed_ = 0, header_bytes_ = 0, rec_ubeg_ = 0, n_rec_ = 0, n_no_coor_ = 0;  // consumed_: record-stream bytes seen
  std::vector<uint8_t> carry_, body_;
  uint32_t need_ = 0;
  std::vector<Rec> recs_;
};

}  // namespace k4bam
