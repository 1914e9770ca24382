// include/k4_sfxarray.hpp -- C++ facade that keeps the CSfxArray call surface CKAligner uses
// (libkit4b/SfxArray.h:524-1023; call sites: `grep m_pSfxArray-> ngskit4b/KAligner.cpp`) on top of the C ABI of
// libk4sfx.so (include/k4sfx.h).  Header-only, plain C++11, no HIP / torch types.
//
// Same names, argument meaning, ownership and error behaviour as the reference:
//   * results < 0 are teBSFrsltCodes, 0..4 tHRslt; text is queued and drained with NumErrMsgs()/GetErrMsg()
//   * the caller owns the probe buffer (left unchanged), pHits[MaxHits] and the ident-node scratch (accepted and ignored:
//     the GPU path keeps its own dedupe state)
//   * AlignReads() forwards one read as a batch of 1; throughput callers use AlignReadsBatch() / KAlignBatch()
#ifndef K4_SFXARRAY_HPP
#define K4_SFXARRAY_HPP
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <string>
#include <vector>
#include "k4sfx.h"

#ifndef K4_HAVE_KIT4B_TYPES  // (define it when libkit4b's own headers already provide these types)
typedef uint8_t etSeqBase;  // libkit4b/commdefs.h:87
enum { eBaseA = 0, eBaseC, eBaseG, eBaseT, eBaseN, eBaseUndef, eBaseInDel, eBaseEOS };

typedef enum etALStrand { eALSboth, eALSWatson, eALSCrick, eALSnone } eALStrand;                 // SfxArray.h:72-77
typedef enum etHRslt { eHRnone = 0, eHRhits, eHRMMDelta, eHRHitInsts, eHRRMMDelta, eHRSeqErrs, eHRFatalError } tHRslt;  // :79-87

#pragma pack(1)
typedef struct TAG_sSegLoci {  // libkit4b/SfxArray.h:239-249 (23 bytes)
  uint16_t ReadOfs;
  uint8_t Strand;
  uint32_t ChromID;
  uint64_t MatchLoci;
  uint16_t MatchLen;
  uint8_t Mismatches;
  uint16_t TrimLeft;
  uint16_t TrimRight;
  uint8_t TrimMismatches;
} tsSegLoci;
typedef struct TAG_tsHitLoci {  // libkit4b/SfxArray.h:251-260 (50 bytes)
  etSeqBase BisBase;
  uint8_t FlgChimeric : 1;
  uint8_t FlgInDel : 1;
  uint8_t FlgInsert : 1;
  uint8_t FlgSplice : 1;
  uint8_t FlgNonOrphan : 1;
  uint16_t Score;
  tsSegLoci Seg[2];
} tsHitLoci;
#pragma pack()
#pragma pack(4)
typedef struct TAG_sSfxHeaderV3 {  // libkit4b/SfxArray.h:194-207 (1224 bytes)
  uint8_t Magic[4];
  int32_t Version;
  uint32_t Attributes;
  uint64_t FileLen;
  uint64_t EntriesOfs;
  uint32_t EntriesSize;
  uint32_t NumSfxBlocks;
  uint64_t SfxBlockSize;
  uint64_t SfxBlockOfs;
  uint8_t szDatasetName[81];
  uint8_t szDescription[1024];
  uint8_t szTitle[64];
} tsSfxHeaderV3;
#pragma pack()
typedef struct TAG_sIdentNode {  // libkit4b/SfxArray.h:144-147 (caller scratch; unused here)
  uint32_t TargSeqID;
  struct TAG_sIdentNode* pNxt;
} tsIdentNode;
#endif  // K4_HAVE_KIT4B_TYPES
static_assert(sizeof(tsSegLoci) == 23 && sizeof(tsHitLoci) == 50 && sizeof(tsSfxHeaderV3) == 1224, "layouts must match libkit4b");

class CSfxArray {
  k4_index* m_pIdx;
  int m_Device;
  int m_MaxIter;
  std::deque<std::string> m_Errs;
  std::mutex m_Mtx;  // CKAligner calls AlignReads / AlignPairedRead from many threads on one object; an
                     // index handle runs one batch at a time: whoever talks to the device holds this
  // AlignReads calls that arrive while a batch is on the device are collected and go out together as the next batch: the
  // first caller to find no batch in flight becomes the leader, runs the pending requests (grouped by parameter set) and
  // wakes their owners; with T caller threads the device sees batches of up to T reads instead of T batches of one.
  struct Req {
    k4_align_params p;
    const etSeqBase* probe;
    uint32_t len;
    int32_t rslt, inst, low, nxt, rc;
    std::vector<k4_hit> hits;
    k4_seg2 seg2;  // the second segment of a microInDel / splice-junction hit (slot 0)
    bool done;
  };
  std::mutex m_QMtx;
  std::condition_variable m_QCv;
  std::vector<Req*> m_Pending;
  bool m_LeaderActive = false;

  void RunGroup(std::vector<Req*>& g) {  // same parameters: one k4_align_reads_batch
    const size_t n = g.size();
    std::vector<uint8_t> cat;
    std::vector<uint64_t> offs(n);
    std::vector<uint32_t> lens(n);
    for (size_t i = 0; i < n; i++) {
      offs[i] = cat.size();
      lens[i] = g[i]->len;
      cat.insert(cat.end(), g[i]->probe, g[i]->probe + g[i]->len);
    }
    cat.resize(cat.size() + 16);
    const int mh = g[0]->p.max_hits;
    std::vector<int32_t> r(4 * n);
    std::vector<k4_hit> h(n * (size_t)mh);
    std::vector<k4_seg2> s2(n);
    int rc;
    {
      std::lock_guard<std::mutex> dev(m_Mtx);
      rc = k4_align_reads_ext_batch(m_pIdx, &g[0]->p, (int64_t)n, cat.data(), offs.data(), lens.data(), r.data(), r.data() + n,
                                    r.data() + 2 * n, r.data() + 3 * n, h.data(), s2.data());
      if (rc != K4_OK) Fail(rc);
    }
    for (size_t i = 0; i < n; i++) {
      Req* q = g[i];
      q->seg2 = s2[i];
      q->rc = rc; q->rslt = r[i]; q->inst = r[n + i]; q->low = r[2 * n + i]; q->nxt = r[3 * n + i];
      q->hits.assign(h.begin() + (ptrdiff_t)(i * mh), h.begin() + (ptrdiff_t)((i + 1) * mh));
    }
  }
  void Submit(Req& rq) {
    std::unique_lock<std::mutex> lk(m_QMtx);
    m_Pending.push_back(&rq);
    while (!rq.done) {
      if (m_LeaderActive) {  // a batch is on the device: this request goes with the next one
        m_QCv.wait(lk);
        continue;
      }
      m_LeaderActive = true;  // lead one batch: everything pending right now, this request included
      std::vector<Req*> batch;
      batch.swap(m_Pending);
      lk.unlock();
      std::vector<char> taken(batch.size(), 0);
      for (size_t i = 0; i < batch.size(); i++) {
        if (taken[i]) continue;
        std::vector<Req*> g;
        for (size_t j = i; j < batch.size(); j++)
          if (!taken[j] && std::memcmp(&batch[j]->p, &batch[i]->p, sizeof(k4_align_params)) == 0) { g.push_back(batch[j]); taken[j] = 1; }
        RunGroup(g);
      }
      lk.lock();
      for (Req* q : batch) q->done = true;
      m_LeaderActive = false;
      m_QCv.notify_all();
    }
  }

  std::vector<uint16_t> m_IdentFlags;  // per entry, the flags half of tsSfxEntry.fBlockID (SfxArray.cpp:2048-2079)

  bool LoadIdentFlags() {
    if (!m_pIdx) return false;
    if (m_IdentFlags.empty()) {
      const int n = GetNumEntries();
      m_IdentFlags.resize((size_t)n, 0);
      for (int e = 1; e <= n; e++) {
        k4_entry ent;
        if (k4_get_entry(m_pIdx, (uint32_t)e, &ent) == K4_OK) m_IdentFlags[(size_t)e - 1] = (uint16_t)((ent.fblock_id >> 8) & 0xffff);
      }
    }
    return true;
  }

  int Fail(int rc) {
    const char* m = m_pIdx ? k4_last_error(m_pIdx) : k4_global_error();
    m_Errs.push_back(m ? m : "");
    return rc;
  }
  // the flat records -> tsHitLoci as LocateCoreMultiples (SfxArray.cpp:6264-6307, chimeric :6123-6146), LocateInDels
  // (:7800-7825) and LocateSpliceJuncts (:7501-7516) leave it
  static void Expand(const k4_hit& h, tsHitLoci* p, const k4_seg2* s2 = nullptr) {
    std::memset(p, 0, sizeof(*p));
    p->BisBase = eBaseN;
    p->FlgChimeric = (h.ext & K4_EXT_CHIMERIC) ? 1 : 0;
    p->FlgInDel = (h.ext & K4_EXT_INDEL) ? 1 : 0;
    p->FlgInsert = (h.ext & K4_EXT_INSERT) ? 1 : 0;
    p->FlgSplice = (h.ext & K4_EXT_SPLICE) ? 1 : 0;
    p->FlgNonOrphan = (h.ext & K4_EXT_NONORPHAN) ? 1 : 0;
    p->Seg[0].Strand = h.strand;
    p->Seg[0].ChromID = h.chrom_id;
    p->Seg[0].MatchLoci = h.match_loci;
    p->Seg[0].MatchLen = h.match_len;
    p->Seg[0].Mismatches = h.mismatches;
    p->Seg[0].TrimMismatches = h.mismatches;
    p->Seg[0].TrimLeft = (uint16_t)K4_HIT_TRIM_LEFT(h);
    p->Seg[0].TrimRight = (uint16_t)K4_HIT_TRIM_RIGHT(h);
    if (s2 && (h.ext & (K4_EXT_INDEL | K4_EXT_SPLICE))) {
      p->Score = s2->score;
      p->Seg[1].ReadOfs = s2->read_ofs;
      p->Seg[1].Strand = h.strand;
      p->Seg[1].ChromID = s2->chrom_id;
      p->Seg[1].MatchLoci = s2->match_loci;
      p->Seg[1].MatchLen = s2->match_len;
      p->Seg[1].Mismatches = s2->mismatches;
      p->Seg[1].TrimMismatches = s2->mismatches;
    }
  }

 public:
  explicit CSfxArray(int Device = 0) : m_pIdx(nullptr), m_Device(Device), m_MaxIter(50000) {}
  ~CSfxArray() { Reset(); }
  CSfxArray(const CSfxArray&) = delete;
  CSfxArray& operator=(const CSfxArray&) = delete;

  int Reset(bool bFlush = true) {  // SfxArray.h:527
    (void)bFlush;
    if (m_pIdx) k4_close(m_pIdx);
    m_pIdx = nullptr;
    m_IdentFlags.clear();
    return 0;
  }
  int Close(bool bFlush = true) { return Reset(bFlush); }  // SfxArray.h:548

  // Open an existing .sfx (creation goes through `ngskit4b index` or k4_build_sa_device + k4_write_sfx).  SfxArray.h:528
  int Open(char* pszSeqFile, bool bCreate = false, bool bBisulfite = false, bool bColorspace = false) {
    Reset();
    if (bCreate || bBisulfite || bColorspace) {
      m_Errs.push_back("CSfxArray::Open: create/bisulfite/colorspace are outside the accelerated path");
      return K4_ERR_UNSUPPORTED;
    }
    int rc = k4_open(pszSeqFile, m_Device, 0, &m_pIdx);
    if (rc != K4_OK) return Fail(rc);
    k4_set_max_iter(m_pIdx, m_MaxIter);
    return 0;
  }
  int SetTargBlock(int BlockID) { return (m_pIdx && BlockID == 1) ? 0 : K4_ERR_PARAMS; }  // SfxArray.h:957: index already resident
  int Next(int PrevBlockID = 0) { return PrevBlockID == 0 && m_pIdx ? 1 : 0; }
  int SetMaxIter(int MaxIter) {  // SfxArray.h:556
    int prev = m_MaxIter;
    m_MaxIter = MaxIter > 0 ? MaxIter : 0;
    if (m_pIdx) k4_set_max_iter(m_pIdx, m_MaxIter);
    return prev;
  }
  int GetMaxIter(void) { return m_MaxIter; }
  int InitialiseCoreKMers(int KMerLen) { (void)KMerLen; return m_pIdx ? 0 : K4_ERR_INTERNAL; }  // SfxArray.h:1017: the k-mer table replaces the memo
  bool IsSOLiD(void) { return false; }

  int GetNumEntries(void) {  // SfxArray.h:958
    k4_info_t i;
    return m_pIdx && k4_info(m_pIdx, &i) == K4_OK ? (int)i.n_entries : 0;
  }
  uint64_t GetTotSeqsLen(void) {  // SfxArray.h:970
    k4_info_t i;
    return m_pIdx && k4_info(m_pIdx, &i) == K4_OK ? i.tot_seqs_len : 0;
  }
  uint32_t GetSeqLen(uint32_t EntryID) {  // SfxArray.h:969
    k4_entry e;
    return m_pIdx && k4_get_entry(m_pIdx, EntryID, &e) == K4_OK ? e.seq_len : 0;
  }
  int GetIdentName(uint32_t EntryID, int MaxLen, char* pszSeqIdent) {  // SfxArray.h:967
    k4_entry e;
    if (!m_pIdx || !pszSeqIdent || MaxLen < 1 || k4_get_entry(m_pIdx, EntryID, &e) != K4_OK) return K4_ERR_ENTRY;
    std::strncpy(pszSeqIdent, e.name, (size_t)MaxLen);
    pszSeqIdent[MaxLen - 1] = '\0';
    return 0;
  }
  int GetIdent(char* pszSeqIdent) { return m_pIdx ? k4_get_ident(m_pIdx, pszSeqIdent) : K4_ERR_ENTRY; }  // SfxArray.h:968
  char* GetDatasetName(void) {
    static thread_local char name[81];
    k4_info_t i;
    name[0] = 0;
    if (m_pIdx && k4_info(m_pIdx, &i) == K4_OK) std::strncpy(name, i.dataset, 80);
    return name;
  }
  uint32_t GetSeq(int EntryID, uint32_t Loci, etSeqBase* pRetSeq, uint32_t Len) {  // SfxArray.h:996
    std::lock_guard<std::mutex> lock(m_Mtx);
    return m_pIdx ? (uint32_t)k4_get_seq(m_pIdx, (uint32_t)EntryID, Loci, pRetSeq, Len) : 0;
  }
  int GetBase(int EntryID, uint32_t Loci) {  // SfxArray.h:992
    etSeqBase b;
    return GetSeq(EntryID, Loci, &b, 1) == 1 ? (int)b : K4_ERR_PARAMS;
  }
  int NumErrMsgs(void) { return (int)m_Errs.size(); }  // CErrorCodes, ErrorCodes.h:99-113
  char* GetErrMsg(void) {
    static thread_local std::string cur;
    if (m_Errs.empty()) return (char*)"";
    cur = m_Errs.front();
    m_Errs.pop_front();
    return (char*)cur.c_str();
  }
  k4_index* Handle(void) { return m_pIdx; }

  // per-entry flags (SfxArray.h:959-965): kept on the host, they never reach the device
  void InitAllIdentFlags(uint16_t Flags = 0) {
    if (LoadIdentFlags()) std::fill(m_IdentFlags.begin(), m_IdentFlags.end(), Flags);
  }
  uint16_t GetIdentFlags(uint32_t EntryID) {
    if (!LoadIdentFlags() || EntryID < 1 || EntryID > m_IdentFlags.size()) return (uint16_t)K4_ERR_ENTRY;
    return m_IdentFlags[EntryID - 1];
  }
  uint16_t SetResetIdentFlags(uint32_t EntryID, uint16_t SetFlags = 0, uint16_t ResetFlags = 0) {  // returns the flags before the call
    if (SetFlags == 0 && ResetFlags == 0) return GetIdentFlags(EntryID);
    if (!LoadIdentFlags() || EntryID < 1 || EntryID > m_IdentFlags.size()) return (uint16_t)K4_ERR_ENTRY;
    const uint16_t prev = m_IdentFlags[EntryID - 1];
    m_IdentFlags[EntryID - 1] = (uint16_t)((prev | SetFlags) & ~ResetFlags);
    return prev;
  }
  int GetSfxHeader(tsSfxHeaderV3* pSfxHeader) {  // SfxArray.h:551
    if (!m_pIdx || !pSfxHeader) return K4_ERR_PARAMS;
    return k4_get_sfx_header(m_pIdx, pSfxHeader);
  }
  int GetColorspaceSeq(int EntryID, uint32_t Loci, etSeqBase* pRetSeq, uint32_t Len) {  // SOLiD only; IsSOLiD() is false here
    (void)EntryID; (void)Loci; (void)pRetSeq; (void)Len;
    m_Errs.push_back("CSfxArray::GetColorspaceSeq: colourspace indexes are outside the accelerated path");
    return K4_ERR_UNSUPPORTED;
  }

  // CSfxArray::AlignReads, SfxArray.h:614-634 -- identical parameter list.
  int AlignReads(uint32_t ExtdProcFlags, uint32_t ReadID, int MinChimericLen, int TotMM, int CoreLen, int CoreDelta,
                 int MaxNumCoreSlides, int MinCoreLen, int MMDelta, eALStrand Align2Strand, int microInDelLen,
                 int MaxSpliceJunctLen, int* pLowHitInstances, int* pLowMMCnt, int* pNxtLowMMCnt, etSeqBase* pProbeSeq,
                 int ProbeLen, int MaxHits, tsHitLoci* pHits, int NumAllocdIdentNodes, tsIdentNode* pAllocsIdentNodes) {
    (void)ExtdProcFlags; (void)ReadID; (void)NumAllocdIdentNodes; (void)pAllocsIdentNodes;
    if (!m_pIdx) return K4_ERR_INTERNAL;
    if (*pLowHitInstances > 0) {  // carried-in hits (never passed by CKAligner::AlignRead, KAligner.cpp:9609-9611)
      m_Errs.push_back("CSfxArray::AlignReads: carried-in LowHitInstances > 0 is not supported");
      return K4_ERR_UNSUPPORTED;
    }
    if (Align2Strand == eALSnone) return eHRnone;
    if (ProbeLen < 1 || MaxHits < 1) return K4_ERR_PARAMS;
    Req rq;
    // MinChimericLen outside 15..99 means "no trimming" to LocateCoreMultiples (:5878-5883) but still runs its extra pass
    rq.p = k4_align_params{TotMM, CoreLen, CoreDelta, MaxNumCoreSlides, MinCoreLen, MMDelta, (int32_t)Align2Strand, MaxHits,
                           MinChimericLen > 0 ? MinChimericLen : 0, microInDelLen > 0 ? microInDelLen : 0,
                           MaxSpliceJunctLen > 0 ? MaxSpliceJunctLen : 0};
    std::memset(&rq.seg2, 0, sizeof(rq.seg2));
    rq.probe = pProbeSeq; rq.len = (uint32_t)ProbeLen;
    rq.rslt = rq.inst = rq.low = rq.nxt = 0; rq.rc = K4_OK; rq.done = false;
    Submit(rq);  // alone: a batch of one; with other threads calling at the same time: one batch for all of them
    if (rq.rc != K4_OK) return rq.rc;
    *pLowHitInstances = rq.inst; *pLowMMCnt = rq.low; *pNxtLowMMCnt = rq.nxt;
    int nvalid = (rq.rslt >= eHRhits && rq.rslt <= eHRHitInsts) ? (rq.inst < MaxHits ? rq.inst : MaxHits) : 0;
    for (int i = 0; i < nvalid; i++) Expand(rq.hits[(size_t)i], &pHits[i], i == 0 ? &rq.seg2 : nullptr);
    return rq.rslt;
  }

  // CSfxArray::LocateBestMatches, SfxArray.h:793-806 -- identical parameter list (CKAligner's -N, KAligner.cpp:9779).
  // Returns 0 (no match), 1..MaxHits, or MaxHits+1 when further matches were sloughed.
  int LocateBestMatches(uint32_t ReadID, int MaxTotMM, int CoreLen, int CoreDelta, int MaxNumCoreSlides, eALStrand Align2Strand,
                        etSeqBase* pProbeSeq, int ProbeLen, int MaxHits, int* pHitInstances, tsHitLoci* pHits, int CurMaxIter,
                        int NumAllocdIdentNodes, tsIdentNode* pAllocsIdentNodes) {
    (void)ReadID; (void)NumAllocdIdentNodes; (void)pAllocsIdentNodes;
    if (!m_pIdx) return K4_ERR_INTERNAL;
    if (pHitInstances) *pHitInstances = 0;
    if (Align2Strand == eALSnone) return 0;
    std::lock_guard<std::mutex> lock(m_Mtx);
    const int prev_iter = CurMaxIter != m_MaxIter ? k4_set_max_iter(m_pIdx, CurMaxIter) : -1;
    k4_align_params p = {MaxTotMM, CoreLen, CoreDelta, MaxNumCoreSlides, 0, 1, (int32_t)Align2Strand, MaxHits};
    uint64_t off = 0;
    uint32_t len = (uint32_t)ProbeLen;
    int32_t rslt = 0, inst = 0;
    std::vector<k4_hit> hits((size_t)MaxHits);
    int rc = k4_best_matches_batch(m_pIdx, &p, 1, pProbeSeq, &off, &len, &rslt, &inst, hits.data());
    if (prev_iter >= 0) k4_set_max_iter(m_pIdx, prev_iter);
    if (rc != K4_OK) return Fail(rc);
    if (pHitInstances) *pHitInstances = inst;
    for (int i = 0; i < inst && i < MaxHits; i++) Expand(hits[(size_t)i], &pHits[i]);
    return rslt;
  }

  // CSfxArray::AlignPairedRead, SfxArray.h:880-896 -- identical parameter list (CKAligner's mate rescue,
  // KAligner.cpp:3372-3386,3476-3490).  -1 errors, 0 no match, 1 placed with mismatches only.
  int AlignPairedRead(bool b3primeExtend, bool bAntisense, uint32_t ChromID, uint32_t StartLoci, uint32_t EndLoci, int MinInsertSize,
                      int MaxInsertSize, int MaxAllowedMM, int MinHamming, int ReadLen, int MinChimericLen, int CoreLen,
                      int CoreDelta, int MaxNumCoreSlides, etSeqBase* pRead, tsHitLoci* pAlign) {
    (void)MinHamming; (void)MaxNumCoreSlides;
    if (!m_pIdx) return -1;
    k4_rescue_task t;
    std::memset(&t, 0, sizeof(t));
    t.chrom_id = ChromID; t.start_loci = StartLoci; t.end_loci = EndLoci; t.read_len = (uint32_t)ReadLen; t.read_off = 0;
    t.b3prime_extend = b3primeExtend ? 1 : 0; t.antisense = bAntisense ? 1 : 0;
    t.min_insert = MinInsertSize; t.max_insert = MaxInsertSize; t.max_allowed_mm = MaxAllowedMM;
    if (MinChimericLen >= 15 && MinChimericLen <= 99 && CoreLen > 0)  // chimeric mode: CoreLen / CoreDelta seed the wide windows
      t.chimeric = (uint32_t)MinChimericLen | ((uint32_t)(CoreLen > 4095 ? 4095 : CoreLen) << 8) |
                   ((uint32_t)(CoreDelta > 4095 ? 4095 : CoreDelta < 0 ? 0 : CoreDelta) << 20);
    int32_t rslt = 0;
    k4_hit h;
    std::lock_guard<std::mutex> lock(m_Mtx);
    int rc = k4_mate_rescue_batch(m_pIdx, 1, &t, pRead, (uint64_t)ReadLen, &rslt, &h);
    if (rc != K4_OK) return Fail(rc);
    if (rslt == 1 && pAlign) Expand(h, pAlign);
    return rslt;
  }

  // Batched form of the same call: n fresh reads, uniform parameters; pHits holds n * MaxHits records.
  int AlignReadsBatch(int TotMM, int CoreLen, int CoreDelta, int MaxNumCoreSlides, int MinCoreLen, int MMDelta,
                      eALStrand Align2Strand, int64_t NumReads, const etSeqBase* pReads, const uint64_t* pOffs,
                      const uint32_t* pLens, int MaxHits, int32_t* pRslts, int32_t* pLowHitInstances, int32_t* pLowMMCnt,
                      int32_t* pNxtLowMMCnt, k4_hit* pHits) {
    if (!m_pIdx) return K4_ERR_INTERNAL;
    k4_align_params p = {TotMM, CoreLen, CoreDelta, MaxNumCoreSlides, MinCoreLen, MMDelta, (int32_t)Align2Strand, MaxHits};
    int rc = k4_align_reads_batch(m_pIdx, &p, NumReads, pReads, pOffs, pLens, pRslts, pLowHitInstances, pLowMMCnt,
                                  pNxtLowMMCnt, pHits);
    return rc == K4_OK ? 0 : Fail(rc);
  }

  // CKAligner::AlignRead for a batch (ngskit4b/KAligner.cpp:9583-10105): per-read parameters + NAR classification.
  int KAlignBatch(const k4_kalign_params& Pars, int64_t NumReads, const etSeqBase* pReads, const uint64_t* pOffs,
                  const uint32_t* pLens, k4_read_result* pResults, k4_hit* pHits) {
    if (!m_pIdx) return K4_ERR_INTERNAL;
    int rc = k4_kalign_batch(m_pIdx, &Pars, NumReads, pReads, pOffs, pLens, pResults, pHits);
    return rc == K4_OK ? 0 : Fail(rc);
  }
};
#endif
