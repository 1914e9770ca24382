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
#include <cstdint>
#include <cstring>
#include <deque>
#include <string>
#include <vector>
#include "k4sfx.h"

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
typedef struct TAG_sIdentNode {  // libkit4b/SfxArray.h:144-147 (caller scratch; unused here)
  uint32_t TargSeqID;
  struct TAG_sIdentNode* pNxt;
} tsIdentNode;
static_assert(sizeof(tsSegLoci) == 23 && sizeof(tsHitLoci) == 50, "tsHitLoci layout must match libkit4b");

class CSfxArray {
  k4_index* m_pIdx;
  int m_Device;
  int m_MaxIter;
  std::deque<std::string> m_Errs;

  int Fail(int rc) {
    const char* m = m_pIdx ? k4_last_error(m_pIdx) : k4_global_error();
    m_Errs.push_back(m ? m : "");
    return rc;
  }
  static void Expand(const k4_hit& h, tsHitLoci* p) {  // what LocateCoreMultiples stores, SfxArray.cpp:6264-6307
    std::memset(p, 0, sizeof(*p));
    p->BisBase = eBaseN;
    p->Seg[0].Strand = h.strand;
    p->Seg[0].ChromID = h.chrom_id;
    p->Seg[0].MatchLoci = h.match_loci;
    p->Seg[0].MatchLen = h.match_len;
    p->Seg[0].Mismatches = h.mismatches;
    p->Seg[0].TrimMismatches = h.mismatches;
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

  // CSfxArray::AlignReads, SfxArray.h:614-634 -- identical parameter list.
  int AlignReads(uint32_t ExtdProcFlags, uint32_t ReadID, int MinChimericLen, int TotMM, int CoreLen, int CoreDelta,
                 int MaxNumCoreSlides, int MinCoreLen, int MMDelta, eALStrand Align2Strand, int microInDelLen,
                 int MaxSpliceJunctLen, int* pLowHitInstances, int* pLowMMCnt, int* pNxtLowMMCnt, etSeqBase* pProbeSeq,
                 int ProbeLen, int MaxHits, tsHitLoci* pHits, int NumAllocdIdentNodes, tsIdentNode* pAllocsIdentNodes) {
    (void)ExtdProcFlags; (void)ReadID; (void)NumAllocdIdentNodes; (void)pAllocsIdentNodes;
    if (!m_pIdx) return K4_ERR_INTERNAL;
    if (MinChimericLen > 0 || microInDelLen > 0 || MaxSpliceJunctLen > 0) {
      m_Errs.push_back("CSfxArray::AlignReads: chimeric / microInDel / splice phases are outside the accelerated path");
      return K4_ERR_UNSUPPORTED;
    }
    if (*pLowHitInstances > 0) {  // carried-in hits (never passed by CKAligner::AlignRead, KAligner.cpp:9609-9611)
      m_Errs.push_back("CSfxArray::AlignReads: carried-in LowHitInstances > 0 is not supported");
      return K4_ERR_UNSUPPORTED;
    }
    if (Align2Strand == eALSnone) return eHRnone;
    k4_align_params p = {TotMM, CoreLen, CoreDelta, MaxNumCoreSlides, MinCoreLen, MMDelta, (int32_t)Align2Strand, MaxHits};
    uint64_t off = 0;
    uint32_t len = (uint32_t)ProbeLen;
    int32_t rslt = 0, inst = 0, low = 0, nxt = 0;
    std::vector<k4_hit> hits((size_t)MaxHits);
    int rc = k4_align_reads_batch(m_pIdx, &p, 1, pProbeSeq, &off, &len, &rslt, &inst, &low, &nxt, hits.data());
    if (rc != K4_OK) return Fail(rc);
    *pLowHitInstances = inst; *pLowMMCnt = low; *pNxtLowMMCnt = nxt;
    int nvalid = (rslt >= eHRhits && rslt <= eHRHitInsts) ? (inst < MaxHits ? inst : MaxHits) : 0;
    for (int i = 0; i < nvalid; i++) Expand(hits[(size_t)i], &pHits[i]);
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
