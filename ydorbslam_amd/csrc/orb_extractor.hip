// Host side of the MI355X ORB extractor + its C ABI (include/ydorb/c_api.h).
// Mirrors YDORBSLAM::OrbExtractor (reference src/orbExtractor.hpp:31-74, orbExtractor.cpp:315-399):
// constructor tables on the host, everything per-frame on the GPU.  There is no CPU fallback: without
// a usable HIP device every entry point returns YDORB_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>
#include <cstdlib>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/ydorb/c_api.h"
#include "extract_kernels.hip.h"
#include "ydorb_host.h"

using namespace ydorb;

namespace {

inline int cvRoundF(float v) { return (int)lrintf(v); }  // round-half-even (default rounding mode)
inline int cvFloorF(float v) { int i = (int)v; return i - (i > v); }
inline int alignUp(int v, int a) { return (v + a - 1) / a * a; }
inline size_t alignUpZ(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct HostPlan {
  int w = 0, h = 0;
  PlanDev dev{};
  std::vector<CellDev> cells;
  size_t pyrFrameStride = 0, blurFrameStride = 0, qtFrameStride = 0;
  int maxCellDim = 0;   // largest FAST cell (without its 6-px halo): sizes the per-wave LDS of k_fast_cells
  int levelCellDim[kMaxLevels]{};   // ... per level: a launch group takes the tile pitch and LDS of ITS largest cell
  // quad-tree launch geometry per level: node table for 4*quota nodes, LDS-resident candidates up to candCap (HBM beyond)
  struct QtLevel { int nodeCap, candCap; size_t lds; int items; bool passOk; } qt[kMaxLevels]{};   // items: candidates per thread k_qt_fast needs for this level
  size_t qtLdsMax = 0;
  size_t qtPassLds = 0;   // the all-levels hand-over launch: largest node table + cell bases, candidates in HBM
  int qtPassNodes = 0;    // ... and how many nodes that table holds (a level that needs more cannot be handed over)
  std::vector<int> tabInt;      // xofs|yofs per level
  std::vector<short> tabShort;  // alpha|beta per level
  struct TabOff { int xofs, yofs, alpha, beta; } tab[kMaxLevels]{};
  // levels whose kernels also write their own reflect-101 pad (k_pyr_level0_f / k_pyr_resize_f): per-column and per-row tables
  struct FusedLevel { bool on = false; int shift = 0, padOff = 0, colOff = 0, rowOff = 0; } fused[kMaxLevels];
  std::vector<PyrPadEntry> padTab;
  std::vector<PyrColEntry> colTab;
  std::vector<int2> rowTab;
};

enum Stage { ST_PYR = 0, ST_FAST, ST_QT, ST_BLUR, ST_DESC, ST_COUNT };
const char* kStageNames[ST_COUNT] = {"pyramid", "fast_cells", "blur", "quadtree_after_blur", "orient_describe"};

}  // namespace

struct ydorb_extractor {
  YdExtractorConfig cfg{};
  std::vector<float> sf, isf, sf2, isf2;
  std::vector<int> perLevel;
  int maxX[16]{};
  int sumQuota = 0;
  hipStream_t stream = nullptr;
  hipStream_t qtStream[kMaxLevels]{};   // side streams of the per-level quad-tree launches
  hipEvent_t evFork = nullptr, evJoin[kMaxLevels]{};
  hipEvent_t evFast[kMaxLevels]{};      // end of the FAST launch that covers level l (levels are launched in groups)
  int descKpw = 2;                      // keypoints per wave of k_orient_describe_n
  bool qtInline = false;                // YDORB_QT_STREAMS=0: the quad-tree launches go on the caller's stream, behind the blur
  int fastGroups = 1;                   // FAST launches per call: 1 = all levels together
  HostPlan plan;
  bool planValid = false;
  int batchCap = 0;
  // device buffers (sized for plan x batchCap)
  uint8_t *d_img = nullptr, *d_pyr = nullptr, *d_blur = nullptr;
  uint32_t *d_cellCount = nullptr, *d_cellCand = nullptr, *d_qtCand = nullptr, *d_qtKeys = nullptr, *d_lvlKp = nullptr;
  uint16_t* d_qtNode = nullptr;
  uint8_t* d_nodeScratch = nullptr;  // HBM node tables of the levels whose quota does not fit the LDS (usually none)
  uint8_t* d_needPass = nullptr;   // [frame][level]: 1 = k_qt_fast left the unit to the pass kernel
  int* d_passList = nullptr;       // [1 + frames * levels]: count, then the units k_qt_fast handed over in this call
  int *d_lvlCount = nullptr, *d_status = nullptr, *d_nOut = nullptr;
  uint8_t* h_pyr = nullptr;           // pinned staging of one frame's pyramid block (ydorb_extractor_read_pyramid)
  size_t h_pyrBytes = 0;
  hipStream_t lastStream = nullptr;   // stream of the last enqueue (a caller's stream on the device-resident path)
  int *d_lvlMaxN = nullptr, *h_lvlMaxN = nullptr;   // largest candidate count seen per level (device max) + [kMaxLevels] the last hand-over list length; copied back after every call
  float* d_lvlAngle = nullptr;
  CellDev* d_cells = nullptr;
  int* d_tabInt = nullptr;
  short* d_tabShort = nullptr;
  PyrPadEntry* d_padTab = nullptr;
  PyrColEntry* d_colTab = nullptr;
  int2* d_rowTab = nullptr;
  bool noFusedPads = false;   // YDORB_PYR_FUSED=0 at create: level kernels + one border launch (the tests compare both)
  YdKeyPointDev* d_kps = nullptr;
  uint8_t* d_desc = nullptr;
  size_t imgBytes = 0;
  // pinned host staging for the host-pointer entry points
  uint8_t* h_img = nullptr;
  YdKeyPoint* h_kps = nullptr;
  uint8_t* h_desc = nullptr;
  int* h_nOut = nullptr;
  int* h_status = nullptr;
  int lastFrames = 0;
  // profiling
  bool profiling = false;
  bool profPending = false;       // the stage events of the last launch have not been read yet
  bool forcePassQuadtree = false;  // YDORB_QT_PASS=1 at create: run every unit through the pass kernel (tests compare both)
  hipEvent_t ev[ST_COUNT + 1]{};
  double stageMs[ST_COUNT]{};
  int stageCalls = 0;
};

namespace {

#define HIPCHK(expr)                                                                          \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess) {                                                                   \
      ydorb::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return YDORB_ERR_HIP;                                                                   \
    }                                                                                         \
  } while (0)

void freeBuffers(ydorb_extractor* e) {
  auto F = [](auto*& p) { if (p) { (void)hipFree(p); p = nullptr; } };
  F(e->d_img); F(e->d_pyr); F(e->d_blur); F(e->d_cellCount); F(e->d_cellCand); F(e->d_qtCand); F(e->d_qtKeys);
  F(e->d_lvlKp); F(e->d_qtNode); F(e->d_needPass); F(e->d_passList); F(e->d_nodeScratch); F(e->d_lvlCount); F(e->d_lvlMaxN); F(e->d_status); F(e->d_nOut); F(e->d_lvlAngle); F(e->d_cells);
  F(e->d_tabInt); F(e->d_tabShort); F(e->d_padTab); F(e->d_colTab); F(e->d_rowTab); F(e->d_kps); F(e->d_desc);
  auto H = [](auto*& p) { if (p) { (void)hipHostFree(p); p = nullptr; } };
  H(e->h_img); H(e->h_kps); H(e->h_desc); H(e->h_nOut); H(e->h_status); H(e->h_lvlMaxN); H(e->h_pyr);
  e->h_pyrBytes = 0;
  e->planValid = false;
}

// Level sizes, cell grid, scratch layout, resize tables for a w x h input.

// smallest instantiated items-per-thread of k_qt_fast whose 256*items registers hold `want` candidates (32 = the largest)
static int qfItemsFor(long want) { return want <= 8L * kQfThreads ? 8 : want <= 16L * kQfThreads ? 16 : 32; }

// Tables of a level whose kernel writes its own pad (extract_kernels.hip.h, k_pyr_level0_f / k_pyr_resize_f).  Leaves fused[l].on
// false when the scheme does not apply: a side below 20 px (a pad byte would need two reflections), horizontal scale factor > 2
// (the two taps of 4 outputs no longer fit one 8-byte load), or no lane shift keeps both end groups inside one wave.
void buildFusedLevel(HostPlan& P, int l) {
  const PlanDev& D = P.dev;
  const LevelDev& L = D.lv[l];
  HostPlan::FusedLevel& F = P.fused[l];
  F.on = false;
  if (L.w < 20 || L.h < 20) return;
  const int nd = (L.w + 2 * kPad + 3) >> 2;
  // mirrored source byte of padded-row byte p: -1 = own (interior), -2 = slack beyond w + 38 (zero)
  auto srcOf = [&](int p) { return p < kPad ? 2 * kPad - p : p < kPad + L.w ? -1 : p < L.w + 2 * kPad ? 2 * L.w + 2 * kPad - 2 - p : -2; };
  int shift = -1;
  for (int sft = 0; sft < 64 && shift < 0; sft++) {
    bool ok = true;
    for (int c = 0; c < nd && ok; c++)
      for (int j = 0; j < 4; j++) {
        const int sp = srcOf(4 * c + j);
        if (sp >= 0 && ((sp >> 2) + sft) >> 6 != (c + sft) >> 6) ok = false;
      }
    if (ok) shift = sft;
  }
  if (shift < 0) return;
  std::vector<PyrColEntry> col;
  if (l > 0) {
    const LevelDev& Lp = D.lv[l - 1];
    const int* xofs = P.tabInt.data() + P.tab[l].xofs;
    const short* alpha = P.tabShort.data() + P.tab[l].alpha;
    col.resize(nd);
    for (int c = 0; c < nd; c++) {
      const int dx0 = std::min(std::max(4 * c - kPad, 0), L.w - 1);
      for (int b = 0; b < 4; b++) {
        const int dx = std::min(std::max(4 * c - kPad + b, 0), L.w - 1);
        const int off = xofs[dx] - xofs[dx0];
        if (off < 0 || off > 6 || xofs[dx0] + 7 >= Lp.w + kPad) return;   // taps beyond one 8-byte load (or the load would leave the padded source row)
        col[c].sel[b] = (uint32_t)off | 0x0c000c00u | ((uint32_t)(off + 1) << 16);
        col[c].ab[b] = (uint32_t)(uint16_t)alpha[2 * dx] | ((uint32_t)(uint16_t)alpha[2 * dx + 1] << 16);
      }
    }
  }
  F.shift = shift;
  F.padOff = (int)P.padTab.size();
  for (int c = 0; c < nd; c++) {
    int dA = -1, dB = -1;
    for (int j = 0; j < 4; j++) {
      const int sp = srcOf(4 * c + j);
      if (sp < 0) continue;
      const int d = sp >> 2;
      if (dA < 0 || d == dA) dA = d;
      else dB = d;
    }
    PyrPadEntry pe{};
    pe.laneA4 = 4u * (unsigned)(((dA < 0 ? c : dA) + shift) & 63);
    pe.laneB4 = 4u * (unsigned)(((dB < 0 ? c : dB) + shift) & 63);
    pe.selM = 0; pe.selF = 0;
    for (int j = 0; j < 4; j++) {
      const int sp = srcOf(4 * c + j);
      uint32_t m = 0x0c, f;
      if (sp == -1) f = (uint32_t)j;
      else if (sp == -2) f = 0x0c;
      else { m = (uint32_t)(sp & 3) + ((sp >> 2) == dA ? 4u : 0u); f = 4u + (uint32_t)j; }
      pe.selM |= m << (8 * j);
      pe.selF |= f << (8 * j);
    }
    P.padTab.push_back(pe);
  }
  if (l > 0) {
    F.colOff = (int)P.colTab.size();
    P.colTab.insert(P.colTab.end(), col.begin(), col.end());
    F.rowOff = (int)P.rowTab.size();
    const int* yofs = P.tabInt.data() + P.tab[l].yofs;
    const short* beta = P.tabShort.data() + P.tab[l].beta;
    const int rows = alignUp(L.h, 4 * kPyrRowsF);
    for (int y = 0; y < rows; y++) {
      const int dy = std::min(y, L.h - 1);
      P.rowTab.push_back(make_int2(yofs[dy], (int)((uint32_t)(uint16_t)beta[2 * dy] | ((uint32_t)(uint16_t)beta[2 * dy + 1] << 16))));
    }
  }
  F.on = true;
}

int buildPlan(ydorb_extractor* e, int w, int h, HostPlan& P) {
  P = HostPlan();
  P.w = w; P.h = h;
  PlanDev& D = P.dev;
  D.nLevels = e->cfg.n_levels;
  for (int v = 0; v < 16; v++) D.maxX[v] = e->maxX[v];
  size_t padOff = 0, blurOff = 0;
  int kpOff = 0, candOff = 0, maxCellDim = 0;
  for (int l = 0; l < D.nLevels; l++) {
    LevelDev& L = D.lv[l];
    L.w = cvRoundF((float)w * e->isf[l]);   // orbExtractor.cpp:608-609
    L.h = cvRoundF((float)h * e->isf[l]);
    if (L.w < 1 || L.h < 1 || L.w > 4000 || L.h > 4000) {
      set_error("level %d size %dx%d out of the supported range (1..4000)", l, L.w, L.h);
      return YDORB_ERR_UNSUPPORTED;
    }
    L.pitch = alignUp(L.w + 2 * kPad, 64);
    L.padOff = (int)padOff;
    padOff += alignUpZ((size_t)(L.h + 2 * kPad) * L.pitch, 256);
    L.blurPitch = alignUp(L.w, 64);
    L.blurOff = (int)blurOff;
    blurOff += alignUpZ((size_t)L.h * L.blurPitch, 256);
    L.quota = e->perLevel[l];
    L.kpOff = kpOff;
    kpOff += L.quota;
    L.scale = e->sf[l];
    L.size = (float)(int)(31 * e->sf[l]);   // orbExtractor.cpp:595
    // cell grid, orbExtractor.cpp:548-580
    L.cellBegin = (int)P.cells.size();
    const int minB = kBorder, maxBX = L.w - kBorder, maxBY = L.h - kBorder;
    const float width = (float)(maxBX - minB), height = (float)(maxBY - minB);
    const int colsNum = (int)(width / 30.0f), rowsNum = (int)(height / 30.0f);
    if (colsNum > 0 && rowsNum > 0) {
      const int cellW = (int)ceilf(width / colsNum), cellH = (int)ceilf(height / rowsNum);
      maxCellDim = std::max(maxCellDim, std::max(cellW, cellH));
      P.levelCellDim[l] = std::max(cellW, cellH);
      for (int i = 0; i < rowsNum; i++) {
        const float iniY = (float)(minB + i * cellH);
        float maxY = iniY + cellH + 6;
        if (iniY >= maxBY - 3) continue;
        if (maxY > maxBY) maxY = (float)maxBY;
        for (int j = 0; j < colsNum; j++) {
          const float iniX = (float)(minB + j * cellW);
          float maxX = iniX + cellW + 6;
          if (iniX >= maxBX - 6) continue;
          if (maxX > maxBX) maxX = (float)maxBX;
          CellDev c{};
          c.level = (short)l;
          c.x0 = (short)iniX; c.y0 = (short)iniY; c.x1 = (short)maxX; c.y1 = (short)maxY;
          c.srcOff = L.padOff + (kPad + c.y0) * L.pitch + kPad + c.x0;
          P.cells.push_back(c);
        }
      }
    }
    L.nCells = (int)P.cells.size() - L.cellBegin;
    // resize tables (cv::resize INTER_LINEAR 8U: float coordinate, 11-bit coefficients), level >= 1
    if (l > 0) {
      const LevelDev& Lp = D.lv[l - 1];
      const double scale_x = 1. / ((double)L.w / Lp.w), scale_y = 1. / ((double)L.h / Lp.h);
      P.tab[l].xofs = (int)P.tabInt.size();
      P.tab[l].alpha = (int)P.tabShort.size();
      for (int dx = 0; dx < L.w; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cvFloorF(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= Lp.w - 1) { fx = 0; sx = Lp.w - 1; }
        P.tabInt.push_back(sx);
        P.tabShort.push_back((short)cvRoundF((1.f - fx) * 2048));
        P.tabShort.push_back((short)cvRoundF(fx * 2048));
      }
      P.tab[l].yofs = (int)P.tabInt.size();
      P.tab[l].beta = (int)P.tabShort.size();
      for (int dy = 0; dy < L.h; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cvFloorF(fy);
        fy -= sy;
        P.tabInt.push_back(sy);
        P.tabShort.push_back((short)cvRoundF((1.f - fy) * 2048));
        P.tabShort.push_back((short)cvRoundF(fy * 2048));
      }
    }
  }
  if (maxCellDim + 6 > kTileMax) {
    set_error("cell size %d exceeds the FAST tile (%d)", maxCellDim, kTileMax - 6);
    return YDORB_ERR_UNSUPPORTED;
  }
  D.nCellsTotal = (int)P.cells.size();
  P.maxCellDim = maxCellDim;
  // NMS survivors are pairwise non-adjacent: at most ceil(w/2)*ceil(h/2) per cell band
  D.cellCap = alignUp(std::max(((maxCellDim + 1) / 2) * ((maxCellDim + 1) / 2), 64), 64);
  D.sumQuota = kpOff;
  for (int l = 0; l < D.nLevels; l++) {
    D.lv[l].candOff = candOff;
    candOff += D.lv[l].nCells * D.cellCap;
  }
  for (int l = 0; l < D.nLevels && !e->noFusedPads; l++) buildFusedLevel(P, l);
  D.borderBegin[0] = 0;
  for (int l = 0; l < kMaxLevels; l++)   // fused levels have no border threads
    D.borderBegin[l + 1] = D.borderBegin[l] + (l < D.nLevels && !P.fused[l].on ? alignUp(2 * kPad * (D.lv[l].pitch / 4) + 12 * D.lv[l].h, 256) : 0);
  D.blurTileBegin[0] = 0;
  for (int l = 0; l < kMaxLevels; l++)
    D.blurTileBegin[l + 1] = D.blurTileBegin[l] + (l < D.nLevels ? ((D.lv[l].w + kBlurTW - 1) / kBlurTW) * ((D.lv[l].h + kBlurTH - 1) / kBlurTH) : 0);
  P.pyrFrameStride = padOff;
  P.blurFrameStride = blurOff;
  P.qtFrameStride = (size_t)candOff;
  // One quad-tree launch per level on its own stream, each with exactly the LDS that level needs: with one uniform
  // launch the level-0 footprint (~100 KB) limited every (frame, level) unit to one or two per CU although the small
  // levels need < 30 KB.  candCap: expected candidate density grows with the level (measured 1/65 .. 1/25 of the band
  // area on noise-like frames); a level with more candidates than candCap runs from HBM scratch instead (same result).
  for (int l = 0; l < D.nLevels; l++) {
    const LevelDev& L = D.lv[l];
    HostPlan::QtLevel& Q = P.qt[l];
    Q.nodeCap = 4 * std::max(L.quota, 1);
    const long band = (long)std::max(L.w - 2 * kBorder, 1) * std::max(L.h - 2 * kBorder, 1);
    Q.candCap = (int)std::min<long>(4608, std::max<long>(512, band / 22)) / 64 * 64;
    Q.lds = (size_t)Q.nodeCap * 48 + (size_t)Q.candCap * 12 + (size_t)(L.nCells + 1) * 4;
    if (Q.lds > 150 * 1024) {  // very large n_features: keep the node table, drop the LDS candidates
      Q.candCap = 0;
      Q.lds = (size_t)Q.nodeCap * 48 + (size_t)(L.nCells + 1) * 4;
    }
    // very large quotas (few levels x many features): the pass kernel's node table (48 B x 4 x quota) no longer fits the LDS; such a
    // level gets its node tables (and candidates) in HBM scratch — slow, but the flat kernel takes every unit with <= 8192 candidates
    // and a tree no deeper than 7 levels, so this is the rare hand-over path of an unusual configuration
    Q.passOk = Q.lds <= 150 * 1024;
    if (Q.passOk) {
      P.qtLdsMax = std::max(P.qtLdsMax, Q.lds);
      P.qtPassLds = std::max(P.qtPassLds, (size_t)Q.nodeCap * 48 + (size_t)(L.nCells + 1) * 4);
      P.qtPassNodes = std::max(P.qtPassNodes, Q.nodeCap);
      D.lv[l].nodeTabOff = -1;
    } else {
      if (Q.nodeCap > 65535) { set_error("n_features=%d: level %d would need %d quad-tree nodes (max 65535)", e->cfg.n_features, l, Q.nodeCap); return YDORB_ERR_UNSUPPORTED; }
      D.lv[l].nodeTabOff = (int)D.nodeTabFrameStride;
      D.nodeTabFrameStride += (unsigned)alignUpZ((size_t)Q.nodeCap * 48, 256);
      Q.candCap = 0;
      Q.lds = (size_t)(L.nCells + 1) * 4 + 64;
      P.qtLdsMax = std::max(P.qtLdsMax, Q.lds);
      P.qtPassLds = std::max(P.qtPassLds, Q.lds);
    }
    // k_qt_fast keeps 256*items candidates in registers.  First guess from the densities above (L0 1/45 .. L7 1/20 of the band);
    // enqueue() re-sizes it from the candidate counts the device actually saw.  A unit with more candidates goes to the pass kernel.
    Q.items = qfItemsFor((long)((double)band / (45.0 - 3.5 * l)) + 64);
  }
  P.qtPassLds = std::max<size_t>(P.qtPassLds, 1024);
  return YDORB_OK;
}

int ensurePlan(ydorb_extractor* e, int w, int h, int nFrames) {
  if (e->planValid && e->plan.w == w && e->plan.h == h && nFrames <= e->batchCap) return YDORB_OK;
  HIPCHK(hipStreamSynchronize(e->stream));
  HostPlan P;
  int rc = buildPlan(e, w, h, P);
  if (rc) return rc;
  freeBuffers(e);
  const int B = std::max(nFrames, std::max(1, e->cfg.max_batch));
  e->plan = P;
  e->batchCap = B;
  e->imgBytes = (size_t)w * h;
  const PlanDev& D = P.dev;
  HIPCHK(hipMalloc(&e->d_img, e->imgBytes * B));
  HIPCHK(hipMalloc(&e->d_pyr, P.pyrFrameStride * B));
  HIPCHK(hipMemsetAsync(e->d_pyr, 0, P.pyrFrameStride * B, e->stream));   // pitch slack beyond w + 38 is never written: keep it zero
  HIPCHK(hipMalloc(&e->d_blur, P.blurFrameStride * B));
  HIPCHK(hipMalloc(&e->d_cellCount, sizeof(uint32_t) * D.nCellsTotal * B));
  HIPCHK(hipMalloc(&e->d_cellCand, sizeof(uint32_t) * (size_t)D.nCellsTotal * D.cellCap * B));
  HIPCHK(hipMalloc(&e->d_qtCand, sizeof(uint32_t) * 2 * P.qtFrameStride * B));
  HIPCHK(hipMalloc(&e->d_qtNode, sizeof(uint16_t) * 2 * P.qtFrameStride * B));
  if (D.nodeTabFrameStride) HIPCHK(hipMalloc(&e->d_nodeScratch, (size_t)D.nodeTabFrameStride * B));
  HIPCHK(hipMalloc(&e->d_needPass, (size_t)kMaxLevels * B));
  HIPCHK(hipMemsetAsync(e->d_needPass, 0, (size_t)kMaxLevels * B, e->stream));
  HIPCHK(hipMalloc(&e->d_passList, sizeof(int) * ((size_t)kMaxLevels * B + 1)));
  HIPCHK(hipMemsetAsync(e->d_passList, 0, sizeof(int), e->stream));

  HIPCHK(hipMalloc(&e->d_lvlKp, sizeof(uint32_t) * (size_t)D.sumQuota * B));
  HIPCHK(hipMalloc(&e->d_lvlAngle, sizeof(float) * (size_t)D.sumQuota * B));
  HIPCHK(hipMalloc(&e->d_lvlCount, sizeof(int) * kMaxLevels * B));
  HIPCHK(hipMalloc(&e->d_lvlMaxN, sizeof(int) * (kMaxLevels + 1)));
  HIPCHK(hipMemsetAsync(e->d_lvlMaxN, 0, sizeof(int) * (kMaxLevels + 1), e->stream));
  HIPCHK(hipHostMalloc(&e->h_lvlMaxN, sizeof(int) * (kMaxLevels + 1)));
  memset(e->h_lvlMaxN, 0, sizeof(int) * (kMaxLevels + 1));
  HIPCHK(hipMalloc(&e->d_status, sizeof(int)));
  HIPCHK(hipMalloc(&e->d_nOut, sizeof(int) * B));
  HIPCHK(hipMalloc(&e->d_cells, sizeof(CellDev) * std::max<size_t>(P.cells.size(), 1)));
  HIPCHK(hipMalloc(&e->d_tabInt, sizeof(int) * std::max<size_t>(P.tabInt.size(), 1)));
  HIPCHK(hipMalloc(&e->d_tabShort, sizeof(short) * std::max<size_t>(P.tabShort.size(), 1)));
  HIPCHK(hipMalloc(&e->d_kps, sizeof(YdKeyPointDev) * (size_t)D.sumQuota * B));
  HIPCHK(hipMalloc(&e->d_desc, (size_t)32 * D.sumQuota * B));
  HIPCHK(hipHostMalloc(&e->h_img, e->imgBytes * B));
  HIPCHK(hipHostMalloc(&e->h_kps, sizeof(YdKeyPoint) * (size_t)D.sumQuota * B));
  HIPCHK(hipHostMalloc(&e->h_desc, (size_t)32 * D.sumQuota * B));
  HIPCHK(hipHostMalloc(&e->h_nOut, sizeof(int) * B));
  HIPCHK(hipHostMalloc(&e->h_status, sizeof(int)));
  if (!P.cells.empty()) HIPCHK(hipMemcpyAsync(e->d_cells, P.cells.data(), sizeof(CellDev) * P.cells.size(), hipMemcpyHostToDevice, e->stream));
  if (!P.tabInt.empty()) HIPCHK(hipMemcpyAsync(e->d_tabInt, P.tabInt.data(), sizeof(int) * P.tabInt.size(), hipMemcpyHostToDevice, e->stream));
  if (!P.tabShort.empty()) HIPCHK(hipMemcpyAsync(e->d_tabShort, P.tabShort.data(), sizeof(short) * P.tabShort.size(), hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipMalloc(&e->d_padTab, sizeof(PyrPadEntry) * std::max<size_t>(P.padTab.size(), 1)));
  HIPCHK(hipMalloc(&e->d_colTab, sizeof(PyrColEntry) * std::max<size_t>(P.colTab.size(), 1)));
  HIPCHK(hipMalloc(&e->d_rowTab, sizeof(int2) * std::max<size_t>(P.rowTab.size(), 1)));
  if (!P.padTab.empty()) HIPCHK(hipMemcpyAsync(e->d_padTab, P.padTab.data(), sizeof(PyrPadEntry) * P.padTab.size(), hipMemcpyHostToDevice, e->stream));
  if (!P.colTab.empty()) HIPCHK(hipMemcpyAsync(e->d_colTab, P.colTab.data(), sizeof(PyrColEntry) * P.colTab.size(), hipMemcpyHostToDevice, e->stream));
  if (!P.rowTab.empty()) HIPCHK(hipMemcpyAsync(e->d_rowTab, P.rowTab.data(), sizeof(int2) * P.rowTab.size(), hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipMemsetAsync(e->d_status, 0, sizeof(int), e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  if (std::max(P.qtLdsMax, P.qtPassLds) > 48 * 1024)
  {
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_quadtree), hipFuncAttributeMaxDynamicSharedMemorySize, (int)std::max(P.qtLdsMax, P.qtPassLds)));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_quadtree_list), hipFuncAttributeMaxDynamicSharedMemorySize, (int)std::max(P.qtLdsMax, P.qtPassLds)));
  }
  {
    int qmax = 1, tabLen = 2;
    for (int l = 0; l < D.nLevels; l++) { qmax = std::max(qmax, D.lv[l].quota); tabLen = std::max(tabLen, std::max(D.lv[l].w - 2 * kBorder, 0) + std::max(D.lv[l].h - 2 * kBorder, 0) + 2); }
    const size_t ldsFast = std::min<size_t>(qt_fast_lds_bytes(qmax, tabLen), 150 * 1024);
    if (ldsFast > 48 * 1024) {
      HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_qt_fast<8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsFast));
      HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_qt_fast<16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsFast));
      HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_qt_fast<32>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsFast));
    }
  }
  e->planValid = true;
  return YDORB_OK;
}

void collectProfile(ydorb_extractor* e);

// Adapt k_qt_fast's candidates-per-thread to the scene: the device keeps the largest candidate count per level, the copy-back at the
// end of every call brings it here (a call behind, which is fine - a unit that does not fit goes to the pass kernel, same keypoints).
static void retuneQuadtree(ydorb_extractor* e) {
  HostPlan& P = e->plan;
  for (int l = 0; l < P.dev.nLevels; l++) {
    const int seen = e->h_lvlMaxN[l];
    if (seen > 0) P.qt[l].items = qfItemsFor((long)seen + seen / 8 + 32);
  }
}

// Enqueue the whole front-end for nFrames device-resident images on `s`.
int enqueue(ydorb_extractor* e, const uint8_t* d_img, int stride, size_t frameStride, int nFrames, YdKeyPointDev* d_kps,
            uint8_t* d_desc, int cap, int* d_nOut, hipStream_t s) {
  retuneQuadtree(e);
  const HostPlan& P = e->plan;
  const PlanDev& D = P.dev;
  // stage events go on whatever stream the work is launched on (so a pipelined caller gets the durations as they really are,
  // other streams' kernels included); the previous launch's events are read first if they have completed — never waited for
  const bool prof = e->profiling;
  if (prof) {
    if (e->profPending && hipEventQuery(e->ev[ST_COUNT]) == hipSuccess) collectProfile(e);
    e->profPending = false;   // a launch whose events were not ready in time is dropped from the average
    HIPCHK(hipEventRecord(e->ev[0], s));
  }
  {
    const LevelDev& L0 = D.lv[0];
    auto fusedGrid = [&](const LevelDev& L, int shift) {
      return dim3((((L.w + 2 * kPad + 3) >> 2) + shift + 63) / 64, (L.h + 4 * kPyrRowsF - 1) / (4 * kPyrRowsF), nFrames);
    };
    if (P.fused[0].on) {
      hipLaunchKernelGGL(k_pyr_level0_f, fusedGrid(L0, P.fused[0].shift), dim3(256), 0, s, d_img, stride, frameStride, e->d_pyr, P.pyrFrameStride, L0,
                         e->d_padTab + P.fused[0].padOff, P.fused[0].shift);
    } else {
      dim3 g((L0.pitch / 4 + 63) / 64, (L0.h + 4 * kPyrRows - 1) / (4 * kPyrRows), nFrames);
      hipLaunchKernelGGL(k_pyr_level0, g, dim3(256), 0, s, d_img, stride, frameStride, e->d_pyr, P.pyrFrameStride, L0);
    }
    for (int l = 1; l < D.nLevels; l++) {
      const LevelDev& L = D.lv[l];
      const double scaleX = 1. / ((double)L.w / D.lv[l - 1].w);
      if (P.fused[l].on) {
        const HostPlan::FusedLevel& F = P.fused[l];
        hipLaunchKernelGGL(k_pyr_resize_f, fusedGrid(L, F.shift), dim3(256), 0, s, e->d_pyr, P.pyrFrameStride, D.lv[l - 1], L, scaleX,
                           e->d_colTab + F.colOff, e->d_padTab + F.padOff, e->d_rowTab + F.rowOff, F.shift);
      } else {
        dim3 gl((L.pitch / 4 + 63) / 64, (L.h + 4 * kPyrRows - 1) / (4 * kPyrRows), nFrames);
        hipLaunchKernelGGL(k_pyr_resize, gl, dim3(256), 0, s, e->d_pyr, P.pyrFrameStride, D.lv[l - 1], L, scaleX,
                           e->d_tabShort + P.tab[l].alpha, e->d_tabInt + P.tab[l].yofs, e->d_tabShort + P.tab[l].beta);
      }
    }
    if (D.borderBegin[D.nLevels] > 0)
      hipLaunchKernelGGL(k_pyr_borders, dim3(D.borderBegin[D.nLevels] / 256, nFrames), dim3(256), 0, s, e->d_pyr, P.pyrFrameStride, D);
  }
  if (prof) HIPCHK(hipEventRecord(e->ev[1], s));
  hipEvent_t fastEv[kMaxLevels]{};
  int grpFirst[kMaxLevels]{}, grpEnd[kMaxLevels]{}, nGroups = 0;   // level ranges of the FAST launches (and of the quad-tree launches)
  {
    // one wave per cell, 4 cells per workgroup; the per-wave LDS (tile, score map, candidate list) is sized for the plan's largest cell
    const int thr = std::min(std::max(e->cfg.ini_fast_thr, 0), 255);
    // LDS of a launch: sized for the largest cell of the levels it covers (one level with 40-px cells - 1241 x 376: level 5 - would
    // otherwise put every launch on the 80-byte tile pitch and 66 KB per workgroup)
    auto fastGeom = [&](int la, int lb, FastLds& fl, bool& narrow) -> size_t {
      int dim = 1;
      for (int l = la; l < lb; l++) dim = std::max(dim, P.levelCellDim[l]);
      const int maxT = dim + 6, maxB = dim;
      narrow = maxT <= 44;
      const int pitch = narrow ? 48 : 80;
      fl.tileBytes = alignUp((maxT + 3) * pitch, 16);                 // + 3 rows: lanes outside the band still read (and discard) a ring
      fl.scoreBytes = alignUp((maxB + 2) * (maxB + 2) + 4, 16);
      fl.listBytes = alignUp(2 * maxB * maxB, 16);
      return (size_t)4 * (fl.tileBytes + fl.scoreBytes + fl.listBytes);
    };
    {
      FastLds flAll; bool nAll;
      const size_t dynAll = fastGeom(0, D.nLevels, flAll, nAll);
      if (dynAll > 48 * 1024) {   // the largest any group can ask for
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_fast_cells<48>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dynAll));
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_fast_cells<80>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dynAll));
      }
    }
    // The cells are launched in level groups - the first fastGroups - 1 levels each on their own, the rest together - so that the
    // quad-tree of a big level (its own stream, below) starts while the cells of the smaller levels are still being searched.
    int l0 = 0;
    for (int g = 0; g < e->fastGroups && l0 < D.nLevels; g++) {
      const int l1 = g == e->fastGroups - 1 ? D.nLevels : l0 + 1;
      const int c0 = D.lv[l0].cellBegin, c1 = D.lv[l1 - 1].cellBegin + D.lv[l1 - 1].nCells;
      if (c1 > c0 && D.nCellsTotal > 0) {
        FastLds fl; bool narrow;
        const size_t dyn = fastGeom(l0, l1, fl, narrow);
        const dim3 grid(((c1 - c0 + 3) / 4 + 7) / 8 * 8, nFrames);
        if (narrow) hipLaunchKernelGGL(k_fast_cells<48>, grid, dim3(256), dyn, s, e->d_pyr, P.pyrFrameStride, D, e->d_cells, c0, c1, thr, fl, e->d_cellCount, e->d_cellCand);
        else hipLaunchKernelGGL(k_fast_cells<80>, grid, dim3(256), dyn, s, e->d_pyr, P.pyrFrameStride, D, e->d_cells, c0, c1, thr, fl, e->d_cellCount, e->d_cellCand);
      }
      HIPCHK(hipEventRecord(e->evFast[l0], s));
      fastEv[l0] = e->evFast[l0];
      grpFirst[nGroups] = l0; grpEnd[nGroups] = l1; nGroups++;
      for (int l = l0 + 1; l < l1; l++) fastEv[l] = e->evFast[l0];
      l0 = l1;
    }
  }
  if (prof) HIPCHK(hipEventRecord(e->ev[2], s));
  // fork: the blur (needs only the pyramid) goes first on `s` so that it heads its hardware queue; the quad-tree launches, one
  // per level on the side streams, wait for the FAST results only and overlap it
  HIPCHK(hipEventRecord(e->evFork, s));
  hipLaunchKernelGGL(k_blur, dim3((D.blurTileBegin[D.nLevels] + 7) / 8 * 8, nFrames), dim3(256), 0, s, e->d_pyr, P.pyrFrameStride, e->d_blur,
                     P.blurFrameStride, D);
  if (prof) HIPCHK(hipEventRecord(e->ev[3], s));   // end of the blur; the quad-tree stage is the interval up to the join below
  // Quad-tree thinning: ONE launch of k_qt_fast per FAST level group (grid = frames x the group's levels), on the group's side stream or
  // - single-stream handles - on `s` behind the blur.  A group whose quotas need more LDS than a workgroup may have (thousands of
  // features on few levels) runs the pass kernel level by level instead.
  bool anyFast = false;
  for (int g = 0; g < nGroups; g++) {
    const int l0 = grpFirst[g], l1 = grpEnd[g];
    const hipStream_t qs = e->qtInline ? s : e->qtStream[g % kMaxLevels];
    if (!e->qtInline) HIPCHK(hipStreamWaitEvent(qs, fastEv[l0] ? fastEv[l0] : e->evFork, 0));
    int items = 8, qmax = 1, tabLen = 2;
    for (int l = l0; l < l1; l++) {
      items = std::max(items, P.qt[l].items); qmax = std::max(qmax, D.lv[l].quota);
      tabLen = std::max(tabLen, std::max(D.lv[l].w - 2 * kBorder, 0) + std::max(D.lv[l].h - 2 * kBorder, 0) + 2);
    }
    const size_t lds = qt_fast_lds_bytes(qmax, tabLen);
    if (!e->forcePassQuadtree && lds <= 150 * 1024) {
      const dim3 grid(nFrames, l1 - l0);
#define YD_QT_FAST(IT) hipLaunchKernelGGL(k_qt_fast<IT>, grid, dim3(kQfThreads), lds, qs, D, e->d_cellCount, e->d_cellCand, l0, qmax, tabLen, e->d_lvlKp, \
                                          e->d_lvlCount, e->d_needPass, e->d_lvlMaxN, e->d_passList, e->d_passList + 1)
      if (items <= 8) YD_QT_FAST(8);
      else if (items <= 16) YD_QT_FAST(16);
      else YD_QT_FAST(32);
#undef YD_QT_FAST
      anyFast = true;
    } else {
      for (int l = l0; l < l1; l++) {
        const HostPlan::QtLevel& Q = P.qt[l];
        hipLaunchKernelGGL(k_quadtree, dim3(1, nFrames), dim3(kQtThreads), Q.lds, qs, D, e->d_cellCount, e->d_cellCand, e->d_qtCand, e->d_qtNode,
                           P.qtFrameStride, Q.nodeCap, Q.candCap, l, e->d_lvlKp, e->d_lvlCount, e->d_status, e->d_nodeScratch);
      }
    }
    if (!e->qtInline) HIPCHK(hipEventRecord(e->evJoin[g], qs));
  }
  if (!e->qtInline) for (int g = 0; g < nGroups; g++) HIPCHK(hipStreamWaitEvent(s, e->evJoin[g], 0));
  if (anyFast) {  // units k_qt_fast handed over (rare): node table in LDS, candidates in HBM scratch; the list's counter is cleared for the next call
    const int seen = e->h_lvlMaxN[kMaxLevels];   // hand-over list length of an earlier call (copied back below)
    const int listGrid = std::max(1, std::min({kQtPassWorkgroups, D.nLevels * nFrames, seen + seen / 4}));
    hipLaunchKernelGGL(k_quadtree_list, dim3(listGrid), dim3(kQtListThreads), P.qtPassLds, s, D, e->d_cellCount, e->d_cellCand,
                       e->d_qtCand, e->d_qtNode, P.qtFrameStride, -std::max(P.qtPassNodes, 1), e->d_lvlKp, e->d_lvlCount, e->d_status, e->d_passList,
                       e->d_passList + 1, e->d_nodeScratch, e->d_lvlMaxN + kMaxLevels);
    HIPCHK(hipMemsetAsync(e->d_passList, 0, sizeof(int), s));
  }
  if (prof) HIPCHK(hipEventRecord(e->ev[4], s));
  {
    const int kpw = e->descKpw;   // keypoints per wave (YDORB_DESC_KPW: 1, 2 or 4)
    const dim3 gd(((D.sumQuota + 4 * kpw - 1) / (4 * kpw) + 7) / 8 * 8, nFrames);
#define YD_DESC(K) hipLaunchKernelGGL((k_orient_describe_n<K>), gd, dim3(256), 0, s, e->d_pyr, P.pyrFrameStride, e->d_blur, P.blurFrameStride, D, \
                                         e->d_lvlKp, e->d_lvlCount, d_kps, d_desc, cap, d_nOut, e->d_lvlAngle)
    if (kpw == 1)
      hipLaunchKernelGGL(k_orient_describe, gd, dim3(256), 0, s, e->d_pyr, P.pyrFrameStride, e->d_blur,
                         P.blurFrameStride, D, e->d_lvlKp, e->d_lvlCount, d_kps, d_desc, cap, d_nOut, e->d_lvlAngle);
    else if (kpw == 2) YD_DESC(2);
    else YD_DESC(4);
#undef YD_DESC
  }
  if (prof) { HIPCHK(hipEventRecord(e->ev[5], s)); e->profPending = true; }
  HIPCHK(hipMemcpyAsync(e->h_lvlMaxN, e->d_lvlMaxN, sizeof(int) * (kMaxLevels + 1), hipMemcpyDeviceToHost, s));
  HIPCHK(hipGetLastError());
  e->lastFrames = nFrames;
  e->lastStream = s;
  return YDORB_OK;
}

int checkStatus(ydorb_extractor* e) {
  if (*e->h_status == 1) { set_error("more than 65535 FAST candidates in one pyramid level"); return YDORB_ERR_CAPACITY; }
  if (*e->h_status == 3) { set_error("a pyramid level needs the pass quad-tree kernel (more candidates than LDS slots, or a very deep tree) but its quota is too large for that kernel's LDS node table"); return YDORB_ERR_CAPACITY; }
  if (*e->h_status) { set_error("quad-tree node table overflow (status %d)", *e->h_status); return YDORB_ERR_CAPACITY; }
  return YDORB_OK;
}

void collectProfile(ydorb_extractor* e) {
  if (!e->profiling || !e->profPending) return;
  e->profPending = false;
  for (int i = 0; i < ST_COUNT; i++) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, e->ev[i], e->ev[i + 1]) == hipSuccess) e->stageMs[i] += ms;
  }
  e->stageCalls++;
}

}  // namespace

int ydorb::extractor_pyramid_view(const ydorb_extractor* e, PyramidView* v) {
  if (!e || !v || !e->planValid || e->lastFrames < 1) { set_error("the extractor holds no pyramid (no call yet)"); return YDORB_ERR_INVALID_ARG; }
  v->device = e->cfg.device; v->nLevels = e->cfg.n_levels; v->frames = e->lastFrames; v->frameStride = (long long)e->plan.pyrFrameStride;
  v->stream = e->stream;
  for (int l = 0; l < e->cfg.n_levels; l++) {
    const LevelDev& L = e->plan.dev.lv[l];
    v->roi[l] = e->d_pyr + L.padOff + (size_t)kPad * L.pitch + kPad;
    v->w[l] = L.w; v->h[l] = L.h; v->pitch[l] = L.pitch;
    v->scale[l] = e->sf[l]; v->invScale[l] = e->isf[l];
    v->quota[l] = e->perLevel[l];
  }
  return YDORB_OK;
}

extern "C" {

int ydorb_extractor_create(const YdExtractorConfig* cfg, ydorb_extractor_t** out) {
  if (!cfg || !out) { set_error("null argument"); return YDORB_ERR_INVALID_ARG; }
  *out = nullptr;
  if (cfg->n_levels < 1 || cfg->n_levels > kMaxLevels || cfg->n_features < 1 || !(cfg->scale_factor > 1.0f)) {
    set_error("unsupported extractor config (n_levels 1..8, n_features >= 1, scale_factor > 1)");
    return YDORB_ERR_INVALID_ARG;
  }
  int rc = ydorb::require_device(cfg->device);
  if (rc) return rc;
  ydorb_extractor* e = new ydorb_extractor();
  e->cfg = *cfg;
  if (const char* v = getenv("YDORB_QT_PASS")) e->forcePassQuadtree = v[0] == '1';
  if (const char* v = getenv("YDORB_PYR_FUSED")) e->noFusedPads = v[0] == '0';
  // keypoints per wave of the orientation + descriptor kernel: 4 for batches (their loads overlap: 0.56 -> 0.47 ms per 512 frames); a
  // single-frame handle keeps one keypoint per wave (1000 short waves spread over the chip finish sooner than 250 long ones)
  e->descKpw = cfg->max_batch <= 8 ? 1 : 4;
  if (const char* v = getenv("YDORB_DESC_KPW")) e->descKpw = atoi(v) == 1 ? 1 : atoi(v) == 4 ? 4 : 2;
  e->cfg.min_fast_thr = cfg->ini_fast_thr;  // reference quirk, orbExtractor.cpp:318
  // constructor tables, orbExtractor.cpp:319-353
  const int L = cfg->n_levels;
  const float s = cfg->scale_factor;
  int per = (int)round(cfg->n_features * (1 - 1.0 / s) / (1.0 - pow(1.0 / s, L)));
  int sum = 0;
  for (int i = 0; i < L; i++) {
    e->sf.push_back((float)pow(s, i));
    e->sf2.push_back((float)pow(s, 2 * i));
    e->isf.push_back((float)pow(s, -i));
    e->isf2.push_back((float)pow(s, -2 * i));
    if (i < L - 1) {
      e->perLevel.push_back(per);
      sum += per;
      per = (int)round((float)per / s);
    } else {
      e->perLevel.push_back(std::max(cfg->n_features - sum, 0));
    }
  }
  e->sumQuota = 0;
  for (int q : e->perLevel) e->sumQuota += q;
  {  // m_v_maxXcords as the reference actually builds it (resize(16) followed by push_back, :341-353)
    const int half = 15;
    std::vector<int> mx(half + 1, 0);
    const int maxY = (int)floor(half * sqrt(2.0) / 2.0 + 1.0), minY = (int)ceil(half * sqrt(2.0) / 2.0);
    for (int v = 0; v <= maxY; v++) mx.push_back((int)round(sqrt((double)half * half + (double)v * v)));
    for (int v = half, i = 0; v >= minY; v--) {
      while (mx[i] == mx[i + 1]) i++;
      mx[v] = i;
      i++;
    }
    for (int v = 0; v < 16; v++) e->maxX[v] = mx[v];
  }
  if (hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) != hipSuccess) {
    set_error("hipStreamCreate failed");
    delete e;
    return YDORB_ERR_HIP;
  }
  for (auto& ev : e->ev) (void)hipEventCreate(&ev);
  (void)hipEventCreateWithFlags(&e->evFork, hipEventDisableTiming);
  for (int l = 0; l < kMaxLevels; l++) {
    // The per-level quad-tree launches share TWO side streams round-robin (YDORB_QT_STREAMS overrides: 1..8).  The device exposes 4
    // hardware queues; with one stream per level (8 + the handle's own + the caller's matcher stream) streams share queues and the
    // kernels of streams on one queue run one after the other.  Measured on the 512-frame pipeline with one handle: 8 side streams
    // 157 Mkeypoints/s, 4: 160, 2: 163, 1: 151 (the level-0 unit no longer overlaps the small levels).
    // A handle for single frames (max_batch <= 8: the adapter's use) keeps one stream per level: its launches are latency bound and the
    // eight units of a frame then run side by side (0.27 ms per extractAndCompute against 0.33 ms with two).
    const char* qtEnv = getenv("YDORB_QT_STREAMS");
    e->qtInline = (qtEnv && atoi(qtEnv) == 0) || (cfg->flags & YDORB_EXTRACTOR_SINGLE_STREAM);
    const int nQt = std::max(1, std::min(qtEnv ? atoi(qtEnv) : (cfg->max_batch <= 8 ? (int)kMaxLevels : 2), (int)kMaxLevels));
    if (l < nQt) (void)hipStreamCreateWithFlags(&e->qtStream[l], hipStreamNonBlocking);
    else e->qtStream[l] = e->qtStream[l % nQt];
    (void)hipEventCreateWithFlags(&e->evJoin[l], hipEventDisableTiming);
    (void)hipEventCreateWithFlags(&e->evFast[l], hipEventDisableTiming);
  }
  {
    const char* g = getenv("YDORB_FAST_GROUPS");
    // batches: level 0 on its own, then the rest - the quad-tree of level 0 (the longest of the eight) starts ~0.5 ms earlier and the
    // chain no longer outlasts the blur when a second handle's kernels share the GPU (alternate-step pipelining: 188 -> 195 Mkeypoints/s)
    e->fastGroups = std::max(1, std::min(g ? atoi(g) : (cfg->max_batch <= 8 ? 1 : 2), (int)kMaxLevels));
  }
  *out = e;
  return YDORB_OK;
}

void ydorb_extractor_destroy(ydorb_extractor_t* e) {
  if (!e) return;
  (void)hipSetDevice(e->cfg.device);
  if (e->stream) (void)hipStreamSynchronize(e->stream);
  freeBuffers(e);
  for (auto& ev : e->ev) if (ev) (void)hipEventDestroy(ev);
  for (int l = 0; l < kMaxLevels; l++) {
    if (e->qtStream[l]) {
      bool shared = false;
      for (int k = 0; k < l; k++) shared = shared || e->qtStream[k] == e->qtStream[l];
      if (!shared) { (void)hipStreamSynchronize(e->qtStream[l]); (void)hipStreamDestroy(e->qtStream[l]); }
    }
    if (e->evJoin[l]) (void)hipEventDestroy(e->evJoin[l]);
    if (e->evFast[l]) (void)hipEventDestroy(e->evFast[l]);
  }
  if (e->evFork) (void)hipEventDestroy(e->evFork);
  if (e->stream) (void)hipStreamDestroy(e->stream);
  delete e;
}

int ydorb_extractor_tables(const ydorb_extractor_t* e, float* scale, float* inv_scale, float* scale_sq, float* inv_scale_sq,
                           int32_t* per_level) {
  if (!e) { set_error("null handle"); return YDORB_ERR_INVALID_ARG; }
  for (int i = 0; i < e->cfg.n_levels; i++) {
    if (scale) scale[i] = e->sf[i];
    if (inv_scale) inv_scale[i] = e->isf[i];
    if (scale_sq) scale_sq[i] = e->sf2[i];
    if (inv_scale_sq) inv_scale_sq[i] = e->isf2[i];
    if (per_level) per_level[i] = e->perLevel[i];
  }
  return YDORB_OK;
}

int ydorb_extractor_max_keypoints(const ydorb_extractor_t* e) { return e ? e->sumQuota : YDORB_ERR_INVALID_ARG; }

int ydorb_extract_batch_device(ydorb_extractor_t* e, const uint8_t* d_img, int32_t w, int32_t h, int32_t stride, size_t frame_stride,
                               int32_t n_frames, YdKeyPoint* d_kps, uint8_t* d_desc, int32_t cap, int32_t* d_n_out, void* stream) {
  if (!e || !d_img || !d_kps || !d_desc || !d_n_out || w <= 0 || h <= 0 || n_frames <= 0 || stride < w) {
    set_error("invalid argument");
    return YDORB_ERR_INVALID_ARG;
  }
  if (cap < e->sumQuota) { set_error("cap %d < max keypoints %d", cap, e->sumQuota); return YDORB_ERR_CAPACITY; }
  HIPCHK(hipSetDevice(e->cfg.device));
  int rc = ensurePlan(e, w, h, n_frames);
  if (rc) return rc;
  return enqueue(e, d_img, stride, frame_stride, n_frames, reinterpret_cast<YdKeyPointDev*>(d_kps), d_desc, cap, d_n_out,
                 stream ? (hipStream_t)stream : e->stream);
}

int ydorb_extractor_synchronize(ydorb_extractor_t* e) {
  if (!e) { set_error("null handle"); return YDORB_ERR_INVALID_ARG; }
  HIPCHK(hipSetDevice(e->cfg.device));
  HIPCHK(hipStreamSynchronize(e->stream));
  if (e->lastStream && e->lastStream != e->stream) HIPCHK(hipStreamSynchronize(e->lastStream));
  if (!e->d_status) return YDORB_OK;
  // the device-resident entry point reports capacity errors of its quad-tree kernels here (the host entry point reads the
  // same word itself): candidates > 65535 in a level, node-table overflow, a level the pass kernel cannot take
  HIPCHK(hipMemcpy(e->h_status, e->d_status, sizeof(int), hipMemcpyDeviceToHost));
  const int rc = checkStatus(e);
  if (rc) HIPCHK(hipMemset(e->d_status, 0, sizeof(int)));   // reported once; later calls start clean
  return rc;
}

int ydorb_extract_batch(ydorb_extractor_t* e, const uint8_t* img, int32_t w, int32_t h, int32_t stride, size_t frame_stride,
                        int32_t n_frames, YdKeyPoint* kps, uint8_t* desc, int32_t cap, int32_t* n_out) {
  if (!e || !n_out) { set_error("null argument"); return YDORB_ERR_INVALID_ARG; }
  if (!img || w <= 0 || h <= 0 || n_frames <= 0) {  // empty image: silent return, orbExtractor.cpp:357-359
    for (int f = 0; f < std::max(n_frames, 0); f++) n_out[f] = 0;
    return YDORB_OK;
  }
  if (!kps || !desc || stride < w) { set_error("invalid argument"); return YDORB_ERR_INVALID_ARG; }
  HIPCHK(hipSetDevice(e->cfg.device));
  int rc = ensurePlan(e, w, h, n_frames);
  if (rc) return rc;
  const int Q = e->sumQuota;
  for (int f = 0; f < n_frames; f++)
    for (int y = 0; y < h; y++) memcpy(e->h_img + (size_t)f * e->imgBytes + (size_t)y * w, img + f * frame_stride + (size_t)y * stride, w);
  HIPCHK(hipMemcpyAsync(e->d_img, e->h_img, e->imgBytes * n_frames, hipMemcpyHostToDevice, e->stream));
  rc = enqueue(e, e->d_img, w, e->imgBytes, n_frames, e->d_kps, e->d_desc, Q, e->d_nOut, e->stream);
  if (rc) return rc;
  HIPCHK(hipMemcpyAsync(e->h_nOut, e->d_nOut, sizeof(int) * n_frames, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipMemcpyAsync(e->h_status, e->d_status, sizeof(int), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipMemcpyAsync(e->h_kps, e->d_kps, sizeof(YdKeyPoint) * (size_t)Q * n_frames, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipMemcpyAsync(e->h_desc, e->d_desc, (size_t)32 * Q * n_frames, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  collectProfile(e);
  rc = checkStatus(e);
  if (rc) {
    HIPCHK(hipMemset(e->d_status, 0, sizeof(int)));   // reported once; the next call starts clean
    return rc;
  }
  for (int f = 0; f < n_frames; f++) {
    const int n = e->h_nOut[f];
    if (n > cap) { set_error("frame %d has %d keypoints, caller capacity %d", f, n, cap); return YDORB_ERR_CAPACITY; }
    memcpy(kps + (size_t)f * cap, e->h_kps + (size_t)f * Q, sizeof(YdKeyPoint) * n);
    memcpy(desc + (size_t)f * cap * 32, e->h_desc + (size_t)f * Q * 32, (size_t)32 * n);
    n_out[f] = n;
  }
  return YDORB_OK;
}

int ydorb_extract(ydorb_extractor_t* e, const uint8_t* img, int32_t w, int32_t h, int32_t stride, YdKeyPoint* kps, uint8_t* desc,
                  int32_t cap, int32_t* n_out) {
  return ydorb_extract_batch(e, img, w, h, stride, 0, 1, kps, desc, cap, n_out);
}

int ydorb_extractor_pyramid(const ydorb_extractor_t* e, int32_t frame, int32_t level, const uint8_t** d_roi, int32_t* w, int32_t* h,
                            int32_t* stride) {
  if (!e || !e->planValid || frame < 0 || frame >= e->lastFrames || level < 0 || level >= e->cfg.n_levels) {
    set_error("no pyramid for frame %d level %d", frame, level);
    return YDORB_ERR_INVALID_ARG;
  }
  const LevelDev& L = e->plan.dev.lv[level];
  if (d_roi) *d_roi = e->d_pyr + (size_t)frame * e->plan.pyrFrameStride + L.padOff + (size_t)kPad * L.pitch + kPad;
  if (w) *w = L.w;
  if (h) *h = L.h;
  if (stride) *stride = L.pitch;
  return YDORB_OK;
}

int ydorb_extractor_read_level(ydorb_extractor_t* e, int32_t frame, int32_t level, uint8_t* dst, size_t dst_bytes) {
  const uint8_t* roi;
  int w, h, stride;
  int rc = ydorb_extractor_pyramid(e, frame, level, &roi, &w, &h, &stride);
  if (rc) return rc;
  const size_t need = (size_t)(w + 2 * kPad) * (h + 2 * kPad);
  if (!dst || dst_bytes < need) { set_error("need %zu bytes", need); return YDORB_ERR_CAPACITY; }
  HIPCHK(hipSetDevice(e->cfg.device));
  HIPCHK(hipStreamSynchronize(e->stream));
  HIPCHK(hipMemcpy2D(dst, w + 2 * kPad, roi - (size_t)kPad * stride - kPad, stride, w + 2 * kPad, h + 2 * kPad, hipMemcpyDeviceToHost));
  return YDORB_OK;
}

int ydorb_extractor_read_pyramid(ydorb_extractor_t* e, int32_t frame, uint8_t* const* dst_levels, const size_t* dst_bytes, int32_t n_levels) {
  if (!e || !e->planValid || frame < 0 || frame >= e->lastFrames || !dst_levels || !dst_bytes || n_levels < 1 || n_levels > e->cfg.n_levels) {
    set_error("invalid argument");
    return YDORB_ERR_INVALID_ARG;
  }
  const HostPlan& P = e->plan;
  for (int l = 0; l < n_levels; l++) {
    const LevelDev& L = P.dev.lv[l];
    const size_t need = (size_t)(L.w + 2 * kPad) * (L.h + 2 * kPad);
    if (!dst_levels[l] || dst_bytes[l] < need) { set_error("level %d needs %zu bytes", l, need); return YDORB_ERR_CAPACITY; }
  }
  HIPCHK(hipSetDevice(e->cfg.device));
  // ONE device-to-host copy of the frame's pyramid block (levels sit back to back, rows at a 64-byte pitch) into pinned staging,
  // then the rows are repacked on the host: eight pitched copies into pageable memory cost ~1.6 ms each.
  if (e->h_pyrBytes < P.pyrFrameStride) {
    if (e->h_pyr) (void)hipHostFree(e->h_pyr);
    e->h_pyr = nullptr; e->h_pyrBytes = 0;
    HIPCHK(hipHostMalloc(&e->h_pyr, P.pyrFrameStride));
    e->h_pyrBytes = P.pyrFrameStride;
  }
  hipStream_t s = e->lastStream ? e->lastStream : e->stream;
  HIPCHK(hipMemcpyAsync(e->h_pyr, e->d_pyr + (size_t)frame * P.pyrFrameStride, P.pyrFrameStride, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  for (int l = 0; l < n_levels; l++) {
    const LevelDev& L = P.dev.lv[l];
    const int rowBytes = L.w + 2 * kPad;
    for (int y = 0; y < L.h + 2 * kPad; y++) memcpy(dst_levels[l] + (size_t)y * rowBytes, e->h_pyr + L.padOff + (size_t)y * L.pitch, rowBytes);
  }
  return YDORB_OK;
}

int ydorb_extractor_debug_read(ydorb_extractor_t* e, int32_t what, int32_t frame, int32_t level, void* dst, size_t dst_bytes,
                               size_t* written) {
  if (!e || !e->planValid || frame < 0 || frame >= e->lastFrames || level < 0 || level >= e->cfg.n_levels || !dst || !written) {
    set_error("invalid argument");
    return YDORB_ERR_INVALID_ARG;
  }
  HIPCHK(hipSetDevice(e->cfg.device));
  HIPCHK(hipStreamSynchronize(e->stream));
  const HostPlan& P = e->plan;
  const PlanDev& D = P.dev;
  const LevelDev& L = D.lv[level];
  if (what == 0) {
    const size_t need = (size_t)L.w * L.h;
    if (dst_bytes < need) { set_error("need %zu bytes", need); return YDORB_ERR_CAPACITY; }
    HIPCHK(hipMemcpy2D(dst, L.w, e->d_blur + (size_t)frame * P.blurFrameStride + L.blurOff, L.blurPitch, L.w, L.h, hipMemcpyDeviceToHost));
    *written = need;
    return YDORB_OK;
  }
  if (what == 1) {
    std::vector<uint32_t> cnt(L.nCells), buf((size_t)L.nCells * D.cellCap);
    if (L.nCells) {
      HIPCHK(hipMemcpy(cnt.data(), e->d_cellCount + (size_t)frame * D.nCellsTotal + L.cellBegin, sizeof(uint32_t) * L.nCells, hipMemcpyDeviceToHost));
      HIPCHK(hipMemcpy(buf.data(), e->d_cellCand + ((size_t)frame * D.nCellsTotal + L.cellBegin) * D.cellCap,
                       sizeof(uint32_t) * buf.size(), hipMemcpyDeviceToHost));
    }
    YdKeyPoint* o = (YdKeyPoint*)dst;
    size_t n = 0;
    for (int c = 0; c < L.nCells; c++)
      for (uint32_t i = 0; i < cnt[c]; i++) {
        if ((n + 1) * sizeof(YdKeyPoint) > dst_bytes) { set_error("candidate buffer too small"); return YDORB_ERR_CAPACITY; }
        const uint32_t pk = buf[(size_t)c * D.cellCap + i];
        o[n++] = YdKeyPoint{(float)qt_x(pk), (float)qt_y(pk), 7.f, -1.f, (float)qt_r(pk), 0, -1};
      }
    *written = n * sizeof(YdKeyPoint);
    return YDORB_OK;
  }
  if (what == 2) {
    int counts[kMaxLevels];
    HIPCHK(hipMemcpy(counts, e->d_lvlCount + frame * kMaxLevels, sizeof(counts), hipMemcpyDeviceToHost));
    const int n = counts[level];
    if ((size_t)n * sizeof(YdKeyPoint) > dst_bytes) { set_error("keypoint buffer too small"); return YDORB_ERR_CAPACITY; }
    std::vector<uint32_t> pk(std::max(n, 1));
    std::vector<float> ang(std::max(n, 1));
    if (n) {
      HIPCHK(hipMemcpy(pk.data(), e->d_lvlKp + (size_t)frame * D.sumQuota + L.kpOff, sizeof(uint32_t) * n, hipMemcpyDeviceToHost));
      HIPCHK(hipMemcpy(ang.data(), e->d_lvlAngle + (size_t)frame * D.sumQuota + L.kpOff, sizeof(float) * n, hipMemcpyDeviceToHost));
    }
    YdKeyPoint* o = (YdKeyPoint*)dst;
    for (int i = 0; i < n; i++)
      o[i] = YdKeyPoint{(float)(qt_x(pk[i]) + kBorder), (float)(qt_y(pk[i]) + kBorder), L.size, ang[i], (float)qt_r(pk[i]), level, -1};
    *written = (size_t)n * sizeof(YdKeyPoint);
    return YDORB_OK;
  }
  if (what == 3) {   // which quad-tree kernel produced the unit: 0 = flat, 1 = pass
    if (dst_bytes < 1) { set_error("need 1 byte"); return YDORB_ERR_CAPACITY; }
    uint8_t v = 1;
    int qmaxAll = 1, tabAll = 2;
    for (int l = 0; l < D.nLevels; l++) { qmaxAll = std::max(qmaxAll, D.lv[l].quota); tabAll = std::max(tabAll, std::max(D.lv[l].w - 2 * kBorder, 0) + std::max(D.lv[l].h - 2 * kBorder, 0) + 2); }
    if (!e->forcePassQuadtree && qt_fast_lds_bytes(qmaxAll, tabAll) <= 150 * 1024)   // (conservative: a group's own largest quota decides at launch)
      HIPCHK(hipMemcpy(&v, e->d_needPass + (size_t)frame * kMaxLevels + level, 1, hipMemcpyDeviceToHost));
    *(uint8_t*)dst = v;
    *written = 1;
    return YDORB_OK;
  }
  set_error("unknown debug stage %d", what);
  return YDORB_ERR_INVALID_ARG;
}

int ydorb_extractor_set_profiling(ydorb_extractor_t* e, int32_t on) {
  if (!e) return YDORB_ERR_INVALID_ARG;
  e->profiling = on != 0;
  for (double& m : e->stageMs) m = 0;
  e->stageCalls = 0;
  return YDORB_OK;
}

int ydorb_extractor_stage_times(ydorb_extractor_t* e, int32_t max_stages, const char** names, float* ms, int32_t* n_stages) {
  if (!e || !n_stages) return YDORB_ERR_INVALID_ARG;
  if (e->profiling && e->profPending) {  // device-resident path: the last launch's events (may be on a caller's stream)
    if (hipEventSynchronize(e->ev[ST_COUNT]) == hipSuccess) collectProfile(e);
  }
  const int n = std::min<int>(max_stages, ST_COUNT);
  for (int i = 0; i < n; i++) {
    if (names) names[i] = kStageNames[i];
    if (ms) ms[i] = e->stageCalls ? (float)(e->stageMs[i] / e->stageCalls) : 0.f;
  }
  *n_stages = n;
  return YDORB_OK;
}

}  // extern "C"
