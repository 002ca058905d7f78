// C-ABI implementation (include/rspchain.h): parameter validation, the AXI4-style
// register file, device resources and kernel dispatch.  No CPU compute path:
// every data-plane call ends in a gfx950 kernel launch or fails.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <new>
#include <vector>

#include "../../include/rspchain.h"
#include "chain_regs.hpp"
#include "fft_lds.hpp"
#include "host_pipe.hpp"
#include "kernels.hpp"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

#define HIP_TRY(expr)                                                                   \
  do {                                                                                  \
    hipError_t e_ = (expr);                                                             \
    if (e_ != hipSuccess)                                                               \
      return fail(RSP_ERR_DEVICE, "%s failed: %s", #expr, hipGetErrorString(e_));       \
  } while (0)

bool is_pow2(int64_t x) { return x > 0 && (x & (x - 1)) == 0; }
int ilog2(int64_t x) {
  int l = 0;
  while ((int64_t(1) << (l + 1)) <= x) ++l;
  return l;
}

// CFAR register offsets in beats: FftMagCfarChainTester.scala:100-132
enum CfarReg {
  kFftSize = 0, kScaler = 1, kLogOrLinear = 2, kDivSum = 3, kPeakGrouping = 4, kAlgorithm = 5,
  kMode = 6, kRefWindow = 7, kGuardWindow = 8, kIndexLagg = 9, kIndexLead = 10, kSubWindow = 11,
  kNumCfarRegs = 12
};

struct TwiddleRom {
  void* d = nullptr;
};

}  // namespace

struct rsp_chain {
  rsp_chain_params p;
  // register file
  uint32_t fft_stages;
  uint32_t mag_mode;
  uint32_t cfar[kNumCfarRegs];
  // device state
  int device;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  std::map<int, TwiddleRom> rom;  // keyed by log2n
  std::map<int, void*> win;       // pre-FFT window tables, keyed by 64 * type + log2n
  int16_t* d_log_lut = nullptr;
  void* d_in = nullptr;
  size_t d_in_bytes = 0;
  uint32_t* d_out = nullptr;
  size_t d_out_bytes = 0;
  rsp_detection* d_list = nullptr;
  size_t d_list_cap = 0;
  uint32_t* d_count = nullptr;   // {found, stored} of the host-buffer detection call
  uint32_t* d_ctr = nullptr;     // compaction counters {found, cursor, ticket}: zero between launches
  uint32_t* d_fcount = nullptr;  // per-frame peak counts of the fused path
  uint2* d_fdet = nullptr;       // per-frame peak slots
  size_t fslots = 0;             // frames the two buffers above hold
  void* d_x1 = nullptr;          // 2-D chain: range-pass spectrum
  size_t d_x1_bytes = 0;
  float* d_mag2 = nullptr;       // 2-D chain: magnitude map
  size_t d_mag2_bytes = 0;
  // per-launch HIP-event timing of the chain kernel alone (rsp_chain_profile_*)
  bool profiling = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events;
  size_t prof_used = 0;
  // rsp_chain_set_option
  uint32_t opt_max_frames = 0;
  bool opt_force_tiled = false;
  bool opt_generic_tail = false;
  bool opt_experiment = false;
  size_t opt_rd_chunk_bytes = 0;  // 2-D chain: intermediates per chunk of channels (0 = whole batch: chunking measured slower)
  size_t opt_host_chunk_bytes = 0;  // host-buffer entry: input bytes per pipeline chunk (0 = automatic)
  rsp::HostPipe hp;               // streams / events / pinned staging ring of the host-buffer entry points
  hipEvent_t ev_switch = nullptr; // rsp_chain_set_stream: the new stream waits for what the old one still holds
};

namespace {

bool cfar_reg_present(const rsp_chain_params& p, int reg) {
  const int alg = p.cfarParams.CFARAlgorithm;
  switch (reg) {
    case kDivSum: return alg != RSP_ALG_GOS;                               // Tester:105-108
    case kAlgorithm: return alg == RSP_ALG_GOSCA;                          // Tester:110-118
    case kIndexLagg:
    case kIndexLead: return alg != RSP_ALG_CA;                             // Tester:123-127
    case kSubWindow: return alg == RSP_ALG_CA && p.cfarParams.includeCASH; // Tester:129-132
    default: return reg >= 0 && reg < kNumCfarRegs;
  }
}

int validate(const rsp_chain_params* p) {
  if (!p) return fail(RSP_ERR_INVALID, "params is NULL");
  const rsp_fft_params& f = p->fftParams;
  const rsp_mag_params& m = p->magParams;
  const rsp_cfar_params& c = p->cfarParams;
  if (p->beatBytes != 4)
    return fail(RSP_ERR_UNSUPPORTED, "beatBytes = %d: the chain's streams are 32-bit (FftMagCfarChain.scala:52,62)", p->beatBytes);
  if (p->dtype != RSP_DTYPE_FIXED16 && p->dtype != RSP_DTYPE_F32)
    return fail(RSP_ERR_INVALID, "dtype %d", p->dtype);
  if (!is_pow2(f.numPoints)) return fail(RSP_ERR_INVALID, "numPoints = %d is not a power of two", f.numPoints);
  const int m_max = ilog2(f.numPoints);
  if (m_max < rsp::kMinLog2NSmall || m_max > rsp::kMaxLog2N)
    return fail(RSP_ERR_UNSUPPORTED, "numPoints = %d: the GPU path holds %d..%d-point frames in LDS",
                f.numPoints, 1 << rsp::kMinLog2NSmall, 1 << rsp::kMaxLog2N);
  if (f.dataWidth != 16 || f.twiddleWidth != 16)
    return fail(RSP_ERR_UNSUPPORTED, "dataWidth/twiddleWidth = %d/%d: only the reference's 16/16 is implemented",
                f.dataWidth, f.twiddleWidth);
  bool stage_opts = false;
  for (int s = 0; s < m_max; ++s) {
    if ((f.expandLogic[s] != 0 && f.expandLogic[s] != 1) || (f.keepMSBorLSB[s] != 0 && f.keepMSBorLSB[s] != 1))
      return fail(RSP_ERR_INVALID, "stage %d: expandLogic = %d / keepMSBorLSB = %d (0 or 1 each)", s, f.expandLogic[s], f.keepMSBorLSB[s]);
    stage_opts |= f.expandLogic[s] != 0 || f.keepMSBorLSB[s] != 1;
  }
  if (stage_opts && p->dtype != RSP_DTYPE_FIXED16)
    return fail(RSP_ERR_INVALID, "expandLogic / keepMSBorLSB are properties of the FixedPoint data path (dtype = RSP_DTYPE_FIXED16)");
  if (f.trimType < RSP_TRIM_FLOOR || f.trimType > RSP_TRIM_CONVERGENT) return fail(RSP_ERR_INVALID, "trimType %d", f.trimType);
  if (f.binPoint < 0 || f.binPoint > 15 || m.binPoint != f.binPoint)
    return fail(RSP_ERR_INVALID, "binPoint fft/mag = %d/%d must agree and lie in 0..15", f.binPoint, m.binPoint);
  if (m.dataWidth != 16 || m.dataWidthLog != 16) return fail(RSP_ERR_UNSUPPORTED, "MAGParams widths %d/%d", m.dataWidth, m.dataWidthLog);
  if (m.log2LookUpWidth < 1 || m.log2LookUpWidth > 12 || m.binPointLog < 0 || m.binPointLog > 14)
    return fail(RSP_ERR_INVALID, "log2LookUpWidth/binPointLog = %d/%d", m.log2LookUpWidth, m.binPointLog);
  if (!m.useLast) return fail(RSP_ERR_UNSUPPORTED, "useLast = false: frames are delimited by TLAST on this path");
  if (c.fftSize != f.numPoints) return fail(RSP_ERR_INVALID, "CFARParams.fftSize = %d != FFTParams.numPoints = %d", c.fftSize, f.numPoints);
  if (!is_pow2(c.leadLaggWindowSize) || c.leadLaggWindowSize > rsp::kMaxRef)
    return fail(c.leadLaggWindowSize > rsp::kMaxRef ? RSP_ERR_UNSUPPORTED : RSP_ERR_INVALID,
                "leadLaggWindowSize = %d (power of two <= %d)", c.leadLaggWindowSize, rsp::kMaxRef);
  if (c.guardWindowSize <= 0) return fail(RSP_ERR_INVALID, "guardWindowSize = %d", c.guardWindowSize);
  if (c.CFARAlgorithm < RSP_ALG_CA || c.CFARAlgorithm > RSP_ALG_GOSCA) return fail(RSP_ERR_INVALID, "CFARAlgorithm %d", c.CFARAlgorithm);
  if (c.edgeMode != RSP_EDGE_ZERO && c.edgeMode != RSP_EDGE_WRAP) return fail(RSP_ERR_INVALID, "edgeMode %d", c.edgeMode);
  const rsp_fixed_proto* protos[3] = {&c.protoIn, &c.protoThreshold, &c.protoScaler};
  for (const rsp_fixed_proto* q : protos) {
    if (q->width < 2 || q->width > 16 || q->binaryPoint < 0 || q->binaryPoint > 15)
      return fail(RSP_ERR_INVALID, "FixedPoint(%d.W, %d.BP)", q->width, q->binaryPoint);
  }
  if (c.protoThreshold.width + m_max + 1 > 32)
    return fail(RSP_ERR_INVALID, "threshold width %d + bin width %d + peak bit exceed the 32-bit output beat", c.protoThreshold.width, m_max);
  const rsp_address_set* as[3] = {&p->fftAddress, &p->magAddress, &p->cfarAddress};
  for (int i = 0; i < 3; ++i) {
    if (as[i]->base & as[i]->mask) return fail(RSP_ERR_INVALID, "AddressSet(0x%x, 0x%x): base overlaps mask", as[i]->base, as[i]->mask);
    for (int j = 0; j < i; ++j) {
      const uint32_t m2 = as[i]->mask | as[j]->mask;
      if ((as[i]->base & ~m2) == (as[j]->base & ~m2)) return fail(RSP_ERR_INVALID, "address sets %d and %d overlap", i, j);
    }
  }
  if (p->cfarAddress.mask < 4 * kNumCfarRegs - 1) return fail(RSP_ERR_INVALID, "cfarAddress mask 0x%x too small for 12 registers", p->cfarAddress.mask);
  if (p->dopplerPoints != 0) {  // 2-D range-Doppler chain (no reference counterpart; BASELINE.json configs 3/5)
    if (p->dtype != RSP_DTYPE_F32 && stage_opts)
      return fail(RSP_ERR_UNSUPPORTED, "2-D chain, FIXED16: expandLogic / keepMSBorLSB stage options are defined for the 1-D chain only");
    if (!is_pow2(p->dopplerPoints) || p->dopplerPoints < 256 || p->dopplerPoints > 1024)
      return fail(RSP_ERR_UNSUPPORTED, "dopplerPoints = %d: 256, 512 or 1024", p->dopplerPoints);
    if (m_max > rsp::kMaxLog2N2d)
      return fail(RSP_ERR_UNSUPPORTED, "2-D chain: numPoints = %d, the range FFT holds up to %d points", f.numPoints, 1 << rsp::kMaxLog2N2d);
    if (p->refDoppler < 1 || p->guardDoppler < 0 || p->refDoppler + p->guardDoppler > 32)
      return fail(RSP_ERR_INVALID, "refDoppler/guardDoppler = %d/%d", p->refDoppler, p->guardDoppler);
    if (c.sendCut || !f.useBitReverse)
      return fail(RSP_ERR_UNSUPPORTED, "2-D chain (no reference counterpart): sendCut = true / useBitReverse = false are defined for the 1-D chain only");
  }
  if (p->window < RSP_WINDOW_NONE || p->window > RSP_WINDOW_BLACKMAN || p->windowDoppler < RSP_WINDOW_NONE ||
      p->windowDoppler > RSP_WINDOW_BLACKMAN)
    return fail(RSP_ERR_INVALID, "window / windowDoppler = %d / %d", p->window, p->windowDoppler);
  if (p->windowDoppler && !p->dopplerPoints) return fail(RSP_ERR_INVALID, "windowDoppler needs the 2-D chain (dopplerPoints > 0)");
  return RSP_OK;
}

void reset_regs(rsp_chain* c) {
  // RunTimeRspChainParams() defaults: RspChainVanillaTester.scala:35-48
  const rsp_chain_params& p = c->p;
  const int ref = std::min(32, p.cfarParams.leadLaggWindowSize);
  c->fft_stages = (uint32_t)ilog2(p.fftParams.numPoints);
  c->mag_mode = RSP_MAG_JPL;
  std::memset(c->cfar, 0, sizeof(c->cfar));
  c->cfar[kFftSize] = (uint32_t)p.cfarParams.fftSize;
  c->cfar[kScaler] = (uint32_t)(3.5 * std::ldexp(1.0, p.cfarParams.protoThreshold.binaryPoint));  // Tester:101
  c->cfar[kLogOrLinear] = 1;
  c->cfar[kDivSum] = (uint32_t)ilog2(ref);
  c->cfar[kPeakGrouping] = 0;
  c->cfar[kAlgorithm] = 0;
  c->cfar[kMode] = RSP_MODE_GO;
  c->cfar[kRefWindow] = (uint32_t)ref;
  c->cfar[kGuardWindow] = (uint32_t)std::min(4, p.cfarParams.guardWindowSize);
  c->cfar[kIndexLagg] = (uint32_t)(ref / 2);
  c->cfar[kIndexLead] = (uint32_t)(ref / 2);
  c->cfar[kSubWindow] = 0;
}

bool uses_gos(const rsp_chain* c) {
  const int alg = c->p.cfarParams.CFARAlgorithm;
  return alg == RSP_ALG_GOS || (alg == RSP_ALG_GOSCA && c->cfar[kAlgorithm] == 1);
}

int check_regs(const rsp_chain* c) {
  const rsp_chain_params& p = c->p;
  const int m_max = ilog2(p.fftParams.numPoints);
  const int m = (int)c->fft_stages;
  if (m > m_max || m < 1) return fail(RSP_ERR_INVALID, "FFT stages = %d exceeds log2(numPoints) = %d", m, m_max);
  if (!p.fftParams.runTime && m != m_max) return fail(RSP_ERR_INVALID, "runTime = false but stages register = %d != %d", m, m_max);
  if (m < rsp::kMinLog2NSmall) return fail(RSP_ERR_UNSUPPORTED, "fftSize = %d: the GPU path needs >= %d points", 1 << m, 1 << rsp::kMinLog2NSmall);
  if (m < rsp::kMinLog2N && p.dopplerPoints) return fail(RSP_ERR_UNSUPPORTED, "2-D chain: range FFT needs >= %d points", 1 << rsp::kMinLog2N);
  const int n = 1 << m;
  if ((int)c->cfar[kFftSize] != n)
    return fail(RSP_ERR_INVALID, "CFAR fftSize register = %u but FFT stages register selects %d points", c->cfar[kFftSize], n);
  if (c->mag_mode > 2) return fail(RSP_ERR_INVALID, "magnitude mode register = %u", c->mag_mode);
  const int R = (int)c->cfar[kRefWindow], G = (int)c->cfar[kGuardWindow];
  // require(...)s of RunTimeRspChainParams: RspChainVanillaTester.scala:50-61
  if (!is_pow2(R)) return fail(RSP_ERR_INVALID, "refWindowSize = %d is not a power of two", R);
  if (G <= 0) return fail(RSP_ERR_INVALID, "guardWindowSize = %d must be > 0", G);
  if (R <= G) return fail(RSP_ERR_INVALID, "refWindowSize = %d must exceed guardWindowSize = %d", R, G);
  if (R > p.cfarParams.leadLaggWindowSize) return fail(RSP_ERR_INVALID, "refWindowSize = %d > leadLaggWindowSize = %d", R, p.cfarParams.leadLaggWindowSize);
  if (G > p.cfarParams.guardWindowSize) return fail(RSP_ERR_INVALID, "guardWindowSize = %d > elaborated maximum %d", G, p.cfarParams.guardWindowSize);
  if (R + G > 256) return fail(RSP_ERR_UNSUPPORTED, "refWindowSize + guardWindowSize = %d exceeds the 256-cell LDS halo", R + G);
  if (2 * (R + G) + 1 > n) return fail(RSP_ERR_INVALID, "window 2*(%d+%d)+1 does not fit a %d-point frame", R, G, n);
  if (c->cfar[kMode] > 3) return fail(RSP_ERR_INVALID, "cfarMode register = %u", c->cfar[kMode]);
  if (c->cfar[kMode] == RSP_MODE_CASH) {
    if (!(p.cfarParams.CFARAlgorithm == RSP_ALG_CA && p.cfarParams.includeCASH))
      return fail(RSP_ERR_INVALID, "cfarMode = CASH needs CACFARType with includeCASH = true");
    const int sub = (int)c->cfar[kSubWindow];
    if (sub <= 0 || sub >= R) return fail(RSP_ERR_INVALID, "subWindowSize = %d must lie in 1..refWindowSize-1", sub);
    if (p.cfarParams.minSubWindowSize > 0 && sub < p.cfarParams.minSubWindowSize)
      return fail(RSP_ERR_INVALID, "subWindowSize = %d < minSubWindowSize = %d", sub, p.cfarParams.minSubWindowSize);
    if (p.dopplerPoints) return fail(RSP_ERR_UNSUPPORTED, "2-D chain: CASH is not defined");
  }
  if (cfar_reg_present(p, kDivSum) && c->cfar[kDivSum] > 15) return fail(RSP_ERR_INVALID, "divSum = %u", c->cfar[kDivSum]);
  if (cfar_reg_present(p, kIndexLagg)) {
    if ((int)c->cfar[kIndexLagg] >= R || (int)c->cfar[kIndexLead] >= R)
      return fail(RSP_ERR_INVALID, "indexLagg/indexLead = %u/%u must be < refWindowSize = %d", c->cfar[kIndexLagg], c->cfar[kIndexLead], R);
  }
  if (c->cfar[kAlgorithm] > 1) return fail(RSP_ERR_INVALID, "cfarAlgorithm register = %u", c->cfar[kAlgorithm]);
  if (uses_gos(c)) {
    if (m >= rsp::kMinLog2N && (R < 4 || R > 64)) return fail(RSP_ERR_UNSUPPORTED, "GOS CFAR: refWindowSize = %d, the GPU sorter is built for 4..64", R);
  }
  if (p.dopplerPoints) {
    if (m != m_max) return fail(RSP_ERR_UNSUPPORTED, "2-D chain: run-time FFT size must equal numPoints");
    if (uses_gos(c) || c->cfar[kPeakGrouping]) return fail(RSP_ERR_UNSUPPORTED, "2-D chain: CA / GO / SO without peak grouping only");
    if (R + G > 32) return fail(RSP_ERR_UNSUPPORTED, "2-D chain: range training + guard half-width %d exceeds the 32-cell tile halo", R + G);
  }
  if (c->cfar[kScaler] > 0xFFFFu) return fail(RSP_ERR_INVALID, "thresholdScaler register = 0x%x exceeds protoScaler's 16 bits", c->cfar[kScaler]);
  return RSP_OK;
}

rsp::ChainRegs snapshot(const rsp_chain* c) {
  const rsp_chain_params& p = c->p;
  rsp::ChainRegs r{};
  const int trim = p.fftParams.trimType;
  r.trim_bias1 = trim == RSP_TRIM_FLOOR ? 0 : 1;
  r.trim_bias15 = trim == RSP_TRIM_FLOOR ? 0 : (1 << 14);
  r.trim_conv = trim == RSP_TRIM_CONVERGENT ? 1 : 0;
  r.mag_mode = (int)c->mag_mode;
  r.bp_data = p.magParams.binPoint;
  r.bp_log = p.magParams.binPointLog;
  r.lut_w = p.magParams.log2LookUpWidth;
  r.bp_in = p.cfarParams.protoIn.binaryPoint;
  r.bp_thr = p.cfarParams.protoThreshold.binaryPoint;
  r.w_thr = p.cfarParams.protoThreshold.width;
  r.bp_scaler = p.cfarParams.protoScaler.binaryPoint;
  r.scaler_raw = c->cfar[kScaler];
  r.scaler_f = (float)std::ldexp((double)c->cfar[kScaler], -r.bp_scaler);
  r.linear = c->cfar[kLogOrLinear] ? 1 : 0;
  r.div_sum = cfar_reg_present(p, kDivSum) ? (int)c->cfar[kDivSum] : 0;
  r.div_f = (float)std::ldexp(1.0, -r.div_sum);
  r.peak_grouping = c->cfar[kPeakGrouping] ? 1 : 0;
  r.algorithm = uses_gos(c) ? 1 : 0;
  r.cfar_mode = (int)c->cfar[kMode];
  r.R = (int)c->cfar[kRefWindow];
  r.G = (int)c->cfar[kGuardWindow];
  r.idx_lagg = (int)c->cfar[kIndexLagg];
  r.idx_lead = (int)c->cfar[kIndexLead];
  r.sub_window = (int)c->cfar[kSubWindow];
#ifdef RSP_ABLATE  // tools/ablate.sh side builds only
  if (const char* m = getenv("RSP_ABLATE_MASK")) r.sub_window = atoi(m);
#endif
  r.edge = p.cfarParams.edgeMode;
  for (int s2 = 0; s2 < (int)c->fft_stages && s2 < RSP_MAX_STAGES; ++s2) {
    // run-time FFT size: the active stages are the LAST log2(fftSize) stages of the elaborated pipeline
    // (a smaller transform enters the SDF chain further down), so stage s of the active transform
    // takes the options of elaborated stage s + (log2(numPoints) - stages)  [BUILD-DEFINED, spec section 3]
    const int es = s2 + ilog2(p.fftParams.numPoints) - (int)c->fft_stages;
    if (p.fftParams.expandLogic[es]) r.expand_mask |= 1u << s2;
    else if (!p.fftParams.keepMSBorLSB[es]) r.keep_lsb_mask |= 1u << s2;
  }
  r.growth = __builtin_popcount(r.expand_mask);
  r.rev_order = p.fftParams.useBitReverse ? 0 : 1;
  r.send_cut = p.cfarParams.sendCut ? 1 : 0;
  r.window = nullptr;
  {
    const int lin = r.bp_in + r.bp_scaler - r.bp_thr;  // oracle: trim_shift(stat * scaler, lin)
    r.lin_shr = lin > 0 ? lin : 0;
    r.lin_shl = lin < 0 ? -lin : 0;
    const int lg = r.bp_in - r.bp_thr;
    r.log_shr = lg > 0 ? lg : 0;
    r.log_shl = lg < 0 ? -lg : 0;
    const int ls = r.bp_scaler - r.bp_thr;
    r.log_scaler = ls >= 0 ? (int32_t)(r.scaler_raw >> ls) : (int32_t)(r.scaler_raw << -ls);
    r.tmax = (1 << (r.w_thr - 1)) - 1;
    r.tmin = -(1 << (r.w_thr - 1));
    // bound of |statistic|: magnitudes are <= 32767 (linear modes) or 16 << binPointLog (log2 mode); a side sum holds
    // R of them before divSum (the 2-D chain sums its whole training region: bounded by the same product of its sizes)
    const int64_t mag_max = std::max<int64_t>(32767, int64_t(16) << r.bp_log);
    int64_t cells = r.algorithm == 1 ? 1 : r.R;
    if (p.dopplerPoints) cells = int64_t(2 * (r.R + r.G) + 1) * (2 * (p.refDoppler + p.guardDoppler) + 1);
    const int64_t smax = ((cells * mag_max) >> r.div_sum) + 1;
    r.fast32 = smax < (int64_t(1) << 23) && ((smax * int64_t(r.scaler_raw)) << r.lin_shl) < (int64_t(1) << 31) ? 1 : 0;
  }
  return r;
}

// Twiddle ROM, built once per frame size: F32 = the per-pass base-twiddle tables of fft_lds.hpp, FIXED16 = W_N^k, k < N/2.
int get_rom(rsp_chain* c, int log2n, const void** out) {
  auto it = c->rom.find(log2n);
  if (it != c->rom.end()) {
    *out = it->second.d;
    return RSP_OK;
  }
  const int n = 1 << log2n, half = n / 2;
  TwiddleRom rom;
  if (c->p.dtype == RSP_DTYPE_F32 && log2n < rsp::kMinLog2N) {
    // 16..128-point frames (small.hip): plain W_N^k, k < N/2
    std::vector<float> h(2 * (size_t)half);
    for (int k = 0; k < half; ++k) {
      const double a = -2.0 * M_PI * (double)k / (double)n;
      h[2 * k] = (float)std::cos(a);
      h[2 * k + 1] = (float)std::sin(a);
    }
    HIP_TRY(hipMalloc(&rom.d, h.size() * sizeof(float)));
    HIP_TRY(hipMemcpy(rom.d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
  } else if (c->p.dtype == RSP_DTYPE_F32) {
    // per pass with LO > 0: entry `low` = {W^low, W^2low, W^4low, W^8low} of W = W_{2^(LO+W)}
    // (fft_lds.hpp load_tw), computed in double
    std::vector<float> h(2 * (size_t)rsp::tw_table_total(log2n));
    for (int p = 0; p < rsp::plan_np(log2n); ++p) {
      const int lo = rsp::plan_lo(log2n, p), w = rsp::plan_w(log2n, p);
      if (lo <= 0) continue;
      float* t = h.data() + 2 * (size_t)rsp::tw_table_offset(log2n, p);
      for (int low = 0; low < (1 << lo); ++low) {
        for (int j = 0; j < 4; ++j) {
          const long long e = (long long)low << j;  // exponent of W_{2^(lo+w)}; j >= w entries are never read
          const double a = -2.0 * M_PI * (double)e / (double)(1 << (lo + w));
          t[2 * (4 * low + j)] = (float)std::cos(a);
          t[2 * (4 * low + j) + 1] = (float)std::sin(a);
        }
      }
    }
    HIP_TRY(hipMalloc(&rom.d, h.size() * sizeof(float)));
    HIP_TRY(hipMemcpy(rom.d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
  } else {
    // Q2.14 (twiddleWidth 16, FftMagCfarChain.scala:80), round to nearest
    // N/2 words {wr, wi}, then the same twiddles as the operand pairs the packed butterflies read (fx_rom_entry)
    std::vector<uint32_t> h((size_t)half * 3);
    for (int k = 0; k < half; ++k) {
      const double a = -2.0 * M_PI * (double)k / (double)n;
      const int wr = (int)std::lround(std::cos(a) * 16384.0), wi = (int)std::lround(std::sin(a) * 16384.0);
      h[k] = ((uint32_t)(wr & 0xFFFF) << 16) | (uint32_t)(wi & 0xFFFF);
      const uint2 e = rsp::fx_rom_entry(h[k]);
      h[(size_t)half + 2 * k] = e.x;
      h[(size_t)half + 2 * k + 1] = e.y;
    }
    HIP_TRY(hipMalloc(&rom.d, h.size() * sizeof(uint32_t)));
    HIP_TRY(hipMemcpy(rom.d, h.data(), h.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
  }
  c->rom[log2n] = rom;
  *out = rom.d;
  return RSP_OK;
}

// Pre-FFT window table (no reference item; spec section 2.1): n coefficients, symmetric form
// w[i] = a0 - a1 cos(2 pi i / (n - 1)) + a2 cos(4 pi i / (n - 1)); fp32 for the F32 path, Q1.15 (x 32767,
// round to nearest) for FIXED16.
int get_window(rsp_chain* c, int type, int log2n, const void** out) {
  *out = nullptr;
  if (type == RSP_WINDOW_NONE) return RSP_OK;
  const int key = 64 * type + log2n;
  auto it = c->win.find(key);
  if (it != c->win.end()) {
    *out = it->second;
    return RSP_OK;
  }
  const int n = 1 << log2n;
  std::vector<double> w((size_t)n);
  for (int i = 0; i < n; ++i) {
    const double x = 2.0 * M_PI * (double)i / (double)(n - 1);
    w[i] = type == RSP_WINDOW_HANN ? 0.5 - 0.5 * std::cos(x)
         : type == RSP_WINDOW_HAMMING ? 0.54 - 0.46 * std::cos(x)
                                      : 0.42 - 0.5 * std::cos(x) + 0.08 * std::cos(2.0 * x);
  }
  void* d = nullptr;
  if (c->p.dtype == RSP_DTYPE_F32) {
    std::vector<float> h((size_t)n);
    for (int i = 0; i < n; ++i) h[i] = (float)w[i];
    HIP_TRY(hipMalloc(&d, h.size() * sizeof(float)));
    HIP_TRY(hipMemcpy(d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
  } else {
    std::vector<int16_t> h((size_t)n);
    for (int i = 0; i < n; ++i) h[i] = (int16_t)std::lround(w[i] * 32767.0);
    HIP_TRY(hipMalloc(&d, h.size() * sizeof(int16_t)));
    HIP_TRY(hipMemcpy(d, h.data(), h.size() * sizeof(int16_t), hipMemcpyHostToDevice));
  }
  c->win[key] = d;
  *out = d;
  return RSP_OK;
}

int ensure(void** ptr, size_t* have, size_t want) {
  if (*have >= want) return RSP_OK;
  if (*ptr) HIP_TRY(hipFree(*ptr));
  *ptr = nullptr;
  *have = 0;
  HIP_TRY(hipMalloc(ptr, want));
  *have = want;
  return RSP_OK;
}

size_t beat_bytes(const rsp_chain* c) { return c->p.dtype == RSP_DTYPE_F32 ? 8 : 4; }

int launch_rd(rsp_chain* c, const void* d_in, size_t n_ch, uint32_t* d_out, rsp_detection* d_list = nullptr,
              uint32_t cap = 0, uint32_t* d_found = nullptr);

// per-frame slot buffers of the fused 1-D list: (re)sized outside the hot path
int ensure_slots(rsp_chain* c, size_t n_frames) {
  if (n_frames <= c->fslots) return RSP_OK;
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (c->d_fcount) HIP_TRY(hipFree(c->d_fcount));
  if (c->d_fdet) HIP_TRY(hipFree(c->d_fdet));
  c->d_fcount = nullptr;
  c->d_fdet = nullptr;
  c->fslots = 0;
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->d_fcount), n_frames * sizeof(uint32_t)));
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->d_fdet), n_frames * rsp::kFrameDetCap * sizeof(uint2)));
  c->fslots = n_frames;
  return RSP_OK;
}

// The 1-D chain kernel over n_frames frames: dense words and / or the per-frame detection slots of frames
// slot0 .. slot0 + n_frames (slot0 < 0: none).  No compaction: callers that cut a batch into chunks compact once.
int launch_frames(rsp_chain* c, const void* d_in, size_t n_frames, uint32_t* d_out, int64_t slot0) {
  rsp::Chain1dLaunch a{};
  a.in = d_in;
  a.out = d_out;
  a.n_frames = (uint32_t)n_frames;
  a.log2n = (int)c->fft_stages;
  a.fixed = c->p.dtype == RSP_DTYPE_FIXED16;
  a.regs = snapshot(c);
  int rc = get_rom(c, a.log2n, &a.twiddles);
  if (rc != RSP_OK) return rc;
  rc = get_window(c, c->p.window, a.log2n, &a.regs.window);
  if (rc != RSP_OK) return rc;
  a.log_lut = c->d_log_lut;
  a.stream = c->stream;
  a.device = c->device;
  a.max_frames_per_launch = c->opt_max_frames;
  a.force_generic_tail = c->opt_generic_tail;
  a.experiment = c->opt_experiment;
  if (slot0 >= 0) {
    a.frame_count = c->d_fcount + slot0;
    a.frame_det = c->d_fdet + (size_t)slot0 * rsp::kFrameDetCap;
  }
  if (c->profiling) {  // events bound to the kernel dispatch itself (begin / end of the kernel, as rocprofv3 reports it)
    if (c->prof_used == c->prof_events.size()) {
      hipEvent_t e0, e1;
      HIP_TRY(hipEventCreate(&e0));
      HIP_TRY(hipEventCreate(&e1));
      c->prof_events.emplace_back(e0, e1);
    }
    a.ev_start = c->prof_events[c->prof_used].first;
    a.ev_stop = c->prof_events[c->prof_used].second;
    ++c->prof_used;
  }
  const hipError_t le = rsp::launch_chain1d(a);
  if (le == hipErrorOutOfMemory)
    return fail(RSP_ERR_UNSUPPORTED, "%d-point frames: this CFAR configuration needs more than the 160 KiB of LDS of a workgroup "
                "(16384 points do not hold the ordered statistic with indexLagg != indexLead, nor the FFT stage options)", 1 << a.log2n);
  HIP_TRY(le);
  return RSP_OK;
}

// per-frame slots of frames 0 .. n_frames -> list; on a failed launch the shared counters are re-zeroed so that
// the next call does not publish stale counts
int launch_list(rsp_chain* c, size_t n_frames, uint32_t* d_out, rsp_detection* d_list, uint32_t cap, uint32_t* d_found) {
  const hipError_t e = rsp::launch_compact_frames(c->d_fcount, c->d_fdet, (uint32_t)n_frames, d_out, (int)c->fft_stages,
                                                  c->p.cfarParams.sendCut ? 1 : 0, d_list, cap, c->d_ctr, d_found, c->stream);
  if (e != hipSuccess) {
    (void)hipMemsetAsync(c->d_ctr, 0, rsp::kCompactCounters * sizeof(uint32_t), c->stream);
    return fail(RSP_ERR_DEVICE, "detection compaction failed: %s", hipGetErrorString(e));
  }
  return RSP_OK;
}

int launch_dense(rsp_chain* c, const void* d_in, size_t n_frames, uint32_t* d_out,
                 rsp_detection* d_list = nullptr, uint32_t cap = 0, uint32_t* d_found = nullptr) {
  int rc = check_regs(c);
  if (rc != RSP_OK) return rc;
  if (c->p.dopplerPoints) return launch_rd(c, d_in, n_frames, d_out, d_list, cap, d_found);
  if (n_frames > 0x7fffffffull) return fail(RSP_ERR_INVALID, "n_frames = %zu too large for one call", n_frames);
  if (n_frames && (!d_in || (!d_out && !d_found))) return fail(RSP_ERR_INVALID, "NULL buffer");
  HIP_TRY(hipSetDevice(c->device));
  if (d_found) {
    rc = ensure_slots(c, n_frames);
    if (rc != RSP_OK) return rc;
  }
  rc = launch_frames(c, d_in, n_frames, d_out, d_found ? 0 : -1);
  if (rc != RSP_OK) return rc;
  if (d_found) return launch_list(c, n_frames, d_out, d_list, cap, d_found);
  return RSP_OK;
}

int launch_rd(rsp_chain* c, const void* d_in, size_t n_ch, uint32_t* d_out, rsp_detection* d_list, uint32_t cap,
              uint32_t* d_found) {
  if (n_ch == 0) {
    if (d_found) HIP_TRY(hipMemsetAsync(d_found, 0, 2 * sizeof(uint32_t), c->stream));
    return RSP_OK;
  }
  if (!d_in || !d_out) return fail(RSP_ERR_INVALID, "NULL buffer (the 2-D chain always writes its dense words)");
  HIP_TRY(hipSetDevice(c->device));
  const size_t cells = (n_ch * (size_t)c->p.dopplerPoints) << c->fft_stages;
  if (n_ch > 0xffffu || cells > 0x7fffffffull) return fail(RSP_ERR_INVALID, "2-D chain: %zu channels too many for one call", n_ch);
  // scratch for ONE chunk of channels (launch_rd2d): reused by every chunk, so it stays in the Infinity Cache
  const size_t chunk_ch = rsp::rd2d_chunk_channels((int)c->fft_stages, ilog2(c->p.dopplerPoints), (uint32_t)n_ch, c->opt_rd_chunk_bytes);
  const size_t chunk_cells = (chunk_ch * (size_t)c->p.dopplerPoints) << c->fft_stages;
  int rc = ensure(&c->d_x1, &c->d_x1_bytes, chunk_cells * 8);
  if (rc != RSP_OK) return rc;
  rc = ensure(reinterpret_cast<void**>(&c->d_mag2), &c->d_mag2_bytes, chunk_cells * 4);
  if (rc != RSP_OK) return rc;
  rsp::Rd2dLaunch a{};
  a.in = d_in;
  a.out = d_out;
  a.n_ch = (uint32_t)n_ch;
  a.log2nr = (int)c->fft_stages;
  a.log2nd = ilog2(c->p.dopplerPoints);
  a.regs = snapshot(c);
  a.ref_d = c->p.refDoppler;
  a.guard_d = c->p.guardDoppler;
  rc = get_rom(c, a.log2nr, &a.tw_range);
  if (rc != RSP_OK) return rc;
  rc = get_rom(c, a.log2nd, &a.tw_doppler);
  if (rc != RSP_OK) return rc;
  rc = get_window(c, c->p.window, a.log2nr, &a.regs.window);
  if (rc != RSP_OK) return rc;
  rc = get_window(c, c->p.windowDoppler, a.log2nd, &a.win_doppler);
  if (rc != RSP_OK) return rc;
  a.scratch_complex = c->d_x1;
  a.scratch_mag = c->d_mag2;
  a.stream = c->stream;
  a.device = c->device;
  a.force_tiled_cfar = c->opt_force_tiled;
  a.chunk_bytes = c->opt_rd_chunk_bytes;
  a.fixed = c->p.dtype != RSP_DTYPE_F32;
  a.log_lut = c->d_log_lut;
  if (d_found) {
    a.det_list = d_list;
    a.det_cap = cap;
    a.det_count = d_found;
  }
  hipEvent_t pe1 = nullptr;
  if (c->profiling) {  // the 2-D chain's three kernels as one bracket
    if (c->prof_used == c->prof_events.size()) {
      hipEvent_t e0, e1;
      HIP_TRY(hipEventCreate(&e0));
      HIP_TRY(hipEventCreate(&e1));
      c->prof_events.emplace_back(e0, e1);
    }
    HIP_TRY(hipEventRecord(c->prof_events[c->prof_used].first, c->stream));
    pe1 = c->prof_events[c->prof_used].second;
    ++c->prof_used;
  }
  HIP_TRY(rsp::launch_rd2d(a));
  if (pe1) HIP_TRY(hipEventRecord(pe1, c->stream));
  return RSP_OK;
}

// The host-buffer entry points as a three-stage pipeline over chunks of frames (host_pipe.hpp):
// H2D(k + 1) || kernels(k) || D2H(k - 1).  out_words = NULL: the dense words stay on the device (detection call).
// want_list: one compaction over the whole batch at the end, into c->d_list / c->d_count.
int host_process(rsp_chain* c, const void* in_beats, size_t n_frames, uint32_t* out_words, bool want_list, uint32_t cap) {
  const bool rd = c->p.dopplerPoints != 0;
  const size_t frame_cells = (size_t)(rd ? c->p.dopplerPoints : 1) << c->fft_stages;
  const size_t cells = n_frames * frame_cells;
  if (rd && cells > 0x7fffffffull) return fail(RSP_ERR_INVALID, "2-D chain: %zu channels too many for one call", n_frames);
  const size_t fin = frame_cells * beat_bytes(c), fout = frame_cells * sizeof(uint32_t) * (c->p.cfarParams.sendCut ? 2 : 1);
  int rc = ensure(&c->d_in, &c->d_in_bytes, n_frames * fin);
  if (rc != RSP_OK) return rc;
  rc = ensure(reinterpret_cast<void**>(&c->d_out), &c->d_out_bytes, n_frames * fout);
  if (rc != RSP_OK) return rc;
  if (want_list) {
    size_t list_bytes = c->d_list_cap * sizeof(rsp_detection);
    rc = ensure(reinterpret_cast<void**>(&c->d_list), &list_bytes, std::max<size_t>(cap, 1) * sizeof(rsp_detection));
    if (rc != RSP_OK) return rc;
    c->d_list_cap = list_bytes / sizeof(rsp_detection);
    if (!rd) {
      rc = ensure_slots(c, n_frames);
      if (rc != RSP_OK) return rc;
    }
  }
  rsp::HostPipe& hp = c->hp;
  HIP_TRY(hp.init());
  // chunk = ~16 MiB of input (whole frames; whole 64-frame workgroup rows when there are that many): long enough
  // for the link to run at its rate, short enough that filling and draining the pipeline stays a small part of it
  const size_t want_bytes = c->opt_host_chunk_bytes ? c->opt_host_chunk_bytes : (size_t(16) << 20);
  size_t cf = std::max<size_t>(1, want_bytes / fin);
  if (cf >= 64) cf &= ~size_t(63);
  cf = std::min(cf, n_frames);
  const size_t n_chunks = (n_frames + cf - 1) / cf;
  HIP_TRY(hp.events(n_chunks));
  const bool in_pinned = rsp::host_range_pinned(in_beats, n_frames * fin);
  const bool out_pinned = out_words && rsp::host_range_pinned(out_words, n_frames * fout);
  if (!in_pinned) HIP_TRY(hp.staging(true, cf * fin));
  if (out_words && !out_pinned) HIP_TRY(hp.staging(false, cf * fout));
  const char* hin = static_cast<const char*>(in_beats);
  char* hout = reinterpret_cast<char*>(out_words);
  char* din = static_cast<char*>(c->d_in);
  char* dout = reinterpret_cast<char*>(c->d_out);
  constexpr int kSlots = rsp::HostPipe::kSlots, kDrainLag = kSlots - 1;
  auto drain = [&](size_t j) -> hipError_t {  // staged output of chunk j: pinned ring -> the caller's buffer
    const size_t f0 = j * cf, nf = std::min(cf, n_frames - f0);
    hipError_t e = hipEventSynchronize(hp.ev_out[j]);
    if (e != hipSuccess) return e;
    hp.copier().copy(hout + f0 * fout, hp.stage_out[j % kSlots], nf * fout);
    return hipSuccess;
  };
  auto run = [&]() -> int {
    for (size_t k = 0; k < n_chunks; ++k) {
      const size_t f0 = k * cf, nf = std::min(cf, n_frames - f0);
      if (in_pinned) {
        HIP_TRY(hipMemcpyAsync(din + f0 * fin, hin + f0 * fin, nf * fin, hipMemcpyHostToDevice, hp.h2d));
      } else {
        void* st = hp.stage_in[k % kSlots];
        if (k >= (size_t)kSlots) HIP_TRY(hipEventSynchronize(hp.ev_in[k - kSlots]));  // the slot's previous chunk has left it
        hp.copier().copy(st, hin + f0 * fin, nf * fin);
        HIP_TRY(hipMemcpyAsync(din + f0 * fin, st, nf * fin, hipMemcpyHostToDevice, hp.h2d));
      }
      HIP_TRY(hipEventRecord(hp.ev_in[k], hp.h2d));
      HIP_TRY(hipStreamWaitEvent(c->stream, hp.ev_in[k], 0));
      uint32_t* dk = reinterpret_cast<uint32_t*>(dout + f0 * fout);
      const int r2 = rd ? launch_rd(c, din + f0 * fin, nf, dk, nullptr, 0, nullptr)
                        : launch_frames(c, din + f0 * fin, nf, dk, want_list ? (int64_t)f0 : -1);
      if (r2 != RSP_OK) return r2;
      if (out_words) {
        HIP_TRY(hipEventRecord(hp.ev_k[k], c->stream));
        HIP_TRY(hipStreamWaitEvent(hp.d2h, hp.ev_k[k], 0));
        if (out_pinned) {
          HIP_TRY(hipMemcpyAsync(hout + f0 * fout, dk, nf * fout, hipMemcpyDeviceToHost, hp.d2h));
        } else {
          // slot k % kSlots was drained kSlots - kDrainLag = 1 iteration ago (below)
          HIP_TRY(hipMemcpyAsync(hp.stage_out[k % kSlots], dk, nf * fout, hipMemcpyDeviceToHost, hp.d2h));
          HIP_TRY(hipEventRecord(hp.ev_out[k], hp.d2h));
          if (k >= (size_t)kDrainLag) HIP_TRY(drain(k - kDrainLag));
        }
      }
    }
    if (out_words && !out_pinned)
      for (size_t j = n_chunks > (size_t)kDrainLag ? n_chunks - kDrainLag : 0; j < n_chunks; ++j) HIP_TRY(drain(j));
    if (want_list) {
      if (rd) {
        HIP_TRY(rsp::launch_compact(c->d_out, cells, c->fft_stages, (uint32_t)ilog2(c->p.dopplerPoints), 0u, c->d_list, cap,
                                    c->d_ctr, c->d_count, c->stream));
      } else {
        const int r3 = launch_list(c, n_frames, c->d_out, c->d_list, cap, c->d_count);
        if (r3 != RSP_OK) return r3;
      }
    }
    return RSP_OK;
  };
  rc = run();
  // every stage is joined before the call returns, also on an error: the caller's buffers must not be in flight
  const hipError_t e0 = hipStreamSynchronize(hp.h2d), e1 = hipStreamSynchronize(c->stream), e2 = hipStreamSynchronize(hp.d2h);
  if (rc != RSP_OK) return rc;
  if (e0 != hipSuccess || e1 != hipSuccess || e2 != hipSuccess)
    return fail(RSP_ERR_DEVICE, "host pipeline: %s", hipGetErrorString(e0 != hipSuccess ? e0 : e1 != hipSuccess ? e1 : e2));
  return RSP_OK;
}

}  // namespace

extern "C" {

uint32_t rsp_abi_version(void) { return RSP_ABI_VERSION; }
const char* rsp_last_error(void) { return g_err; }

void rsp_chain_default_params(rsp_chain_params* p) {
  if (!p) return;
  std::memset(p, 0, sizeof(*p));
  // FftMagCfarChainVanillaApp: FftMagCfarChain.scala:77-116
  rsp_fft_params& f = p->fftParams;
  f.dataWidth = 16;
  f.twiddleWidth = 16;
  f.numPoints = 1024;
  f.useBitReverse = 1;
  f.runTime = 1;
  f.numAddPipes = 1;
  f.numMulPipes = 1;
  for (int s = 0; s < RSP_MAX_STAGES; ++s) {
    f.expandLogic[s] = 0;
    f.keepMSBorLSB[s] = 1;
  }
  f.minSRAMdepth = 1024;
  f.binPoint = 12;
  f.trimType = RSP_TRIM_CONVERGENT;
  rsp_mag_params& m = p->magParams;
  m.dataWidth = 16;
  m.binPoint = 12;
  m.dataWidthLog = 16;
  m.binPointLog = 9;
  m.log2LookUpWidth = 9;
  m.useLast = 1;
  m.numAddPipes = 1;
  m.numMulPipes = 1;
  rsp_cfar_params& c = p->cfarParams;
  c.protoIn = {16, 12};
  c.protoThreshold = {16, 12};
  c.protoScaler = {16, 12};
  c.leadLaggWindowSize = 64;
  c.guardWindowSize = 4;
  c.sendCut = 0;
  c.fftSize = 1024;
  c.minSubWindowSize = -1;
  c.includeCASH = 0;
  c.CFARAlgorithm = RSP_ALG_CA;
  c.numMulPipes = 1;
  c.edgeMode = RSP_EDGE_ZERO;
  p->fftAddress = {0x30000100u, 0xFFu};
  p->magAddress = {0x30000200u, 0xFFu};
  p->cfarAddress = {0x30002000u, 0xFFFu};
  p->beatBytes = 4;
  p->dtype = RSP_DTYPE_FIXED16;
  p->device = 0;
}

int rsp_chain_validate_params(const rsp_chain_params* p) { return validate(p); }

int rsp_chain_create(const rsp_chain_params* p, rsp_chain** out) {
  if (!out) return fail(RSP_ERR_INVALID, "out is NULL");
  *out = nullptr;
  int rc = validate(p);
  if (rc != RSP_OK) return rc;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(RSP_ERR_DEVICE, "no HIP device: this library has no CPU path");
  if (p->device < 0 || p->device >= ndev) return fail(RSP_ERR_DEVICE, "device %d of %d", p->device, ndev);
  rsp_chain* c = new (std::nothrow) rsp_chain();
  if (!c) return fail(RSP_ERR_NOMEM, "out of host memory");
  c->p = *p;
  c->device = p->device;
  reset_regs(c);
  auto bail = [&](int code) {
    rsp_chain_destroy(c);
    return code;
  };
  if (hipSetDevice(c->device) != hipSuccess) return bail(fail(RSP_ERR_DEVICE, "hipSetDevice(%d) failed", c->device));
  if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) return bail(fail(RSP_ERR_DEVICE, "hipStreamCreate failed"));
  c->stream = c->own_stream;
  if (hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) return bail(fail(RSP_ERR_DEVICE, "hipEventCreate failed"));
  if (hipMalloc(reinterpret_cast<void**>(&c->d_count), 2 * sizeof(uint32_t)) != hipSuccess ||
      hipMalloc(reinterpret_cast<void**>(&c->d_ctr), rsp::kCompactCounters * sizeof(uint32_t)) != hipSuccess ||
      hipMemset(c->d_ctr, 0, rsp::kCompactCounters * sizeof(uint32_t)) != hipSuccess)
    return bail(fail(RSP_ERR_DEVICE, "hipMalloc failed"));
  if (p->dtype == RSP_DTYPE_FIXED16) {
    // log2 fraction table of the logMagMux (MAGParams log2LookUpWidth / binPointLog,
    // FftMagCfarChain.scala:95-96); entry f = round(log2(1 + f / 2^lw) * 2^bpLog)
    const int lw = p->magParams.log2LookUpWidth;
    std::vector<int16_t> lut((size_t)1 << lw);
    for (size_t i = 0; i < lut.size(); ++i)
      lut[i] = (int16_t)std::llround(std::log2(1.0 + (double)i / (double)(1u << lw)) * (double)(1 << p->magParams.binPointLog));
    if (hipMalloc(reinterpret_cast<void**>(&c->d_log_lut), lut.size() * sizeof(int16_t)) != hipSuccess ||
        hipMemcpy(c->d_log_lut, lut.data(), lut.size() * sizeof(int16_t), hipMemcpyHostToDevice) != hipSuccess)
      return bail(fail(RSP_ERR_DEVICE, "log2 table upload failed"));
  }
  *out = c;
  return RSP_OK;
}

void rsp_chain_destroy(rsp_chain* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->own_stream) (void)hipStreamSynchronize(c->own_stream);
  for (auto& kv : c->rom)
    if (kv.second.d) (void)hipFree(kv.second.d);
  for (auto& kv : c->win)
    if (kv.second) (void)hipFree(kv.second);
  if (c->d_log_lut) (void)hipFree(c->d_log_lut);
  if (c->d_in) (void)hipFree(c->d_in);
  if (c->d_out) (void)hipFree(c->d_out);
  if (c->d_list) (void)hipFree(c->d_list);
  if (c->d_count) (void)hipFree(c->d_count);
  if (c->d_ctr) (void)hipFree(c->d_ctr);
  if (c->d_fcount) (void)hipFree(c->d_fcount);
  if (c->d_fdet) (void)hipFree(c->d_fdet);
  if (c->d_x1) (void)hipFree(c->d_x1);
  if (c->d_mag2) (void)hipFree(c->d_mag2);
  for (auto& pr : c->prof_events) {
    (void)hipEventDestroy(pr.first);
    (void)hipEventDestroy(pr.second);
  }
  c->hp.destroy();
  if (c->ev_switch) (void)hipEventDestroy(c->ev_switch);
  if (c->ev0) (void)hipEventDestroy(c->ev0);
  if (c->ev1) (void)hipEventDestroy(c->ev1);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  delete c;
}

static int decode(const rsp_chain* c, uint32_t addr, int* block, int* reg) {
  const rsp_address_set* as[3] = {&c->p.fftAddress, &c->p.magAddress, &c->p.cfarAddress};
  for (int b = 0; b < 3; ++b) {
    if ((addr & ~as[b]->mask) == as[b]->base) {
      const uint32_t off = addr & as[b]->mask;
      if (off % (uint32_t)c->p.beatBytes) return fail(RSP_ERR_ADDRESS, "unaligned register address 0x%08x", addr);
      *block = b;
      *reg = (int)(off / (uint32_t)c->p.beatBytes);
      return RSP_OK;
    }
  }
  return fail(RSP_ERR_ADDRESS, "address 0x%08x decodes to no block", addr);
}

int rsp_chain_write_reg(rsp_chain* c, uint32_t addr, uint32_t value) {
  if (!c) return fail(RSP_ERR_INVALID, "chain is NULL");
  int block, reg;
  int rc = decode(c, addr, &block, &reg);
  if (rc != RSP_OK) return rc;
  if (block == 0 && reg == 0) { c->fft_stages = value; return RSP_OK; }  // Tester:82
  if (block == 1 && reg == 0) { c->mag_mode = value; return RSP_OK; }    // Tester:84
  if (block == 2 && cfar_reg_present(c->p, reg)) { c->cfar[reg] = value; return RSP_OK; }
  return fail(RSP_ERR_ADDRESS, "address 0x%08x: no register at offset %d of block %d in this build", addr, reg * c->p.beatBytes, block);
}

int rsp_chain_read_reg(rsp_chain* c, uint32_t addr, uint32_t* value) {
  if (!c || !value) return fail(RSP_ERR_INVALID, "NULL argument");
  int block, reg;
  int rc = decode(c, addr, &block, &reg);
  if (rc != RSP_OK) return rc;
  if (block == 0 && reg == 0) { *value = c->fft_stages; return RSP_OK; }
  if (block == 1 && reg == 0) { *value = c->mag_mode; return RSP_OK; }
  if (block == 2 && cfar_reg_present(c->p, reg)) { *value = c->cfar[reg]; return RSP_OK; }
  return fail(RSP_ERR_ADDRESS, "address 0x%08x: no register there in this build", addr);
}

int rsp_chain_check_regs(rsp_chain* c) {
  if (!c) return fail(RSP_ERR_INVALID, "chain is NULL");
  return check_regs(c);
}

int rsp_chain_process_device(rsp_chain* c, const void* d_in, size_t n_frames, uint32_t* d_out) {
  if (!c) return fail(RSP_ERR_INVALID, "chain is NULL");
  return launch_dense(c, d_in, n_frames, d_out);
}

int rsp_chain_process_detect_device(rsp_chain* c, const void* d_in, size_t n_frames,
                                    uint32_t* d_out_words, rsp_detection* d_list, uint32_t cap,
                                    uint32_t* d_count) {
  if (!c) return fail(RSP_ERR_INVALID, "chain is NULL");
  if (!d_count || (cap && !d_list)) return fail(RSP_ERR_INVALID, "NULL buffer");
  if (n_frames == 0) {
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemsetAsync(d_count, 0, 2 * sizeof(uint32_t), c->stream));
    return RSP_OK;
  }
  return launch_dense(c, d_in, n_frames, d_out_words, d_list, cap, d_count);
}

int rsp_chain_process(rsp_chain* c, const void* in_beats, size_t n_frames, uint32_t* out_words) {
  if (!c) return fail(RSP_ERR_INVALID, "chain is NULL");
  int rc = check_regs(c);
  if (rc != RSP_OK) return rc;
  if (n_frames == 0) return RSP_OK;
  if (!in_beats || !out_words) return fail(RSP_ERR_INVALID, "NULL buffer");
  HIP_TRY(hipSetDevice(c->device));
  if (n_frames > (c->p.dopplerPoints ? 0xffffull : 0x7fffffffull))
    return fail(RSP_ERR_INVALID, "n_frames = %zu too large for one call", n_frames);
  return host_process(c, in_beats, n_frames, out_words, false, 0);
}

int rsp_chain_detections_device(rsp_chain* c, const uint32_t* d_out_words, size_t n_frames,
                                rsp_detection* d_list, uint32_t cap, uint32_t* d_count) {
  if (!c) return fail(RSP_ERR_INVALID, "chain is NULL");
  if (!d_count || (cap && !d_list) || (n_frames && !d_out_words)) return fail(RSP_ERR_INVALID, "NULL buffer");
  int rc = check_regs(c);  // the word layout (bin width, rows per frame) comes from the register file
  if (rc != RSP_OK) return rc;
  if (n_frames > 0x7fffffffull) return fail(RSP_ERR_INVALID, "n_frames = %zu too large for one call", n_frames);
  HIP_TRY(hipSetDevice(c->device));
  const uint32_t log2_rows = c->p.dopplerPoints ? (uint32_t)ilog2(c->p.dopplerPoints) : 0u;
  const uint64_t cells = ((uint64_t)n_frames << c->fft_stages) << log2_rows;
  HIP_TRY(rsp::launch_compact(d_out_words, cells, c->fft_stages, log2_rows, c->p.cfarParams.sendCut ? 1u : 0u, d_list, cap,
                              c->d_ctr, d_count, c->stream));
  return RSP_OK;
}

int rsp_chain_process_detections(rsp_chain* c, const void* in_beats, size_t n_frames,
                                 rsp_detection* list, size_t cap, size_t* n_found) {
  if (!c || !n_found) return fail(RSP_ERR_INVALID, "NULL argument");
  *n_found = 0;
  int rc = check_regs(c);
  if (rc != RSP_OK) return rc;
  if (n_frames == 0) return RSP_OK;
  if (!in_beats || (cap && !list)) return fail(RSP_ERR_INVALID, "NULL buffer");
  if (cap > 0xffffffffull) cap = 0xffffffffull;
  if (n_frames > (c->p.dopplerPoints ? 0xffffull : 0x7fffffffull))
    return fail(RSP_ERR_INVALID, "n_frames = %zu too large for one call", n_frames);
  HIP_TRY(hipSetDevice(c->device));
  // dense words are kept on the device: a frame with more than RSP_FRAME_DET_CAP peaks is completed
  // from them, so the host call never truncates a frame
  rc = host_process(c, in_beats, n_frames, nullptr, true, (uint32_t)cap);
  if (rc != RSP_OK) return rc;
  uint32_t counts[2] = {0, 0};  // {found, stored}
  HIP_TRY(hipMemcpy(counts, c->d_count, sizeof(counts), hipMemcpyDeviceToHost));
  const size_t stored = std::min<size_t>(counts[1], cap);
  if (stored) HIP_TRY(hipMemcpy(list, c->d_list, stored * sizeof(rsp_detection), hipMemcpyDeviceToHost));
  std::sort(list, list + stored, [](const rsp_detection& a, const rsp_detection& b) {
    if (a.frame != b.frame) return a.frame < b.frame;
    if (a.doppler != b.doppler) return a.doppler < b.doppler;
    return a.bin < b.bin;
  });
  *n_found = counts[0];
  return RSP_OK;
}

int rsp_chain_set_option(rsp_chain* c, int option, int64_t value) {
  if (!c) return fail(RSP_ERR_INVALID, "chain is NULL");
  switch (option) {
    case RSP_OPT_MAX_FRAMES_PER_LAUNCH:
      if (value < 0 || value > 0x7fffffffll) return fail(RSP_ERR_INVALID, "max frames per launch = %lld", (long long)value);
      c->opt_max_frames = (uint32_t)value;
      return RSP_OK;
    case RSP_OPT_FORCE_TILED_CFAR2D:
      c->opt_force_tiled = value != 0;
      return RSP_OK;
    case RSP_OPT_FORCE_GENERIC_TAIL:
      c->opt_generic_tail = value != 0;
      return RSP_OK;
    case RSP_OPT_EXPERIMENT:
      c->opt_experiment = value != 0;
      return RSP_OK;
    case RSP_OPT_RD_CHUNK_BYTES:
      if (value < 0) return fail(RSP_ERR_INVALID, "chunk bytes = %lld", (long long)value);
      c->opt_rd_chunk_bytes = (size_t)value;
      return RSP_OK;
    case RSP_OPT_HOST_CHUNK_BYTES:
      if (value < 0) return fail(RSP_ERR_INVALID, "chunk bytes = %lld", (long long)value);
      c->opt_host_chunk_bytes = (size_t)value;
      return RSP_OK;
    default:
      return fail(RSP_ERR_INVALID, "unknown option %d", option);
  }
}

int rsp_chain_set_stream(rsp_chain* c, void* hip_stream) {
  if (!c) return fail(RSP_ERR_INVALID, "chain is NULL");
  hipStream_t ns = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : c->own_stream;
  if (ns != c->stream) {
    // per-handle scratch (compaction counters, per-frame slots) assumes ONE stream in flight: what the old stream
    // still holds is ordered in front of the new one's work
    HIP_TRY(hipSetDevice(c->device));
    if (!c->ev_switch) HIP_TRY(hipEventCreateWithFlags(&c->ev_switch, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(c->ev_switch, c->stream));
    HIP_TRY(hipStreamWaitEvent(ns, c->ev_switch, 0));
  }
  c->stream = ns;
  return RSP_OK;
}

int rsp_chain_synchronize(rsp_chain* c) {
  if (!c) return fail(RSP_ERR_INVALID, "chain is NULL");
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return RSP_OK;
}

int rsp_chain_timer_start(rsp_chain* c) {
  if (!c) return fail(RSP_ERR_INVALID, "chain is NULL");
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipEventRecord(c->ev0, c->stream));
  return RSP_OK;
}

int rsp_chain_timer_stop(rsp_chain* c, float* elapsed_ms) {
  if (!c || !elapsed_ms) return fail(RSP_ERR_INVALID, "NULL argument");
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipEventRecord(c->ev1, c->stream));
  HIP_TRY(hipEventSynchronize(c->ev1));
  HIP_TRY(hipEventElapsedTime(elapsed_ms, c->ev0, c->ev1));
  return RSP_OK;
}

int rsp_chain_profile_enable(rsp_chain* c, int on) {
  if (!c) return fail(RSP_ERR_INVALID, "chain is NULL");
  c->profiling = on != 0;
  c->prof_used = 0;
  return RSP_OK;
}

int rsp_chain_profile_read(rsp_chain* c, float* total_ms, uint32_t* launches) {
  if (!c || !total_ms || !launches) return fail(RSP_ERR_INVALID, "NULL argument");
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipStreamSynchronize(c->stream));
  float sum = 0.f;
  for (size_t i = 0; i < c->prof_used; ++i) {
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, c->prof_events[i].first, c->prof_events[i].second));
    sum += ms;
  }
  *total_ms = sum;
  *launches = (uint32_t)c->prof_used;
  c->prof_used = 0;
  return RSP_OK;
}

int rsp_device_count(int* n) {
  if (!n) return fail(RSP_ERR_INVALID, "NULL argument");
  *n = 0;
  int k = 0;
  if (hipGetDeviceCount(&k) != hipSuccess) return fail(RSP_ERR_DEVICE, "hipGetDeviceCount failed (no HIP device?)");
  *n = k;
  return RSP_OK;
}

int rsp_device_malloc(int device, void** ptr, size_t bytes) {
  if (!ptr) return fail(RSP_ERR_INVALID, "NULL argument");
  HIP_TRY(hipSetDevice(device));
  HIP_TRY(hipMalloc(ptr, bytes));
  return RSP_OK;
}

int rsp_device_free(int device, void* ptr) {
  HIP_TRY(hipSetDevice(device));
  HIP_TRY(hipFree(ptr));
  return RSP_OK;
}

int rsp_host_alloc(int device, void** ptr, size_t bytes) {
  if (!ptr) return fail(RSP_ERR_INVALID, "NULL argument");
  *ptr = nullptr;
  HIP_TRY(hipSetDevice(device));
  HIP_TRY(hipHostMalloc(ptr, bytes ? bytes : 1, hipHostMallocDefault));
  return RSP_OK;
}

int rsp_host_free(void* ptr) {
  if (ptr) HIP_TRY(hipHostFree(ptr));
  return RSP_OK;
}

int rsp_host_register(int device, void* ptr, size_t bytes) {
  if (!ptr || !bytes) return fail(RSP_ERR_INVALID, "NULL / empty range");
  HIP_TRY(hipSetDevice(device));
  HIP_TRY(hipHostRegister(ptr, bytes, hipHostRegisterDefault));
  return RSP_OK;
}

int rsp_host_unregister(void* ptr) {
  if (!ptr) return fail(RSP_ERR_INVALID, "NULL argument");
  HIP_TRY(hipHostUnregister(ptr));
  return RSP_OK;
}

int rsp_memcpy_h2d(int device, void* dst, const void* src, size_t bytes) {
  HIP_TRY(hipSetDevice(device));
  HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
  return RSP_OK;
}

int rsp_memcpy_d2h(int device, void* dst, const void* src, size_t bytes) {
  HIP_TRY(hipSetDevice(device));
  HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
  return RSP_OK;
}

uint32_t rsp_pack_iq(int32_t re, int32_t im) {
  return ((uint32_t)(re & 0xFFFF) << 16) | (uint32_t)(im & 0xFFFF);
}

void rsp_unpack_word(uint32_t word, int32_t log2_fft_size, int32_t* threshold, uint32_t* bin,
                     uint32_t* peak) {
  if (threshold) *threshold = (int32_t)word >> (log2_fft_size + 1);  // signed Int shift: Tester:164
  if (bin) *bin = (word >> 1) & ((1u << log2_fft_size) - 1u);
  if (peak) *peak = word & 1u;
}

void rsp_unpack_word_f32(uint32_t word, float* threshold, uint32_t* peak) {
  if (threshold) {
    const uint32_t b = word & ~1u;
    std::memcpy(threshold, &b, sizeof(b));
  }
  if (peak) *peak = word & 1u;
}

}  // extern "C"
