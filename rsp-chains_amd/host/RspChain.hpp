// C++ host-side mirror of the reference's chain interface (package rspChain), header-only,
// above the C ABI of include/rspchain.h.  The reference is Scala/Chisel and no JVM exists
// in this pipeline, so the compiled-language host layer is C++; names, argument meaning and
// failure behaviour follow the reference:
//   FFTParams::fixed / MAGParams::fixed / CFARParams      FftMagCfarChain.scala:78-112
//   FftMagCfarVanillaParameters                            FftMagCfarChain.scala:21-29
//   RunTimeRspChainParams (+ its requires)                 RspChainVanillaTester.scala:35-62
//   FftMagCfarChainVanilla: memWriteWord / stream          FftMagCfarChainTester.scala:82-151
// A Scala `require` failure is std::invalid_argument; an unimplemented feature is
// std::domain_error; a device error is std::runtime_error.
#pragma once
#include <cmath>
#include <cstdint>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/rspchain.h"

namespace rspChain {

inline int log2Up(int x) { int l = 0; while ((1 << l) < x) ++l; return l < 1 ? 1 : l; }
inline bool isPow2(int x) { return x > 0 && (x & (x - 1)) == 0; }
inline void require(bool c, const char* what = "") { if (!c) throw std::invalid_argument(std::string("requirement failed: ") + what); }

inline void check(int rc) {
  if (rc == RSP_OK) return;
  const std::string msg = rsp_last_error();
  if (rc == RSP_ERR_INVALID) throw std::invalid_argument("requirement failed: " + msg);
  if (rc == RSP_ERR_UNSUPPORTED) throw std::domain_error(msg);
  if (rc == RSP_ERR_ADDRESS) throw std::out_of_range(msg);
  throw std::runtime_error(msg);
}

enum CFARAlgorithmType { CACFARType = RSP_ALG_CA, GOSCFARType = RSP_ALG_GOS, GOSCACFARType = RSP_ALG_GOSCA };

struct FixedPoint { int width, binaryPoint; };
struct AddressSet { uint32_t base, mask; };

struct FFTParams {
  rsp_fft_params c{};
  // FFTParams.fixed(dataWidth, twiddleWidth, numPoints, useBitReverse, runTime, numAddPipes,
  //                 numMulPipes, expandLogic, keepMSBorLSB, minSRAMdepth, binPoint)
  static FFTParams fixed(int dataWidth = 16, int twiddleWidth = 16, int numPoints = 1024, bool useBitReverse = true,
                         bool runTime = true, int numAddPipes = 1, int numMulPipes = 1,
                         std::vector<int> expandLogic = {}, std::vector<bool> keepMSBorLSB = {},
                         int minSRAMdepth = 1024, int binPoint = 12) {
    FFTParams p;
    const int stages = log2Up(numPoints);
    if (expandLogic.empty()) expandLogic.assign(stages, 0);
    if (keepMSBorLSB.empty()) keepMSBorLSB.assign(stages, true);
    require((int)expandLogic.size() == stages && (int)keepMSBorLSB.size() == stages && stages <= RSP_MAX_STAGES,
            "expandLogic/keepMSBorLSB need one entry per stage");
    p.c.dataWidth = dataWidth; p.c.twiddleWidth = twiddleWidth; p.c.numPoints = numPoints;
    p.c.useBitReverse = useBitReverse; p.c.runTime = runTime; p.c.numAddPipes = numAddPipes;
    p.c.numMulPipes = numMulPipes; p.c.minSRAMdepth = minSRAMdepth; p.c.binPoint = binPoint;
    p.c.trimType = RSP_TRIM_CONVERGENT;
    for (int s = 0; s < RSP_MAX_STAGES; ++s) {
      p.c.expandLogic[s] = s < stages ? expandLogic[s] : 0;
      p.c.keepMSBorLSB[s] = s < stages ? (int)keepMSBorLSB[s] : 1;
    }
    return p;
  }
};

struct MAGParams {
  rsp_mag_params c{};
  static MAGParams fixed(int dataWidth = 16, int binPoint = 12, int dataWidthLog = 16, int binPointLog = 9,
                         int log2LookUpWidth = 9, bool useLast = true, int numAddPipes = 1, int numMulPipes = 1) {
    MAGParams p;
    p.c = {dataWidth, binPoint, dataWidthLog, binPointLog, log2LookUpWidth, (int)useLast, numAddPipes, numMulPipes};
    return p;
  }
};

struct CFARParams {
  rsp_cfar_params c{};
  CFARParams(FixedPoint protoIn = {16, 12}, FixedPoint protoThreshold = {16, 12}, FixedPoint protoScaler = {16, 12},
             int leadLaggWindowSize = 64, int guardWindowSize = 4, bool sendCut = false, int fftSize = 1024,
             std::optional<int> minSubWindowSize = std::nullopt, bool includeCASH = false,
             CFARAlgorithmType CFARAlgorithm = CACFARType, int numMulPipes = 1) {
    c.protoIn = {protoIn.width, protoIn.binaryPoint};
    c.protoThreshold = {protoThreshold.width, protoThreshold.binaryPoint};
    c.protoScaler = {protoScaler.width, protoScaler.binaryPoint};
    c.leadLaggWindowSize = leadLaggWindowSize; c.guardWindowSize = guardWindowSize; c.sendCut = sendCut;
    c.fftSize = fftSize; c.minSubWindowSize = minSubWindowSize.value_or(-1); c.includeCASH = includeCASH;
    c.CFARAlgorithm = CFARAlgorithm; c.numMulPipes = numMulPipes; c.edgeMode = RSP_EDGE_ZERO;
  }
};

struct FftMagCfarVanillaParameters {
  FFTParams fftParams; MAGParams magParams; CFARParams cfarParams;
  AddressSet fftAddress{0x30000100u, 0xFFu}, magAddress{0x30000200u, 0xFFu}, cfarAddress{0x30002000u, 0xFFFu};
  int beatBytes = 4;
  int dtype = RSP_DTYPE_FIXED16, device = 0;  // GPU extensions
  rsp_chain_params to_c() const {
    rsp_chain_params p{};
    p.fftParams = fftParams.c; p.magParams = magParams.c; p.cfarParams = cfarParams.c;
    p.fftAddress = {fftAddress.base, fftAddress.mask}; p.magAddress = {magAddress.base, magAddress.mask};
    p.cfarAddress = {cfarAddress.base, cfarAddress.mask};
    p.beatBytes = beatBytes; p.dtype = dtype; p.device = device;
    return p;
  }
};

struct RunTimeRspChainParams {  // RspChainVanillaTester.scala:35-48
  std::optional<std::string> CFARAlgorithm = "CA";
  std::string CFARMode = "Greatest Of";
  int refWindowSize = 32, guardWindowSize = 4;
  std::optional<int> subWindowSize;
  int fftSize = 1024;
  double thresholdScaler = 3.5;
  std::optional<int> divSum = 5;
  int peakGrouping = 0;
  std::optional<int> indexLagg, indexLead;
  int magMode = 2, logOrLinearMode = 1;
  void validate() const {  // :50-61
    require(isPow2(refWindowSize) && isPow2(fftSize));
    require(refWindowSize > 0 && guardWindowSize > 0);
    require(refWindowSize > guardWindowSize);
    if (subWindowSize) require(*subWindowSize < refWindowSize);
    if (indexLead) require(*indexLead < refWindowSize);
    if (indexLagg) require(*indexLagg < refWindowSize);
  }
};

class FftMagCfarChainVanilla {
 public:
  explicit FftMagCfarChainVanilla(const FftMagCfarVanillaParameters& params) : params_(params) {
    const rsp_chain_params c = params.to_c();
    check(rsp_chain_create(&c, &h_));
  }
  ~FftMagCfarChainVanilla() { rsp_chain_destroy(h_); }
  FftMagCfarChainVanilla(const FftMagCfarChainVanilla&) = delete;
  FftMagCfarChainVanilla& operator=(const FftMagCfarChainVanilla&) = delete;

  void memWriteWord(uint32_t addr, uint32_t value) { check(rsp_chain_write_reg(h_, addr, value)); }
  uint32_t memReadWord(uint32_t addr) { uint32_t v = 0; check(rsp_chain_read_reg(h_, addr, &v)); return v; }

  // CSR sequence of FftMagCfarChainTester.scala:82-132
  void configure(const RunTimeRspChainParams& rt) {
    rt.validate();
    const auto& p = params_;
    const uint32_t bb = (uint32_t)p.beatBytes, base = p.cfarAddress.base;
    memWriteWord(p.fftAddress.base, (uint32_t)log2Up(rt.fftSize));
    memWriteWord(p.magAddress.base, (uint32_t)rt.magMode);
    const int bpThr = p.cfarParams.c.protoThreshold.binaryPoint;
    memWriteWord(base, (uint32_t)rt.fftSize);
    memWriteWord(base + bb, (uint32_t)(int)(rt.thresholdScaler * std::pow(2.0, bpThr)));
    memWriteWord(base + 2 * bb, (uint32_t)rt.logOrLinearMode);
    const int alg = p.cfarParams.c.CFARAlgorithm;
    if (alg != GOSCFARType) { require(rt.divSum.has_value(), "divSum"); memWriteWord(base + 3 * bb, (uint32_t)*rt.divSum); }
    memWriteWord(base + 4 * bb, (uint32_t)rt.peakGrouping);
    if (alg == GOSCACFARType) {
      require(rt.CFARAlgorithm.has_value(), "CFARAlgorithm");
      memWriteWord(base + 5 * bb, *rt.CFARAlgorithm == "GOS" ? 1u : 0u);
    }
    uint32_t mode = 0;
    if (rt.CFARMode == "Greatest Of") mode = 1; else if (rt.CFARMode == "Smallest Of") mode = 2; else if (rt.CFARMode == "CASH") mode = 3;
    memWriteWord(base + 6 * bb, mode);
    memWriteWord(base + 7 * bb, (uint32_t)rt.refWindowSize);
    memWriteWord(base + 8 * bb, (uint32_t)rt.guardWindowSize);
    if (alg != CACFARType) {
      require(rt.indexLagg && rt.indexLead, "indexLagg/indexLead");
      memWriteWord(base + 9 * bb, (uint32_t)*rt.indexLagg);
      memWriteWord(base + 10 * bb, (uint32_t)*rt.indexLead);
    }
    if (alg == CACFARType && p.cfarParams.c.includeCASH) {
      require(rt.subWindowSize.has_value(), "subWindowSize");
      memWriteWord(base + 11 * bb, (uint32_t)*rt.subWindowSize);
    }
  }

  // stream whole frames in (TLAST closes each), collect fftSize words per frame
  std::vector<uint32_t> stream(const std::vector<uint32_t>& beats) {
    const size_t n = (size_t)1 << memReadWord(params_.fftAddress.base);
    require(beats.size() % n == 0, "beats must be whole frames");
    std::vector<uint32_t> out(beats.size() * (params_.cfarParams.c.sendCut ? 2 : 1));  // sendCut: {word, cut} per cell
    check(rsp_chain_process(h_, beats.data(), beats.size() / n, out.data()));
    return out;
  }
  rsp_chain* handle() { return h_; }

 private:
  FftMagCfarVanillaParameters params_;
  rsp_chain* h_ = nullptr;
};

// RspChainTesterUtils.scala:105-109
inline uint32_t formAXI4StreamComplexData(int re, int im) { return rsp_pack_iq(re, im); }

}  // namespace rspChain
