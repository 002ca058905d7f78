// C++ counterpart of FftMagCfarChainVanillaSpec (src/test/scala/FftMagCfarChainTester.scala:195-249)
// driven through RspChain.hpp -> C ABI -> HIP.  Reads inputDataReal.txt / inputDataImag.txt in the
// tester's own dump format (%04x per line, Tester:56-68), writes outputData.txt / thresholdData.txt
// (Tester:155-175), so a dump pair produced by the real Chisel simulation elsewhere can be replayed here.
#include <cstdio>
#include <cstdlib>
#include <string>

#include "RspChain.hpp"

static std::vector<int> read_hex16(const std::string& path) {
  std::vector<int> v;
  FILE* f = std::fopen(path.c_str(), "r");
  if (!f) { std::perror(path.c_str()); std::exit(2); }
  unsigned x;
  while (std::fscanf(f, "%x", &x) == 1) v.push_back((int)(int16_t)(x & 0xFFFF));  // %04x of an Int: low 16 bits
  std::fclose(f);
  return v;
}

int main(int argc, char** argv) {
  using namespace rspChain;
  const std::string dir = argc > 1 ? argv[1] : ".";
  try {
    const std::vector<int> re = read_hex16(dir + "/inputDataReal.txt"), im = read_hex16(dir + "/inputDataImag.txt");
    if (re.size() != im.size() || re.empty()) { std::fprintf(stderr, "bad input dumps\n"); return 2; }
    RunTimeRspChainParams rt;  // all defaults, Tester:241
    rt.fftSize = (int)re.size();
    FftMagCfarVanillaParameters params;
    params.fftParams = FFTParams::fixed(16, 16, rt.fftSize);
    params.magParams = MAGParams::fixed();
    params.cfarParams = CFARParams({16, 12}, {16, 12}, {16, 12}, 64, 4, false, rt.fftSize);
    FftMagCfarChainVanilla dut(params);
    dut.configure(rt);
    std::vector<uint32_t> beats(re.size());
    for (size_t i = 0; i < re.size(); ++i) beats[i] = formAXI4StreamComplexData(re[i], im[i]);
    const std::vector<uint32_t> out = dut.stream(beats);
    const int fftBinWidth = log2Up(rt.fftSize);
    FILE* fo = std::fopen((dir + "/outputData.txt").c_str(), "w");
    FILE* ft = std::fopen((dir + "/thresholdData.txt").c_str(), "w");
    int peaks = 0;
    for (uint32_t w : out) {
      std::fprintf(fo, "%04x\n", w);
      std::fprintf(ft, "%04x\n", (unsigned)((int32_t)w >> (fftBinWidth + 1)));
      peaks += (int)(w & 1u);
    }
    std::fclose(fo);
    std::fclose(ft);
    std::printf("fftSize %d peaks %d\n", rt.fftSize, peaks);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  return 0;
}
