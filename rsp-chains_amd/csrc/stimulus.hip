// PLFG -> NCO stimulus on the device (include/rspchain.h, rsp_stimulus_*).
// Replaces plfg/nco of RspChainVanilla (/root/reference/src/main/scala/RspChain.scala:41-42,57-58).
// The PLFG program is periodic, so the host expands ONE period into a prefix table of phase
// increments; sample n's phase is then a closed form (whole periods * period sum + prefix), and
// every sample is independent: one thread per sample, 4-byte coalesced stores, quarter-wave table
// in LDS.  Bound: HBM writes (4 B/sample).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/rspchain.h"

namespace {

thread_local char g_serr[256] = "";
#define SFAIL(code, ...) (snprintf(g_serr, sizeof(g_serr), __VA_ARGS__), (code))

__global__ void __launch_bounds__(256)
stimulus_kernel(uint32_t* __restrict__ beats, uint64_t n, const uint32_t* __restrict__ prefix /* period */,
                uint32_t period, uint32_t period_sum, const int32_t* __restrict__ quarter /* table_size + 1 */,
                uint32_t table_size, uint32_t phase_mask, int32_t lim) {
  extern __shared__ int32_t q[];
  for (uint32_t i = threadIdx.x; i <= table_size; i += 256) q[i] = quarter[i];
  __syncthreads();
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    // phase after the (i+1)-th increment: the accumulator steps BEFORE each output (index from 1)
    const uint64_t whole = i / period;
    const uint32_t rem = (uint32_t)(i - whole * period);
    const uint32_t phase = ((uint32_t)(whole * period_sum) + prefix[rem]) & phase_mask;
    auto sn = [&](uint32_t ph) {
      const uint32_t quad = ph / table_size, r = ph % table_size;
      const int32_t v = (quad & 1u) ? q[table_size - r] : q[r];
      return (quad & 2u) ? -v : v;
    };
    const int32_t s = min(sn(phase), lim), c = min(sn((phase + table_size) & phase_mask), lim);
    beats[i] = ((uint32_t)(c & 0xFFFF) << 16) | (uint32_t)(s & 0xFFFF);
  }
}

}  // namespace

struct rsp_stimulus {
  rsp_stimulus_params p;
  uint32_t regs[64];
  std::vector<uint32_t> ram;
  uint32_t* d_prefix = nullptr;
  size_t d_prefix_n = 0;
  int32_t* d_quarter = nullptr;
  uint32_t* d_beats = nullptr;
  size_t d_beats_n = 0;
};

namespace {
enum { kEnable = 0, kReset = 1, kFrames = 2, kChirps = 4, kStart = 5, kSegNums = 6 };

// one period of PLFG output values (oracle: orc_plfg)
std::vector<int32_t> plfg_period(const rsp_stimulus* s) {
  std::vector<int32_t> v;
  const int repOff = kSegNums + 4, ordOff = repOff + 8;  // RspChainVanillaTester.scala:80-82
  if (!s->regs[kEnable]) return {0};
  const int nch = (int)s->regs[kChirps];
  for (int ch = 0; ch < nch && ch < 8; ++ch) {
    const int o = (int)s->regs[ordOff + ch];
    for (uint32_t rep = 0; rep < s->regs[repOff + ch]; ++rep) {
      int32_t val = (int32_t)s->regs[kStart];
      for (uint32_t sg = 0; sg < s->regs[kSegNums + (o & 3)]; ++sg) {
        const size_t row = (size_t)o * s->p.plfgParams.maxNumOfSegments + sg;
        const uint32_t w = row < s->ram.size() ? s->ram[row] : 0u;
        int32_t slope = (int32_t)((w >> 8) & 0xFFFF);
        if (w & 2u) slope = -slope;
        if (w & 1u) val = (int32_t)s->regs[kStart];
        for (uint32_t k = 0; k < (w >> 24); ++k) {
          v.push_back(val);
          val += slope;
        }
      }
    }
  }
  if (v.empty()) v.push_back((int32_t)s->regs[kStart]);
  return v;
}
}  // namespace

extern "C" {

void rsp_stimulus_default_params(rsp_stimulus_params* p) {
  if (!p) return;
  std::memset(p, 0, sizeof(*p));
  p->plfgParams = {4, 8, 8, 4, 4, 8, 16, 0};
  p->ncoParams = {128, 16, 9, 0, 0, 0, 0, 1, 0, 0, 0};
  p->plfgAddress = {0x30000000u, 0xFFu};
  p->plfgRAM = {0x30001000u, 0xFFFu};
  p->ncoAddress = {0x30000300u, 0xFu};
  p->beatBytes = 4;
}

int rsp_stimulus_create(const rsp_stimulus_params* p, rsp_stimulus** out) {
  if (!p || !out) return SFAIL(RSP_ERR_INVALID, "NULL argument");
  *out = nullptr;
  const rsp_nco_params& n = p->ncoParams;
  if (n.tableSize < 4 || (n.tableSize & (n.tableSize - 1)) || (4 * n.tableSize) != (1 << n.phaseWidth))
    return SFAIL(RSP_ERR_UNSUPPORTED, "NCO: 4 * tableSize (%d) must equal 2^phaseWidth (%d)", n.tableSize, n.phaseWidth);
  if (n.tableWidth < 4 || n.tableWidth > 16) return SFAIL(RSP_ERR_INVALID, "NCO tableWidth %d", n.tableWidth);
  if (n.rasterizedMode || n.nInterpolationTerms || n.ditherEnable || !n.phaseAccEnable || n.roundingMode || n.pincType || n.poffType)
    return SFAIL(RSP_ERR_UNSUPPORTED, "NCO: only the reference's configuration (phase accumulator, streaming increment, RoundHalfUp, no dither/interpolation) is modelled");
  if (p->plfgParams.maxNumOfSegments < 1 || p->plfgParams.maxNumOfSegments > 8 || p->plfgParams.outputWidthFrac != 0)
    return SFAIL(RSP_ERR_UNSUPPORTED, "PLFG: maxNumOfSegments 1..8, outputWidthFrac 0");
  if (p->beatBytes != 4) return SFAIL(RSP_ERR_UNSUPPORTED, "beatBytes %d", p->beatBytes);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return SFAIL(RSP_ERR_DEVICE, "no HIP device: this library has no CPU path");
  rsp_stimulus* s = new (std::nothrow) rsp_stimulus();
  if (!s) return SFAIL(RSP_ERR_NOMEM, "out of memory");
  s->p = *p;
  std::memset(s->regs, 0, sizeof(s->regs));
  s->ram.assign(64, 0u);
  std::vector<int32_t> q((size_t)n.tableSize + 1);
  for (int k = 0; k <= n.tableSize; ++k)
    q[k] = (int32_t)std::floor(std::sin(2.0 * M_PI * k / (4.0 * n.tableSize)) * std::ldexp(1.0, n.tableWidth - 2) + 0.5);
  if (hipSetDevice(p->device) != hipSuccess || hipMalloc(reinterpret_cast<void**>(&s->d_quarter), q.size() * 4) != hipSuccess ||
      hipMemcpy(s->d_quarter, q.data(), q.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
    rsp_stimulus_destroy(s);
    return SFAIL(RSP_ERR_DEVICE, "device setup failed");
  }
  *out = s;
  return RSP_OK;
}

void rsp_stimulus_destroy(rsp_stimulus* s) {
  if (!s) return;
  (void)hipSetDevice(s->p.device);
  if (s->d_prefix) (void)hipFree(s->d_prefix);
  if (s->d_quarter) (void)hipFree(s->d_quarter);
  if (s->d_beats) (void)hipFree(s->d_beats);
  delete s;
}

static int sdecode(rsp_stimulus* s, uint32_t addr, uint32_t** slot) {
  const rsp_address_set& a = s->p.plfgAddress;
  const rsp_address_set& r = s->p.plfgRAM;
  if ((addr & ~a.mask) == a.base && (addr & a.mask) % 4 == 0 && (addr & a.mask) / 4 < 64) { *slot = &s->regs[(addr & a.mask) / 4]; return RSP_OK; }
  if ((addr & ~r.mask) == r.base && (addr & r.mask) % 4 == 0 && (addr & r.mask) / 4 < s->ram.size()) { *slot = &s->ram[(addr & r.mask) / 4]; return RSP_OK; }
  return SFAIL(RSP_ERR_ADDRESS, "address 0x%08x decodes to no PLFG register / RAM word (the NCO has none in this configuration)", addr);
}

int rsp_stimulus_write_reg(rsp_stimulus* s, uint32_t addr, uint32_t value) {
  if (!s) return SFAIL(RSP_ERR_INVALID, "NULL");
  uint32_t* slot;
  int rc = sdecode(s, addr, &slot);
  if (rc == RSP_OK) *slot = value;
  return rc;
}

int rsp_stimulus_read_reg(rsp_stimulus* s, uint32_t addr, uint32_t* value) {
  if (!s || !value) return SFAIL(RSP_ERR_INVALID, "NULL");
  uint32_t* slot;
  int rc = sdecode(s, addr, &slot);
  if (rc == RSP_OK) *value = *slot;
  return rc;
}

int rsp_stimulus_generate_device(rsp_stimulus* s, uint32_t* d_beats, size_t n, void* hip_stream) {
  if (!s || (n && !d_beats)) return SFAIL(RSP_ERR_INVALID, "NULL");
  if (n == 0) return RSP_OK;
  if (hipSetDevice(s->p.device) != hipSuccess) return SFAIL(RSP_ERR_DEVICE, "hipSetDevice failed");
  hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
  const std::vector<int32_t> v = plfg_period(s);
  std::vector<uint32_t> prefix(v.size());
  uint32_t acc = 0;
  for (size_t i = 0; i < v.size(); ++i) { acc += (uint32_t)v[i]; prefix[i] = acc; }
  if (s->d_prefix_n < prefix.size()) {
    if (s->d_prefix) (void)hipFree(s->d_prefix);
    if (hipMalloc(reinterpret_cast<void**>(&s->d_prefix), prefix.size() * 4) != hipSuccess) return SFAIL(RSP_ERR_DEVICE, "hipMalloc failed");
    s->d_prefix_n = prefix.size();
  }
  if (hipMemcpyAsync(s->d_prefix, prefix.data(), prefix.size() * 4, hipMemcpyHostToDevice, st) != hipSuccess ||
      hipStreamSynchronize(st) != hipSuccess)  // prefix is a host temporary
    return SFAIL(RSP_ERR_DEVICE, "prefix upload failed");
  const rsp_nco_params& nc = s->p.ncoParams;
  uint64_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(stimulus_kernel, dim3((uint32_t)blocks), dim3(256), (nc.tableSize + 1) * 4, st, d_beats, (uint64_t)n,
                     s->d_prefix, (uint32_t)prefix.size(), acc, s->d_quarter, (uint32_t)nc.tableSize,
                     (1u << nc.phaseWidth) - 1u, (1 << (nc.tableWidth - 1)) - 1);
  if (hipGetLastError() != hipSuccess) return SFAIL(RSP_ERR_DEVICE, "stimulus kernel launch failed");
  return RSP_OK;
}

int rsp_stimulus_generate(rsp_stimulus* s, uint32_t* beats, size_t n) {
  if (!s || (n && !beats)) return SFAIL(RSP_ERR_INVALID, "NULL");
  if (n == 0) return RSP_OK;
  if (hipSetDevice(s->p.device) != hipSuccess) return SFAIL(RSP_ERR_DEVICE, "hipSetDevice failed");
  if (s->d_beats_n < n) {
    if (s->d_beats) (void)hipFree(s->d_beats);
    s->d_beats = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&s->d_beats), n * 4) != hipSuccess) return SFAIL(RSP_ERR_DEVICE, "hipMalloc failed");
    s->d_beats_n = n;
  }
  int rc = rsp_stimulus_generate_device(s, s->d_beats, n, nullptr);
  if (rc != RSP_OK) return rc;
  if (hipMemcpy(beats, s->d_beats, n * 4, hipMemcpyDeviceToHost) != hipSuccess) return SFAIL(RSP_ERR_DEVICE, "copy back failed");
  return RSP_OK;
}

const char* rsp_stimulus_last_error(void) { return g_serr; }

}  // extern "C"
