// chain1d_wave_kernel: the F32 4096-point chain with ONE WAVE PER FRAME.
// STATUS: alternative formulation, opt-in with RSP_CHAIN_WAVE=1, parity-tested (tests/test_gpu_wave.py),
// currently ~10 % SLOWER than chain1d_kernel -- kept because it removes every barrier and both
// overlap problems of that kernel and its remaining cost is plain instruction count.
//
// Same path and same register semantics as chain1d_kernel (chain1d.hip): sdf-fft -> logMagMux ->
// CFAR of FftMagCfarChainVanilla (/root/reference/src/main/scala/FftMagCfarChain.scala:31-73),
// cell-averaging family (CA / GO / SO), for the headline shape (BASELINE.json configs[1]).
//
// Why a second formulation: the workgroup-per-frame kernel spends its time in phases that do not
// overlap (HBM latency, 6 barriers, LDS exchanges) with 4 waves per SIMD to hide them.  Here a
// frame never leaves one wave, so there is NO barrier at all, and the memory stream is decoupled
// from the arithmetic by construction:
//   * 64 samples per lane (sample i = 64 j + lane in register j): the 4096-point DIF FFT is two
//     64-point DFTs entirely in registers (radix-2 stages with compile-time W64 twiddles) around
//     ONE 64 x 64 transpose through wave-private LDS -- one exchange instead of two;
//   * the wave is persistent (one 4-wave workgroup per CU = one wave per SIMD; the waves share only
//     the per-lane twiddle table): the loads of its NEXT frame are issued in four bursts of 16
//     spread over the FFT of the current one, into a second register set (the wave owns all 512
//     registers of its SIMD lane: the spare set lands in AGPRs), so that loads and the CFAR stage's
//     word stores never crowd the 6-bit vmcnt counter (63 operations in flight per wave) together;
//   * bin k = lane + 64 p sits in register bitrev6(p): magnitudes, CFAR reads and the dense store
//     are all lane-contiguous with compile-time offsets;
//   * prefix sums are relative to 64-cell blocks (one block per lane: no cross-lane scan); a window
//     sum is pb[v] - pb[u] (+ the total of u's block when the window crosses a block edge), which
//     needs refWindow + guardWindow + 1 <= 64 -- larger windows, CASH and GOS stay on chain1d_kernel.
#include <hip/hip_runtime.h>
#include <float.h>
#include <stdlib.h>

#include "chain_regs.hpp"
#include "fft_lds.hpp"
#include "kernels.hpp"

namespace rsp {
namespace {

constexpr int kWvM = 12, kWvN = 1 << kWvM;
constexpr int kWvHalo = 64;  // one 64-cell block either side of the frame

// cos / sin of 2 pi k / 64
constexpr float kC64[32] = {
    1.f, 0.99518472667219693f, 0.98078528040323043f, 0.95694033573220882f, 0.92387953251128674f,
    0.88192126434835505f, 0.83146961230254524f, 0.77301045336273699f, 0.70710678118654757f,
    0.63439328416364549f, 0.55557023301960229f, 0.47139673682599781f, 0.38268343236508984f,
    0.29028467725446233f, 0.19509032201612833f, 0.09801714032956077f, 0.0f, -0.098017140329560645f,
    -0.19509032201612819f, -0.29028467725446216f, -0.38268343236508973f, -0.4713967368259977f,
    -0.55557023301960196f, -0.63439328416364538f, -0.70710678118654746f, -0.77301045336273699f,
    -0.83146961230254535f, -0.88192126434835494f, -0.92387953251128674f, -0.95694033573220882f,
    -0.98078528040323043f, -0.99518472667219682f};
constexpr float kS64[32] = {
    0.0f, 0.098017140329560604f, 0.19509032201612825f, 0.29028467725446233f, 0.38268343236508978f,
    0.47139673682599764f, 0.55557023301960218f, 0.63439328416364549f, 0.70710678118654746f,
    0.77301045336273699f, 0.83146961230254524f, 0.88192126434835494f, 0.92387953251128674f,
    0.95694033573220894f, 0.98078528040323043f, 0.99518472667219682f, 1.f, 0.99518472667219693f,
    0.98078528040323043f, 0.95694033573220894f, 0.92387953251128674f, 0.88192126434835505f,
    0.83146961230254546f, 0.7730104533627371f, 0.70710678118654757f, 0.63439328416364549f,
    0.55557023301960218f, 0.47139673682599786f, 0.38268343236508989f, 0.29028467725446239f,
    0.19509032201612861f, 0.098017140329560826f};

// d * exp(-2 pi i k / 64); k is a compile-time constant after unrolling
__device__ __forceinline__ f32x2 mul_w64(f32x2 d, int k) {
  if (k == 0) return d;
  const f32x2 dyx = {d.y, d.x};
  // -i: one packed multiply (the swap is an op_sel modifier); everything else, the W8 family
  // included, is the generic multiply + fma: 2 packed ops
  if (k == 16) return dyx * f32x2{1.0f, -1.0f};
  const f32x2 ss = {kS64[k], -kS64[k]}, cc = {kC64[k], kC64[k]};
  return __builtin_elementwise_fma(dyx, ss, d * cc);
}

// 64-point DIF DFT in registers; output q lands in register bitrev6(q).  Butterflies are written
// four at a time, phase by phase (add/sub, multiply, fma): with one wave per SIMD nothing else fills
// the wait states between dependent packed ops, so independent work has to sit next to each other.
__device__ __forceinline__ void dft64(f32x2 (&x)[64]) {
#pragma unroll
  for (int st = 0; st < 6; ++st) {
    const int bl = 5 - st;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      f32x2 d[4];
      int r1s[4], ks[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = g * 4 + u;                                  // butterfly number 0..31
        const int r0 = ((i >> bl) << (bl + 1)) | (i & ((1 << bl) - 1));
        const int r1 = r0 | (1 << bl);
        r1s[u] = r1;
        ks[u] = (r0 & ((1 << bl) - 1)) << (5 - bl);
        const f32x2 a = x[r0], b = x[r1];
        x[r0] = a + b;
        d[u] = a - b;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) x[r1s[u]] = mul_w64(d[u], ks[u]);
    }
  }
}

// wave-private LDS: the transpose image (64 rows x 66 slots of 8 B), later overlaid by the
// magnitude and prefix arrays: cell x (offset by the halo) at slot x + 4 (x >> 6) -- a lane's 64-cell run
// starts 16-byte aligned and 4 banks after its neighbour's, so ds_read/write_b128 are conflict-free
constexpr int kWvPitch = 68;                                   // slots per 64-cell block
constexpr int kWvXchPitch = 66;                                // transpose image: 8-byte slots per row
constexpr int kWvArrSlots = ((kWvN + 2 * kWvHalo) >> 6) * kWvPitch + 8;  // 66 blocks
constexpr int kWvMagOff = 0;
constexpr int kWvPbOff = kWvArrSlots * 4;
constexpr int kWvBsOff = 2 * kWvArrSlots * 4;              // block totals, blocks -1 .. 64
constexpr int kWvDetOff = kWvBsOff + 68 * 4;               // count + kFrameDetCap {bin, word}
constexpr int kWvLdsBytes = ((kWvDetOff + 8 + kFrameDetCap * 8 + 63) / 64) * 64;  // per wave
constexpr int kWvTwBytes = 16 * 64 * 8;  // shared: W_4096^(lane b), W_4096^(8 lane b), b < 8
constexpr int kWvWaves = 4;
#ifndef RSP_WAVE_BATCH
#define RSP_WAVE_BATCH 2
#endif
static_assert(64 * kWvXchPitch * 8 <= kWvBsOff, "transpose image fits under the block totals");

__device__ __forceinline__ void lds_order() { asm volatile("" ::: "memory"); }

#ifdef RSP_WAVE_PROF
#define PROF_DECL unsigned long long prof_t = clock64(), prof_acc[10] = {0}
#define PROF(i) do { __builtin_amdgcn_s_waitcnt(0xc07f); const unsigned long long t_ = clock64(); prof_acc[i] += t_ - prof_t; prof_t = t_; } while (0)
#define PROF_VM(i) do { __builtin_amdgcn_s_waitcnt(0x0070); const unsigned long long t_ = clock64(); prof_acc[i] += t_ - prof_t; prof_t = t_; } while (0)
#else
#define PROF_DECL
#define PROF(i)
#define PROF_VM(i)
#endif

__device__ __forceinline__ float mag_f32w(f32x2 z, int mode) {
  const float ar = fabsf(z.x), ai = fabsf(z.y);
  const float u = fmaxf(ar, ai), v = fminf(ar, ai);
  const float jpl = fmaxf(u + v * 0.125f, u * 0.875f + v * 0.5f);  // RspChainTesterUtils.scala:120-127
  if (mode == 2) return jpl;
  if (mode == 0) return z.x * z.x + z.y * z.y;
  return __log2f(fmaxf(jpl, FLT_MIN));
}

__global__ void __launch_bounds__(64 * kWvWaves)
chain1d_wave_kernel(const void* __restrict__ in, uint32_t* __restrict__ out, uint32_t n_frames, ChainRegs rg,
                    uint32_t* __restrict__ fcount, uint2* __restrict__ fdet, uint32_t* __restrict__ zero_a,
                    uint32_t* __restrict__ zero_b) {
  constexpr int N = kWvN;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (zero_a) *zero_a = 0u;
    if (zero_b) *zero_b = 0u;
  }
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // per-lane twiddles of the pass between the two DFTs: W_4096^(lane q), q = 8 a + b, as the product
  // of tw[b][lane] = W^(lane b) and tw[8 + a][lane] = W^(8 lane a); built once, shared by the waves
  f32x2* tw = reinterpret_cast<f32x2*>(smem_all);
  for (int e = wv; e < 16; e += kWvWaves) {
    float sn, cs;
    const int m = e < 8 ? lane * e : lane * 8 * (e - 8);
    sincospif(-2.0f * (float)m / (float)N, &sn, &cs);
    tw[e * 64 + lane] = f32x2{cs, sn};
  }
  __syncthreads();  // the only barrier: from here on the waves are independent
  unsigned char* smem = smem_all + kWvTwBytes + wv * kWvLdsBytes;
  f32x2* xch = reinterpret_cast<f32x2*>(smem);
  float* mag = reinterpret_cast<float*>(smem + kWvMagOff);
  float* pb = reinterpret_cast<float*>(smem + kWvPbOff);
  float* bs = reinterpret_cast<float*>(smem + kWvBsOff);  // bs[b + 1] = total of block b
  uint32_t* det_cnt = reinterpret_cast<uint32_t*>(smem + kWvDetOff);
  uint2* det_stage = reinterpret_cast<uint2*>(smem + kWvDetOff + 8);

  // CFAR geometry (FftMagCfarChain.scala:105-106): lagging cells [k-G-R, k-G), leading cells
  // [k+G+1, k+G+R+1); cell k = lane + 64 p at slot base(off) + 68 p
  const int R = rg.R, G = rg.G;
  auto slot_base = [&](int off) { const int x = lane + off + kWvHalo; return x + 4 * (x >> 6); };
  auto block_base = [&](int off) { return ((lane + off + kWvHalo) >> 6); };  // bs index of the cell's block
  const float* pu0 = pb + slot_base(-G - R);
  const float* pv0 = pb + slot_base(-G);
  const float* pu1 = pb + slot_base(G + 1);
  const float* pv1 = pb + slot_base(G + R + 1);
  const float* bu0 = bs + block_base(-G - R);
  const float* bu1 = bs + block_base(G + 1);
  const float c0 = block_base(-G) != block_base(-G - R) ? 1.0f : 0.0f;
  const float c1 = block_base(G + R + 1) != block_base(G + 1) ? 1.0f : 0.0f;
  const float* pml = mag + slot_base(-1);
  const float* pmr = mag + slot_base(1);
  float* pm = mag + slot_base(0);
  // F32 threshold = comb * kA + kB: (div * scaler, 0) linear, (div, scaler) log domain
  const float kA = rg.linear ? rg.div_f * rg.scaler_f : rg.div_f, kB = rg.linear ? 0.0f : rg.scaler_f;
  const bool wrap = rg.edge != 0;
  const float scale = 1.0f / (float)N;  // net 1/N: FftMagCfarChainTester.scala:77

  // one cell in two halves, so that the sweep can issue the LDS reads of the next cells before it
  // does the arithmetic of the current ones (with one wave per SIMD nothing else hides LDS latency).
  // o = 68 p (slot offset), p = k >> 6.  The rare detection path reuses both: same bits.
  struct CellIn { float cut, ml, mr, pv0, pu0, pv1, pu1, b0, b1; };
  auto fetch = [&](auto group_c, int o, int p) -> CellIn {
    constexpr bool GROUP = decltype(group_c)::value;
    CellIn c;
    c.cut = pm[o];
    c.ml = GROUP ? pml[o] : 0.0f;
    c.mr = GROUP ? pmr[o] : 0.0f;
    c.pv0 = pv0[o]; c.pu0 = pu0[o]; c.pv1 = pv1[o]; c.pu1 = pu1[o];
    c.b0 = bu0[p]; c.b1 = bu1[p];
    return c;
  };
  auto word_of = [&](auto mode_c, auto group_c, const CellIn& c) -> uint32_t {
    constexpr int MODE = decltype(mode_c)::value;
    constexpr bool GROUP = decltype(group_c)::value;
    bool group_ok = true;
    if constexpr (GROUP) group_ok = (c.cut > c.ml) & (c.cut > c.mr);
    // scalar on purpose: the four prefix values arrive in unrelated registers, packing them costs
    // more moves than the packed subtract / fma would save
    const float sw0 = __fmaf_rn(c0, c.b0, c.pv0 - c.pu0);
    const float sw1 = __fmaf_rn(c1, c.b1, c.pv1 - c.pu1);
    float comb;
    if constexpr (MODE == 0) comb = sw0 + sw1;
    else if constexpr (MODE == 1) comb = fmaxf(sw0, sw1);
    else comb = fminf(sw0, sw1);
    const float thr = __fmaf_rn(comb, MODE == 0 ? kA * 0.5f : kA, kB);
    const uint32_t peak = (uint32_t)((c.cut > thr) & group_ok);
    return (__float_as_uint(thr) & ~1u) | peak;
  };

  // buffer addressing: frame and register offsets live in SGPRs / the immediate field, the lane
  // offset is loop-invariant; a NULL `out` gets a zero-length buffer (every store dropped) so the
  // sweep stays branch-free.  One launch covers < 4 GiB of input (launch_chain1d splits).
  constexpr int kRsrc3 = 0x00020000;
  const __amdgpu_buffer_rsrc_t rin =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(in), 0, (int)(n_frames * (uint32_t)(N * 8)), kRsrc3);
  const __amdgpu_buffer_rsrc_t rout =
      __builtin_amdgcn_make_buffer_rsrc(out, 0, out ? (int)(n_frames * (uint32_t)(N * 4)) : 0, kRsrc3);
  const uint32_t stride = gridDim.x * kWvWaves;
  f32x2 x[64], y[64];  // y = the frame in flight
  auto load_rows = [&](uint32_t f, int j0) {
#pragma unroll
    for (int j = j0; j < j0 + 16; ++j)
      y[j] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rin, lane * 8, f * (uint32_t)(N * 8) + j * 512, 0));
  };
  uint32_t f = blockIdx.x * kWvWaves + wv;
  if (f < n_frames) { load_rows(f, 0); load_rows(f, 16); load_rows(f, 32); load_rows(f, 48); }
  // 64 stores to nowhere (zero-length buffer): in the loop the next frame's loads are followed by the
  // 64 word stores, and the compiler's wait-count model merges both entries of the loop head; without
  // these the first entry (loads last) forces vmcnt(0) there, i.e. a wait for every word store.
  {
    const __amdgpu_buffer_rsrc_t rnull = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(in), 0, 0, kRsrc3);
#pragma unroll
    for (int j = 0; j < 64; ++j) __builtin_amdgcn_raw_buffer_store_b32(0u, rnull, lane * 4, j * 256, 0);
  }
  PROF_DECL;
  for (; f < n_frames; f += stride) {
    PROF_VM(0);  // wait for this frame's loads
#pragma unroll
    for (int j = 0; j < 64; ++j) x[j] = y[j];
    const bool more = f + stride < n_frames;
    const uint32_t fn = more ? f + stride : f;  // the last frame is simply read again (no branch)
    __builtin_amdgcn_sched_barrier(0);
    load_rows(fn, 0);
    load_rows(fn, 16);
    __builtin_amdgcn_sched_barrier(0);
    // ---- FFT: DFT over the register index, twiddle, transpose, DFT over the old lane index ----
    dft64(x);
    {
      f32x2 wa[8], wb[8];
#pragma unroll
      for (int b = 1; b < 8; ++b) { wa[b] = tw[b * 64 + lane]; wb[b] = tw[(8 + b) * 64 + lane]; }
#pragma unroll
      for (int r = 1; r < 64; ++r) {
        const int q = bitrev_c(r, 6), a = q >> 3, b = q & 7;
        const f32x2 t = a == 0 ? wa[b] : (b == 0 ? wb[a] : cmul(wb[a], wa[b]));
        x[r] = cmul(x[r], t);
      }
    }
    PROF(1);  // DFT A + twiddle
    __builtin_amdgcn_sched_barrier(0);
    load_rows(fn, 32);
    load_rows(fn, 48);
    __builtin_amdgcn_sched_barrier(0);
    lds_order();
#pragma unroll
    for (int r = 0; r < 64; ++r) xch[bitrev_c(r, 6) * kWvXchPitch + lane] = x[r];
    lds_order();
#pragma unroll
    for (int l = 0; l < 64; ++l) x[l] = xch[lane * kWvXchPitch + l];
    lds_order();
    PROF(2);  // transpose
    dft64(x);  // register r now holds bin lane + 64 bitrev6(r), unscaled
    PROF(3);  // DFT B
    // ---- magnitudes to LDS in natural bin order (overlays the transpose image) ----
    if (rg.mag_mode == 2) {
#pragma unroll
      for (int r = 0; r < 64; ++r) pm[kWvPitch * bitrev_c(r, 6)] = mag_f32w(x[r] * scale, 2);
    } else if (rg.mag_mode == 0) {
#pragma unroll
      for (int r = 0; r < 64; ++r) pm[kWvPitch * bitrev_c(r, 6)] = mag_f32w(x[r] * scale, 0);
    } else {
#pragma unroll
      for (int r = 0; r < 64; ++r) pm[kWvPitch * bitrev_c(r, 6)] = mag_f32w(x[r] * scale, 1);
    }
    if (lane == 0) *det_cnt = 0u;
    PROF(4);  // mag
    lds_order();
    // ---- exclusive prefix sums relative to the lane's own 64-cell block ----
    {
      const int s0 = kWvPitch * (lane + 1);  // cells 64 lane .. 64 lane + 63 are contiguous slots
      // two levels (4 runs of 16): a prefix is ONE rounding at the magnitude of the block's running
      // total, not a chain of them -- matters in the squared-magnitude mode next to a strong target
      float off = 0.0f;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float acc = 0.0f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const float v = mag[s0 + 16 * c + e];
          pb[s0 + 16 * c + e] = off + acc;
          acc += v;
        }
        off += acc;
      }
      bs[lane + 1] = off;
    }
    lds_order();
    PROF(5);  // load issue + scan
    // ---- halos: block -1 (slots 0 .. 63) and block 64 (slots 68 * 65 ..): zeros or the wrapped image ----
    {
      constexpr int HI = kWvPitch * 65;  // slot of cell N
      const float m_lo = wrap ? mag[kWvPitch * 64 + lane] : 0.0f, p_lo = wrap ? pb[kWvPitch * 64 + lane] : 0.0f;
      const float m_hi = wrap ? mag[kWvPitch + lane] : 0.0f, p_hi = wrap ? pb[kWvPitch + lane] : 0.0f;
      const float b_lo = wrap ? bs[64] : 0.0f, b_hi = wrap ? bs[1] : 0.0f;
      mag[lane] = m_lo;
      pb[lane] = p_lo;
      mag[HI + lane] = m_hi;
      pb[HI + lane] = p_hi;
      if (lane == 0) {
        bs[0] = b_lo;
        bs[65] = b_hi;
      }
    }
    lds_order();

    PROF(6);  // halo
    // ---- CFAR sweep: cell k = lane + 64 p; branch-free body, mode hoisted; words go straight to
    // HBM (256 B per wave-instruction), only a peak bitmap stays in registers ----
    uint32_t h0 = 0, h1 = 0;
    auto sweep = [&](auto mode_c, auto group_c) {
      constexpr int B = RSP_WAVE_BATCH;  // cells per batch: LDS reads of the next batch fly during this one's arithmetic
      CellIn cur[B], nxt[B];
#pragma unroll
      for (int u = 0; u < B; ++u) cur[u] = fetch(group_c, kWvPitch * u, u);
#pragma unroll
      for (int p = 0; p < 64; p += B) {
        if (p + B < 64) {
#pragma unroll
          for (int u = 0; u < B; ++u) nxt[u] = fetch(group_c, kWvPitch * (p + B + u), p + B + u);
        }
        __builtin_amdgcn_sched_barrier(0);  // the next batch's reads stay ahead of this batch's arithmetic
#pragma unroll
        for (int u = 0; u < B; ++u) {
          const uint32_t w = word_of(mode_c, group_c, cur[u]);
          __builtin_amdgcn_raw_buffer_store_b32(w, rout, lane * 4, f * (uint32_t)(N * 4) + 256 * (p + u), 0);
          if (p + u < 32) h0 |= (w & 1u) << (p + u); else h1 |= (w & 1u) << (p + u - 32);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < B; ++u) cur[u] = nxt[u];
      }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    if (rg.peak_grouping) {
      if (rg.cfar_mode == 0) sweep(I0{}, std::true_type{});
      else if (rg.cfar_mode == 1) sweep(I1{}, std::true_type{});
      else sweep(I2{}, std::true_type{});
    } else {
      if (rg.cfar_mode == 0) sweep(I0{}, std::false_type{});
      else if (rg.cfar_mode == 1) sweep(I1{}, std::false_type{});
      else sweep(I2{}, std::false_type{});
    }
    PROF(7);  // sweep
    // ---- per-frame detection slots: peaks are rare (~1 per 1000 cells), recompute their words ----
    if (fcount) {
      while (h0 | h1) {
        const int p = h0 ? __ffs(h0) - 1 : 32 + __ffs(h1) - 1;
        if (h0) h0 &= h0 - 1; else h1 &= h1 - 1;
        uint32_t w;
        if (rg.peak_grouping) {
          const CellIn c = fetch(std::true_type{}, kWvPitch * p, p);
          w = rg.cfar_mode == 0 ? word_of(I0{}, std::true_type{}, c)
            : rg.cfar_mode == 1 ? word_of(I1{}, std::true_type{}, c) : word_of(I2{}, std::true_type{}, c);
        } else {
          const CellIn c = fetch(std::false_type{}, kWvPitch * p, p);
          w = rg.cfar_mode == 0 ? word_of(I0{}, std::false_type{}, c)
            : rg.cfar_mode == 1 ? word_of(I1{}, std::false_type{}, c) : word_of(I2{}, std::false_type{}, c);
        }
        const uint32_t slot = atomicAdd(det_cnt, 1u);
        if (slot < (uint32_t)kFrameDetCap) det_stage[slot] = make_uint2((uint32_t)(lane + 64 * p), w);
      }
      lds_order();
      const uint32_t cnt = *det_cnt;
      if (lane == 0) fcount[f] = cnt;
      if ((uint32_t)lane < min(cnt, (uint32_t)kFrameDetCap)) fdet[(size_t)f * kFrameDetCap + lane] = det_stage[lane];
    }
    lds_order();
    PROF(8);  // detection
  }
#ifdef RSP_WAVE_PROF
  if (blockIdx.x == 7 && threadIdx.x == 64)
    printf("wave prof (cycles): wait %llu dftA %llu xch %llu dftB %llu mag %llu scan %llu halo %llu sweep %llu det %llu\n", prof_acc[0],
           prof_acc[1], prof_acc[2], prof_acc[3], prof_acc[4], prof_acc[5], prof_acc[6], prof_acc[7], prof_acc[8]);
#endif
}

}  // namespace

// Opt-in (RSP_CHAIN_WAVE=1): measured 69-70 us per 4096 x 4096 batch against 57-65 us for
// chain1d_kernel on the same boxes (DESIGN.md 3.1f has the per-phase cycle counts and why).
bool chain1d_wave_supports(const Chain1dLaunch& a) {
  const char* on = getenv("RSP_CHAIN_WAVE");
  return on && on[0] == '1' && !a.fixed && a.log2n == kWvM && a.regs.algorithm == 0 && a.regs.cfar_mode <= 2 &&
         a.regs.R + a.regs.G + 1 <= kWvHalo;
}

hipError_t launch_chain1d_wave(const Chain1dLaunch& a) {
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  // one 4-wave workgroup per CU (its LDS and VGPR footprint admit no second one): a wave per SIMD
  uint32_t grid = (uint32_t)cus;
  if (grid > (a.n_frames + kWvWaves - 1) / kWvWaves) grid = (a.n_frames + kWvWaves - 1) / kWvWaves;
  constexpr int lds = kWvTwBytes + kWvWaves * kWvLdsBytes;
  static_assert(lds <= 160 * 1024, "one workgroup per CU");
  auto k = chain1d_wave_kernel;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k, dim3(grid), dim3(64 * kWvWaves), lds, a.stream, a.in, a.out, a.n_frames, a.regs,
                     a.frame_count, a.frame_det, a.zero_a, a.zero_b);
  return hipGetLastError();
}

}  // namespace rsp
