// Fused 1-D chain kernel: FFT -> magnitude -> CA-family CFAR, one launch.
//
// Replaces the stream wiring
//   cfar.streamNode := AXI4StreamBuffer() := mag.streamNode := AXI4StreamBuffer() := fft.streamNode
// (/root/reference/src/main/scala/FftMagCfarChain.scala:47): a frame enters as
// 2^M beats from HBM, stays in LDS through all three blocks and leaves as 2^M
// 32-bit words (FftMagCfarChainTester.scala:145-151,163-167).  Algorithmic HBM
// traffic: F32 8 B in + 4 B out per cell; FIXED16 4 B in + 4 B out.
//
// Workgroup = frames_per_wg(M) frames x (2^M / 16) threads, 16 cells per thread.
// CFAR sliding sums come from block-relative prefix sums held in LDS (blocks of
// 256 cells = one 16-lane row of the scan), so a window sum is a difference of
// two nearby prefixes plus at most one block total: exact for integers, no
// long-range cancellation for fp32.
#include <hip/hip_runtime.h>
#include <float.h>

#include "chain_regs.hpp"
#include "fft_lds.hpp"
#include "kernels.hpp"

namespace rsp {

// -DRSP_ABLATE builds (tools/ablate.sh) read a phase mask from ChainRegs::sub_window (unused by
// the CA-family kernel) to switch phases off at run time; production builds compile it away.
#ifdef RSP_ABLATE
#define ABL(bit) (rg.sub_window & (1 << (bit)))
#else
#define ABL(bit) false
#endif

// -DRSP_STAMP builds (tools/stamp.sh): s_memtime at the phase boundaries of a few workgroups, printed
// by one wave each -- where a workgroup's lifetime goes while the chip is loaded.  Production builds
// compile it away.
#ifdef RSP_STAMP
__device__ uint64_t g_stamp[16];  // per-thread copy would cost VGPRs; stamps are wave-uniform SGPR values
#define STAMP_DECL uint64_t st_[12] = {}
#define STAMP(i) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); st_[i] = __builtin_amdgcn_s_memtime(); } while (0)
#define STAMP_ARG , st_
#define STAMP_PARAM , uint64_t (&st_)[12]
#else
#define STAMP_DECL
#define STAMP(i)
#define STAMP_ARG
#define STAMP_PARAM
#endif

// ---------------------------------------------------------------- LDS layout per frame
// After the FFT the frame's LDS is re-used for the CFAR working set (4-byte slots):
//   mag : cell x in [-16, N+16)   at slot pad(x + 16)    (1-cell halo for peak grouping)
//   pb  : cell x in [-256, N+256] at slot pad(x + 256)   block-relative exclusive prefix
//   bs  : block b in [-1, N/256]  at slot b + 1          block totals
//   det : detection staging (count + kFrameDetCap x {bin, word})
// The halos hold zeros (edge = zero) or wrapped copies (edge = wrap), so the
// per-cell CFAR code needs no clamping and no edge branches.
constexpr int kHalo = 256;  // >= refWindow + guardWindow + 1 (checked on the host)

template <int M>
struct FrameLds {
  static constexpr int N = 1 << M;
  static constexpr int PADN = fft_image_slots(M);
  static constexpr int MAG_SLOTS = pad_slots(N + 32) + 1;
  static constexpr int PB_SLOTS = pad_slots(N + 2 * kHalo) + 2;
  static constexpr int BS_SLOTS = N / 256 + 3;  // blocks -1 .. N/256, + one slot that holds 0
  static constexpr int MAG_OFF = 0;
  static constexpr int PB_OFF = MAG_OFF + 4 * MAG_SLOTS;
  static constexpr int BS_OFF = PB_OFF + 4 * PB_SLOTS;
  static constexpr int DET_OFF = (BS_OFF + 4 * BS_SLOTS + 7) & ~7;
  static constexpr int CFAR_BYTES = DET_OFF + 8 + 8 * kFrameDetCap;
  static constexpr int FFT_BYTES = 8 * PADN;             // f32x2 per slot (FIXED16 uses 4 B)
  static constexpr int BYTES = ((CFAR_BYTES > FFT_BYTES ? CFAR_BYTES : FFT_BYTES) + 15) & ~15;
  static constexpr int ROM_BYTES = 4 * (N / 2);  // FIXED16: LDS copy of the Q2.14 twiddle ROM, per workgroup
};

__device__ __forceinline__ int mag_slot(int x) { return pad(x + 16); }
__device__ __forceinline__ int pb_slot(int x) { return pad(x + kHalo); }

#ifndef RSP_PART_FX
size_t chain1d_lds_bytes(int log2n) {
  switch (log2n) {
    case 8: return FrameLds<8>::BYTES * frames_per_wg(8);
    case 9: return FrameLds<9>::BYTES * frames_per_wg(9);
    case 10: return FrameLds<10>::BYTES * frames_per_wg(10);
    case 11: return FrameLds<11>::BYTES * frames_per_wg(11);
    case 12: return FrameLds<12>::BYTES * frames_per_wg(12);
    case 13: return FrameLds<13>::BYTES * frames_per_wg(13);
    default: return 0;
  }
}
#endif

// ---------------------------------------------------------------- magnitude (logMagMux)

// JPL approximation: RspChainTesterUtils.scala:120-127; mode select = MAG CSR 0
// (FftMagCfarChainTester.scala:84).  Spec of the other two modes: oracle/rsp_oracle.c.
__device__ __forceinline__ float mag_f32(f32x2 z, int mode) {
  const float ar = fabsf(z.x), ai = fabsf(z.y);
  const float u = fmaxf(ar, ai), v = fminf(ar, ai);
  const float jpl = fmaxf(u + v * 0.125f, u * 0.875f + v * 0.5f);
  if (mode == 2) return jpl;
  if (mode == 0) return z.x * z.x + z.y * z.y;
  return __log2f(fmaxf(jpl, FLT_MIN));
}

// lane l receives the value of lane l - S of its 16-lane row, 0 for the first S lanes
template <int S, typename V>
__device__ __forceinline__ V row_shr(V v) {
  static_assert(sizeof(V) == 4, "32-bit lanes");
  const int r = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x110 + S, 0xf, 0xf, true);
  return __builtin_bit_cast(V, r);
}

// ---------------------------------------------------------------- the kernel

// ---------------------------------------------------------------- shared front end
// Frame -> FFT (passes through LDS at fbase) -> magnitude of this thread's 16 bins in registers:
// mg[g * 2^WL + p] is bin (bitrev(p) << (M - WL)) | bitrev(g T + tau).
template <int M, bool FIXED, typename V, int FX = -1>
__device__ __forceinline__ void front_end(const void* __restrict__ in, uint32_t frame, bool live, int tau,
                                          unsigned char* fbase, const ChainRegs& rg,
                                          const void* __restrict__ tw,
                                          const int16_t* __restrict__ log_lut, uint32_t* rom,
                                          V (&mg)[16] STAMP_PARAM) {
  constexpr int N = 1 << M, NP = plan_np(M);
  if constexpr (!FIXED) {
    f32x2* buf = reinterpret_cast<f32x2*>(fbase);
    const f32x2* twf = reinterpret_cast<const f32x2*>(tw);
    f32x2 x[16];
    TwAll<M> twb;
    twb.load(tau, twf);
    {
      constexpr int W = plan_w(M, 0), LO = plan_lo(M, 0);
      // uniform base (SGPR pair) + one 32-bit per-thread byte offset; the per-register part is a
      // compile-time constant (the launcher keeps one launch's input below 4 GiB)
      const char* gbase = reinterpret_cast<const char*>(in);
      // a dead frame (ragged last workgroup) re-reads frame 0 and never stores
      const uint32_t voff = ((live ? frame : 0u) * (uint32_t)N + (uint32_t)elem_index<M, LO, W>(tau, 0)) * 8u;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const size_t eo = (size_t)(elem_index<M, LO, W>(0, e) - elem_index<M, LO, W>(0, 0)) * 8u;
        x[e] = *reinterpret_cast<const f32x2*>(gbase + (size_t)voff + eo);
      }
      if (rg.window) {  // pre-FFT window (build extension): one fp32 coefficient per sample
        const float* wt = reinterpret_cast<const float*>(rg.window) + elem_index<M, LO, W>(tau, 0);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const float wv = wt[elem_index<M, LO, W>(0, e) - elem_index<M, LO, W>(0, 0)];
          x[e] = x[e] * f32x2{wv, wv};
        }
      }
    }
    STAMP(1);
    if (!ABL(0)) pass_f32<M, 0>(x, twb.template get<0>());
    STAMP(2);
    // passes 1..NP-1 through LDS
    auto exchange = [&](auto pc) {
      constexpr int P = decltype(pc)::value;
      constexpr int W0 = plan_w(M, P - 1), LO0 = plan_lo(M, P - 1);
      constexpr int W1 = plan_w(M, P), LO1 = plan_lo(M, P);
      constexpr bool LAST = P == NP - 1;
      if (!ABL(3)) {
#pragma unroll
      for (int g = 0; g < (16 >> W0); ++g) {
        f32x2* b0 = buf + slot_base<M, LO0, W0, LAST>(tau, g);
#pragma unroll
        for (int r = 0; r < (1 << W0); ++r) b0[slot_delta<M, LO0, W0>(r)] = x[g * (1 << W0) + r];
      }
      }
      __syncthreads();
      if (!ABL(3)) {
#pragma unroll
      for (int g = 0; g < (16 >> W1); ++g) {
        const f32x2* b1 = buf + slot_base<M, LO1, W1, LAST>(tau, g);
#pragma unroll
        for (int r = 0; r < (1 << W1); ++r) x[g * (1 << W1) + r] = b1[slot_delta<M, LO1, W1>(r)];
      }
      }
      STAMP(2 * P + 1);
      if (!ABL(0)) pass_f32<M, P>(x, twb.template get<P>());
      STAMP(2 * P + 2);
    };
    exchange(std::integral_constant<int, 1>{});
    if constexpr (NP > 2) exchange(std::integral_constant<int, 2>{});
    if constexpr (NP > 3) exchange(std::integral_constant<int, 3>{});
    const float scale = 1.0f / (float)N;  // net 1/N: FftMagCfarChainTester.scala:77
    // mode select hoisted out of the per-bin loop (a uniform branch per bin costs ~15 SALU each).
    // JPL and squared magnitude are homogeneous, so the power-of-two 1/N scale is applied to the
    // magnitude (bit-identical to scaling the spectrum first) and two bins share every packed op.
    auto jpl_pairs = [&]() {
      const f32x2 k8 = {0.125f, 0.125f}, k78 = {0.875f, 0.875f}, k2 = {0.5f, 0.5f}, ss = {scale, scale};
#pragma unroll
      for (int e = 0; e < 16; e += 2) {
        const f32x2 a = x[e], b = x[e + 1];
        const f32x2 uu = {fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(b.x), fabsf(b.y))};
        const f32x2 vv = {fminf(fabsf(a.x), fabsf(a.y)), fminf(fabsf(b.x), fabsf(b.y))};
        const f32x2 t1 = __builtin_elementwise_fma(vv, k8, uu);
        const f32x2 t2 = __builtin_elementwise_fma(uu, k78, vv * k2);
        const f32x2 m = f32x2{fmaxf(t1.x, t2.x), fmaxf(t1.y, t2.y)} * ss;
        mg[e] = m.x;
        mg[e + 1] = m.y;
      }
    };
#ifdef RSP_COUNT_PATH
    if (true) jpl_pairs();
    else
#endif
    if (ABL(4)) {
#pragma unroll
      for (int e = 0; e < 16; ++e) mg[e] = x[e].x;
    } else if (rg.mag_mode == 2) {
      jpl_pairs();
    } else if (rg.mag_mode == 0) {
      const f32x2 s2 = {scale * scale, scale * scale};
#pragma unroll
      for (int e = 0; e < 16; e += 2) {
        const f32x2 a = x[e], b = x[e + 1];
        const f32x2 re = {a.x, b.x}, im = {a.y, b.y};
        const f32x2 m = __builtin_elementwise_fma(im, im, re * re) * s2;
        mg[e] = m.x;
        mg[e + 1] = m.y;
      }
    } else {
#pragma unroll
      for (int e = 0; e < 16; ++e) mg[e] = mag_f32(x[e] * scale, 1);
    }
  } else {
    const uint32_t* twq = reinterpret_cast<const uint32_t*>(tw);
    int xr[16], xi[16];
    {
      constexpr int W = plan_w(M, 0), LO = plan_lo(M, 0);
      const char* gbase = reinterpret_cast<const char*>(in);
      const uint32_t voff = ((live ? frame : 0u) * (uint32_t)N + (uint32_t)elem_index<M, LO, W>(tau, 0)) * 4u;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        // beat = {re[31:16], im[15:0]}: RspChainTesterUtils.scala:105-109
        const size_t eo = (size_t)(elem_index<M, LO, W>(0, e) - elem_index<M, LO, W>(0, 0)) * 4u;
        const uint32_t b = *reinterpret_cast<const uint32_t*>(gbase + (size_t)voff + eo);
        xr[e] = (int)(short)(b >> 16);
        xi[e] = (int)(short)(b & 0xffffu);
      }
      if (rg.window) {  // Q1.15 coefficient, product rounded half-up back to 16 bits (spec section 2.1)
        const int16_t* wt = reinterpret_cast<const int16_t*>(rg.window) + elem_index<M, LO, W>(tau, 0);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int wq = wt[elem_index<M, LO, W>(0, e) - elem_index<M, LO, W>(0, 0)];
          xr[e] = (int)(short)((xr[e] * wq + (1 << 14)) >> 15);
          xi[e] = (int)(short)((xi[e] * wq + (1 << 14)) >> 15);
        }
      }
    }
    // twiddle ROM -> LDS once per workgroup (the sample loads above are already in flight)
    for (int i = threadIdx.x; i < N / 2; i += wg_size(M)) rom[i] = twq[i];
    __syncthreads();
    fft_fx_frame<M, FX>(xr, xi, tau, fbase, rom, rg);
    if (rg.mag_mode == 2) {
#pragma unroll
      for (int e = 0; e < 16; ++e) mg[e] = jpl_fx(xr[e], xi[e]);
    } else {
#pragma unroll
      for (int e = 0; e < 16; ++e) mg[e] = mag_fx(xr[e], xi[e], rg, log_lut);
    }
  }

}

// magnitudes -> LDS in natural bin order, cell x at slot x' + PM (x' >> 4), x' = x + moff (moff a multiple
// of 16): PM = 1 is the FFT image's padding, PM = 4 the 16-byte-aligned one of the quad tail
template <int M, typename V, int PM = 1>
__device__ __forceinline__ void write_mag(V* mag, int moff, int tau, const V (&mg)[16], bool rev_order = false) {
  constexpr int T = threads_per_frame(M), NP = plan_np(M), WL = plan_w(M, NP - 1);
#pragma unroll
  for (int g = 0; g < (16 >> WL); ++g) {
    if (rev_order) {
      // useBitReverse = false: bin b sits at stream position bitrev(b) = (bitrev(g T + tau) << WL) | p:
      // the thread's 2^WL values of this group are consecutive positions inside one 16-run
      const int x = (int)((__brev((unsigned)(g * T + tau)) >> (32 - (M - WL))) << WL) + moff;
      V* mb = mag + x + PM * (x >> 4);
#pragma unroll
      for (int p = 0; p < (1 << WL); ++p) mb[p] = mg[g * (1 << WL) + p];
      continue;
    }
    // bin = (q << (M-WL)) | (g T + tau) (fft_lds.hpp, last pass): q << (M-WL) is a multiple of 16, so its
    // slot offset is constant, and consecutive lanes write consecutive slots
    const int x = g * T + tau + moff;
    V* mb = mag + x + PM * (x >> 4);
#pragma unroll
    for (int p = 0; p < (1 << WL); ++p) {
      constexpr int QS = (1 << (M - WL)) + PM * (1 << (M - WL - 4));
      mb[bitrev_c(p, WL) * QS] = mg[g * (1 << WL) + p];
    }
  }
}

// dense words to HBM (256 B per wave-instruction) + optional per-frame detection slots
template <int M, typename V>
__device__ __forceinline__ void emit_words(const uint32_t (&word)[16], uint32_t* __restrict__ out,
                                           uint32_t frame, bool live, int tau, uint32_t* det_cnt,
                                           uint2* det_stage, uint32_t* __restrict__ fcount,
                                           uint2* __restrict__ fdet, const V* cut_lds = nullptr, int cut_stride = 0) {
  constexpr int N = 1 << M, T = threads_per_frame(M);
  if (live && out && cut_lds) {  // sendCut = true: 64-bit beat {word, cut}; cut of cell tau + T j at cut_lds[cut_stride j]
    uint2* obase = reinterpret_cast<uint2*>(out) + (size_t)frame * N + tau;
#pragma unroll
    for (int j = 0; j < 16; ++j) obase[T * j] = make_uint2(word[j], __builtin_bit_cast(uint32_t, cut_lds[cut_stride * j]));
  } else if (live && out) {
    char* obase = reinterpret_cast<char*>(out);
    const uint32_t ooff = (frame * (uint32_t)N + (uint32_t)tau) * 4u;
#pragma unroll
    for (int j = 0; j < 16; ++j)
      *reinterpret_cast<uint32_t*>(obase + (size_t)ooff + (size_t)(T * j) * 4u) = word[j];
  }
#ifdef RSP_COUNT_PATH
  if (false) {
#else
  if (fcount) {
#endif
    uint32_t hits = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) hits |= (word[j] & 1u) << j;
    while (hits) {  // rare: ~1 peak per 1000 cells
      const int j = __ffs(hits) - 1;
      hits &= hits - 1;
      uint32_t w = word[0];
#pragma unroll
      for (int q = 1; q < 16; ++q) w = (j == q) ? word[q] : w;
      const uint32_t slot = atomicAdd(det_cnt, 1u);
      if (slot < (uint32_t)kFrameDetCap) det_stage[slot] = make_uint2((uint32_t)(tau + T * j), w);
    }
    // per-frame detection slots (no global atomics): count + first kFrameDetCap peaks
    __syncthreads();
    if (live) {
      const uint32_t cnt = *det_cnt;
      if (tau == 0) fcount[frame] = cnt;
      for (uint32_t i = tau; i < min(cnt, (uint32_t)kFrameDetCap); i += T)
        fdet[(size_t)frame * kFrameDetCap + i] = det_stage[i];
    }
  }
}

template <int M, bool FIXED, int FX>
__global__ void __launch_bounds__(wg_size(M))
chain1d_kernel(const void* __restrict__ in, uint32_t* __restrict__ out, uint32_t n_frames,
               ChainRegs rg, const void* __restrict__ tw, const int16_t* __restrict__ log_lut,
               uint32_t* __restrict__ fcount, uint2* __restrict__ fdet) {
  constexpr int N = 1 << M, T = threads_per_frame(M), FPW = frames_per_wg(M), NP = plan_np(M);
  using L = FrameLds<M>;
  using V = typename std::conditional<FIXED, int, float>::type;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int fl = tid / T, tau = tid % T;
  const uint32_t frame = blockIdx.x * FPW + fl;
  const bool live = frame < n_frames;  // dead frames still walk every barrier
  unsigned char* fbase = smem + fl * L::BYTES;

  V mg[16];
  STAMP_DECL;
  front_end<M, FIXED, V, FX>(in, frame, live, tau, fbase, rg, tw, log_lut,
                         reinterpret_cast<uint32_t*>(smem + (size_t)L::BYTES * FPW), mg STAMP_ARG);

  // ---- magnitudes to LDS in natural bin order ----
  V* mag = reinterpret_cast<V*>(fbase + L::MAG_OFF);
  V* pb = reinterpret_cast<V*>(fbase + L::PB_OFF);
  V* bs = reinterpret_cast<V*>(fbase + L::BS_OFF) + 1;  // bs[-1] .. bs[N/256]
  uint32_t* det_cnt = reinterpret_cast<uint32_t*>(fbase + L::DET_OFF);
  uint2* det_stage = reinterpret_cast<uint2*>(fbase + L::DET_OFF + 8);
  const bool wrap = rg.edge != 0;
  __syncthreads();  // every thread is done reading the FFT image this overlays
  write_mag<M, V>(mag, 16, tau, mg, rg.rev_order != 0);
  if (tau == 0) *det_cnt = 0u;
  __syncthreads();

  // ---- block-relative exclusive prefix sums: thread owns cells 16 tau .. 16 tau + 15 ----
  if (!ABL(2)) {
    V loc[16];
    V acc = V(0), first = V(0), last = V(0);
    const int m0 = mag_slot(16 * tau);  // the 16-cell chunk is contiguous in LDS
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const V v = mag[m0 + e];
      if (e == 0) first = v;
      if (e == 15) last = v;
      loc[e] = acc;
      acc += v;
    }
    // inclusive scan of chunk totals over the 16-lane DPP row (= 256 cells): row_shr shifts
    // zeros in at the row start (bound_ctrl), so no lane masking is needed
    V inc = acc;
    inc += row_shr<1>(inc);
    inc += row_shr<2>(inc);
    inc += row_shr<4>(inc);
    inc += row_shr<8>(inc);
    const V exc = row_shr<1>(inc);
    const int p0 = pb_slot(16 * tau);
#pragma unroll
    for (int e = 0; e < 16; ++e) pb[p0 + e] = exc + loc[e];
    // halos: zeros, or the wrapped image of the first / last block
    if (tau < 16) {
      const int ph = pb_slot(16 * tau + N);
#pragma unroll
      for (int e = 0; e < 16; ++e) pb[ph + e] = wrap ? exc + loc[e] : V(0);
    }
    if (tau >= T - 16) {
      const int pl = pb_slot(16 * tau - N);
#pragma unroll
      for (int e = 0; e < 16; ++e) pb[pl + e] = wrap ? exc + loc[e] : V(0);
    }
    if ((tau & 15) == 15) {
      const int blk = tau >> 4;
      bs[blk] = inc;
      if (blk == N / 256 - 1) bs[-1] = wrap ? inc : V(0);
      if (blk == 0) bs[N / 256] = wrap ? inc : V(0);
    }
    if (tau == 0) {
      pb[pb_slot(N + kHalo)] = V(0);
      mag[mag_slot(N)] = wrap ? first : V(0);
    }
    if (tau == T - 1) mag[mag_slot(-1)] = wrap ? last : V(0);
  }
  __syncthreads();

  // ---- CFAR: cell k = tau + T j; window geometry FftMagCfarChain.scala:105-106 ----
  // lagging cells [k-G-R, k-G), leading cells [k+G+1, k+G+R+1); a window sum is
  // pb[v] - pb[u] (+ the total of u's block when the window crosses a block edge).
  // The loop body is branch-free; cfarMode / peakGrouping are hoisted out of it.
  uint32_t word[16];
  {
    const int R = rg.R, G = rg.G;
    const int xu0 = tau - G - R, xv0 = tau - G, xu1 = tau + G + 1, xv1 = tau + G + R + 1;
    constexpr int JS = T + T / 16;  // slot stride between a thread's consecutive cells
    const V* pu0 = pb + pb_slot(xu0);
    const V* pv0 = pb + pb_slot(xv0);
    const V* pu1 = pb + pb_slot(xu1);
    const V* pv1 = pb + pb_slot(xv1);
    const V* pm = mag + mag_slot(tau);
    // immediate neighbours (peak grouping): +-1 cell = +-1 slot, +-2 across a pad slot
    const int dl = ((tau & 15) == 0) ? 2 : 1, dr = ((tau & 15) == 15) ? 2 : 1;
    // F32 threshold = comb * kA + kB: (div * scaler, 0) linear, (div, scaler) log domain
    const float kA = rg.linear ? rg.div_f * rg.scaler_f : rg.div_f, kB = rg.linear ? 0.0f : rg.scaler_f;
    auto cells = [&](auto mode_c, auto group_c) {
      constexpr int MODE = decltype(mode_c)::value;
      constexpr bool GROUP = decltype(group_c)::value;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        int bu0, bu1;
        bool c0, c1;
        if constexpr (T % 256 == 0) {  // block of cell k is a per-thread constant + j T/256
          bu0 = (xu0 >> 8) + j * (T / 256);
          bu1 = (xu1 >> 8) + j * (T / 256);
          c0 = (xv0 >> 8) != (xu0 >> 8);
          c1 = (xv1 >> 8) != (xu1 >> 8);
        } else {
          bu0 = (xu0 + T * j) >> 8;
          bu1 = (xu1 + T * j) >> 8;
          c0 = ((xv0 + T * j) >> 8) != bu0;
          c1 = ((xv1 + T * j) >> 8) != bu1;
        }
        const V cut = pm[JS * j];
        bool group_ok = true;
        if constexpr (GROUP) group_ok = cut > pm[JS * j - dl] && cut > pm[JS * j + dr];
        if constexpr (MODE == 3) {
          // CASH (cfarMode 3, CACFARType with includeCASH): each window is cut into sub-windows of
          // subWindowSize cells; per side the largest sub-window sum, then the smaller side
          // (BUILD-DEFINED, oracle/rsp_oracle.c orc_cfar_fixed).  Sub-window sums are prefix
          // differences like the whole-window sums, with the block fix-up computed per access.
          const int k = tau + T * j;
          V best[2];
#pragma unroll
          for (int side = 0; side < 2; ++side) {
            const int a = side == 0 ? k - G - R : k + G + 1;
            V b = V(0);
            bool first = true;
            for (int s0 = 0; s0 + rg.sub_window <= R; s0 += rg.sub_window) {
              const int u = a + s0, v = u + rg.sub_window;
              V ss = pb[pb_slot(v)] - pb[pb_slot(u)];
              if ((v >> 8) != (u >> 8)) ss += bs[u >> 8];
              b = first ? ss : (ss > b ? ss : b);
              first = false;
            }
            best[side] = b;
          }
          const V stat = CfarMath<V>::side(best[0] < best[1] ? best[0] : best[1], rg);
          word[j] = CfarMath<V>::finish(stat, cut, group_ok, k, M, rg);
        } else if constexpr (!FIXED) {
          // both windows in one packed subtract / fma; divSum, the 1/2 of CA and the scaler are
          // folded into kA (powers of two except the scaler: the same single rounding as the spec)
          const f32x2 pv = {pv0[JS * j], pv1[JS * j]}, pu = {pu0[JS * j], pu1[JS * j]};
          const f32x2 ff = {bs[bu0], bs[bu1]}, cm = {c0 ? 1.0f : 0.0f, c1 ? 1.0f : 0.0f};
          const f32x2 sw = __builtin_elementwise_fma(cm, ff, pv - pu);
          float comb;
          if constexpr (MODE == 0) comb = sw.x + sw.y;
          else if constexpr (MODE == 1) comb = fmaxf(sw.x, sw.y);
          else comb = fminf(sw.x, sw.y);
          const float thr = __fmaf_rn(comb, MODE == 0 ? kA * 0.5f : kA, kB);
          const uint32_t peak = (cut > thr) && group_ok;
          word[j] = (__float_as_uint(thr) & ~1u) | peak;
        } else {
          V s0 = pv0[JS * j] - pu0[JS * j];
          V s1 = pv1[JS * j] - pu1[JS * j];
          const V f0 = bs[bu0], f1 = bs[bu1];
          s0 += c0 ? f0 : V(0);
          s1 += c1 ? f1 : V(0);
          const V lagg = CfarMath<V>::side(s0, rg), lead = CfarMath<V>::side(s1, rg);
          V stat;
          if constexpr (MODE == 0) stat = CfarMath<V>::half_sum(lagg, lead);
          else if constexpr (MODE == 1) stat = lagg > lead ? lagg : lead;
          else stat = lagg < lead ? lagg : lead;
          word[j] = CfarMath<V>::finish(stat, cut, group_ok, tau + T * j, M, rg);
        }
      }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
#ifdef RSP_COUNT_PATH
    if (true) cells(I0{}, std::false_type{});
    else
#endif
    if (ABL(1)) {
#pragma unroll
      for (int j = 0; j < 16; ++j) word[j] = __builtin_bit_cast(uint32_t, pm[JS * j]);
    } else if (rg.peak_grouping) {
      if (rg.cfar_mode == 0) cells(I0{}, std::true_type{});
      else if (rg.cfar_mode == 1) cells(I1{}, std::true_type{});
      else if (rg.cfar_mode == 2) cells(I2{}, std::true_type{});
      else cells(I3{}, std::true_type{});
    } else {
      if (rg.cfar_mode == 0) cells(I0{}, std::false_type{});
      else if (rg.cfar_mode == 1) cells(I1{}, std::false_type{});
      else if (rg.cfar_mode == 2) cells(I2{}, std::false_type{});
      else cells(I3{}, std::false_type{});
    }
  }
  emit_words<M, V>(word, out, frame, live, tau, det_cnt, det_stage, fcount, fdet,
                   rg.send_cut ? mag + mag_slot(tau) : nullptr, T + T / 16);
}

// ---------------------------------------------------------------- quad tail (the default CA/GO/SO path)
// Same front end; the CFAR stage works on QUADS of 4 consecutive cells so that every LDS access of
// the tail is one conflict-free 16-byte ds_read/ds_write_b128 on UNPADDED images and every word store
// one 16-byte global store (1 KiB per wave-instruction).  Thread tau owns quads tau + T e, e = 0..3
// (cells 4 tau + 4 T e + i) for BOTH the prefix scan and the cells, so a quad's magnitudes are read
// from LDS once and stay in registers:
//   scan   in-quad prefix (3 adds), then an inclusive DPP scan of the quad totals over the wave: for a
//          fixed e the 64 lanes of a wave hold 64 consecutive quads = one 256-cell block, so the
//          block-relative prefix needs no cross-wave step and no second level;
//   cells  6 quad reads (4 prefixes, 2 magnitudes), 1 quad store.
// A window sum needs the exclusive prefix P at two positions.  With refWindow and guardWindow
// multiples of 4 the lagging positions k - G - R, k - G are quad-aligned; the leading ones
// k + G + 1, k + G + R + 1 are off by one, so that side uses P[x + 1] = P[x] + m[x] at the aligned
// x = k + G, k + G + R (two more quad reads of the magnitudes, no second prefix array).
// Halo cells of the quad tail's images, in two sizes: windows with R + G + 4 <= 48 (every reference configuration:
// R = 32, G = 4) take the SMALL one.  The halos are a fixed cost per frame, so they set the occupancy of SMALL
// frames: with 144 / 256-cell halos a 1024-point frame takes 11.4 KiB (three 4-frame workgroups per CU), with
// 48 / 64 cells 9.5 KiB (four) -- 53 -> 45.5 us per 16.7 M cells at 1024 points, 82 -> 43 us at 256 points.
template <bool SMALL> struct QuadHalo {
  static constexpr int MAG = SMALL ? 48 : 144;  // magnitude cells kept right of the frame: >= R + G + 4, a multiple of 16
  static constexpr int PB = SMALL ? 64 : 256;   // prefix cells kept on either side of the frame: >= R + G + 4, a multiple of 16
};
constexpr int kQHalo = QuadHalo<false>::MAG;     // the largest window the quad tail serves: R + G + 4 <= 144

template <int M, bool SMALL>
struct QuadLds {
  static constexpr int N = 1 << M, QH = QuadHalo<SMALL>::MAG, PBH = QuadHalo<SMALL>::PB;
  static constexpr int MAG_SLOTS = 16 + N + QH;      // cell x in [-16, N + QH) at x + 16
  static constexpr int PB_SLOTS = N + 2 * PBH;       // cell x in [-PBH, N + PBH) at x + PBH
  static constexpr int BS_SLOTS = N / 256 + 3;       // blocks -1 .. N/256, + one slot that holds 0
  static constexpr int MAG_OFF = 0;
  static constexpr int PB_OFF = MAG_OFF + 4 * MAG_SLOTS;
  static constexpr int BS_OFF = PB_OFF + 4 * PB_SLOTS;
  static constexpr int DET_OFF = (BS_OFF + 4 * BS_SLOTS + 7) & ~7;
  static constexpr int CFAR_BYTES = DET_OFF + 8 + 8 * kFrameDetCap;
  static constexpr int FFT_BYTES = 8 * fft_image_slots(M);
  static constexpr int BYTES = ((CFAR_BYTES > FFT_BYTES ? CFAR_BYTES : FFT_BYTES) + 15) & ~15;
  static constexpr int ROM_BYTES = 4 * (N / 2);
};

__device__ __forceinline__ uint32_t bits_of(float v) { return __float_as_uint(v); }
__device__ __forceinline__ uint32_t bits_of(int v) { return (uint32_t)v; }

template <typename V> struct Vec4;
template <> struct Vec4<float> { typedef float type __attribute__((ext_vector_type(4))); };
template <> struct Vec4<int> { typedef int type __attribute__((ext_vector_type(4))); };
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// the geometry the quad tail is built for (host-side dispatch, launch_m)
__host__ __device__ inline bool quad_tail_supports(int log2n, const ChainRegs& rg) {
  (void)log2n;
  return rg.algorithm == 0 && rg.cfar_mode <= 2 && (rg.R & 3) == 0 && (rg.G & 3) == 0 && rg.R + rg.G + 4 <= kQHalo;
}

template <int CTRL, int RMASK, bool BOUND, typename V>
__device__ __forceinline__ V dpp_v(V v) {
  static_assert(sizeof(V) == 4, "32-bit lanes");
  return __builtin_bit_cast(V, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, RMASK, 0xf, BOUND));
}
// v += (lane 15 of the previous row -> rows 1 and 3) / (lane 31 -> rows 2 and 3): ONE v_add_*_dpp whose
// disabled rows keep their value.  Written as inline asm because the compiler does not fold the masked
// broadcast into the add (it emits v_mov 0 + v_mov_dpp + v_add); the s_nop covers the 2 wait states a DPP
// read needs after a VALU write of the same register, which the compiler does not insert for asm.
template <int BCAST, typename V>
__device__ __forceinline__ V row_bcast_add(V v) {
  static_assert(BCAST == 15 || BCAST == 31, "row_bcast:15 / row_bcast:31");
  if constexpr (std::is_same<V, float>::value) {
    if constexpr (BCAST == 15) asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa" : "+v"(v));
    else asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc" : "+v"(v));
  } else {
    if constexpr (BCAST == 15) asm volatile("s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa" : "+v"(v));
    else asm volatile("s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc" : "+v"(v));
  }
  return v;
}
// inclusive scan over aligned segments of WD = 16, 32 or 64 lanes
template <int WD, typename V>
__device__ __forceinline__ V seg_scan(V v) {
  v += dpp_v<0x111, 0xf, true>(v);  // row_shr:1,2,4,8: inclusive scan of each 16-lane row
  v += dpp_v<0x112, 0xf, true>(v);
  v += dpp_v<0x114, 0xf, true>(v);
  v += dpp_v<0x118, 0xf, true>(v);
  if constexpr (WD >= 32) v = row_bcast_add<15>(v);
  if constexpr (WD >= 64) v = row_bcast_add<31>(v);
  return v;
}

template <int M, bool FIXED, int FX, bool SMALL>
__global__ void __launch_bounds__(wg_size(M))
chain1d_quad_kernel(const void* __restrict__ in, uint32_t* __restrict__ out, uint32_t n_frames,
                    ChainRegs rg, const void* __restrict__ tw, const int16_t* __restrict__ log_lut,
                    uint32_t* __restrict__ fcount, uint2* __restrict__ fdet) {
  constexpr int N = 1 << M, T = threads_per_frame(M), FPW = frames_per_wg(M);
  constexpr int WD = T < 64 ? T : 64;  // lanes of a wave that belong to one frame
  constexpr int SPB = 64 / WD;         // lane segments (values of e) per 256-cell block
  using L = QuadLds<M, SMALL>;
  constexpr int QH = L::QH, PBH = L::PBH;
  using V = typename std::conditional<FIXED, int, float>::type;
  using V4 = typename Vec4<V>::type;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int fl = tid / T, tau = tid % T;
  const uint32_t frame = blockIdx.x * FPW + fl;
  const bool live = frame < n_frames;  // dead frames still walk every barrier
  unsigned char* fbase = smem + fl * L::BYTES;

  V mg[16];
  STAMP_DECL;
  STAMP(0);
  front_end<M, FIXED, V, FX>(in, frame, live, tau, fbase, rg, tw, log_lut,
                         reinterpret_cast<uint32_t*>(smem + (size_t)L::BYTES * FPW), mg STAMP_ARG);
  STAMP(7);

  V* mag = reinterpret_cast<V*>(fbase + L::MAG_OFF) + 16;   // mag[x], x in [-16, N + QH)
  V* pb = reinterpret_cast<V*>(fbase + L::PB_OFF) + PBH;    // pb[x], x in [-PBH, N + PBH)
  V* bs = reinterpret_cast<V*>(fbase + L::BS_OFF) + 1;      // bs[-1] .. bs[N/256], bs[N/256 + 1] = 0
  uint32_t* det_cnt = reinterpret_cast<uint32_t*>(fbase + L::DET_OFF);
  uint2* det_stage = reinterpret_cast<uint2*>(fbase + L::DET_OFF + 8);
  const bool wrap = rg.edge != 0;
  __syncthreads();  // every thread is done reading the FFT image this overlays
  {  // magnitudes to LDS in natural bin order: register (g, p) holds bin (bitrev(p) << (M - WL)) | (g T + tau)
    constexpr int NP = plan_np(M), WL = plan_w(M, NP - 1);
    if (rg.rev_order) {  // useBitReverse = false: bin b at stream position bitrev(b) = (bitrev(g T + tau) << WL) | p
#pragma unroll
      for (int g = 0; g < (16 >> WL); ++g) {
        V* mb = mag + ((__brev((unsigned)(g * T + tau)) >> (32 - (M - WL))) << WL);
#pragma unroll
        for (int p = 0; p < (1 << WL); ++p) mb[p] = mg[g * (1 << WL) + p];
      }
    } else {
#pragma unroll
      for (int g = 0; g < (16 >> WL); ++g) {
        V* mb = mag + g * T + tau;
#pragma unroll
        for (int p = 0; p < (1 << WL); ++p) mb[bitrev_c(p, WL) << (M - WL)] = mg[g * (1 << WL) + p];
      }
    }
  }
  if (tau == 0) {
    *det_cnt = 0u;
    bs[N / 256 + 1] = V(0);  // the "no block total" slot of the window fix-ups
  }
  __syncthreads();
  STAMP(8);

  // ---- scan: block-relative exclusive prefix sums of the thread's 4 quads ----
  V4 mq[4];
  {
    const V4 zero4 = {V(0), V(0), V(0), V(0)};
    V inc[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) mq[e] = *reinterpret_cast<const V4*>(mag + 4 * (tau + T * e));
    V p1[4], p2[4], p3[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      p1[e] = mq[e][0];
      p2[e] = p1[e] + mq[e][1];
      p3[e] = p2[e] + mq[e][2];
      inc[e] = seg_scan<WD, V>(p3[e] + mq[e][3]);
    }
    V tot[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int q = tau + T * e;  // quad index; block q >> 6, position q & 63
      tot[e] = inc[e];            // inclusive through this quad, within the lane segment
      V exc = inc[e] - (p3[e] + mq[e][3]);
      if constexpr (SPB > 1) {    // a block spans SPB values of e (frames of 256 / 512 points): carry the earlier ones
        V carry = V(0);
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2) {
          if (e2 < e && e2 >= e - e % SPB)
            carry += __builtin_bit_cast(V, __shfl(__builtin_bit_cast(int, inc[e2]), (threadIdx.x & 63 & ~(WD - 1)) | (WD - 1)));
        }
        exc += carry;
        tot[e] += carry;
      }
      const V4 pq = V4{V(0), p1[e], p2[e], p3[e]} + exc;
      *reinterpret_cast<V4*>(pb + 4 * q) = pq;
      // halos: zeros, or the wrapped image of the first / last block (prefixes), of the first QH
      // cells (magnitudes right of the frame) and of the last cell (left neighbour of cell 0)
      if (q < PBH / 4) {
        *reinterpret_cast<V4*>(pb + 4 * q + N) = wrap ? pq : zero4;
        if (q < QH / 4) *reinterpret_cast<V4*>(mag + 4 * q + N) = wrap ? mq[e] : zero4;
      }
      if (q >= N / 4 - PBH / 4) *reinterpret_cast<V4*>(pb + 4 * q - N) = wrap ? pq : zero4;
      if (q == N / 4 - 1) mag[-1] = wrap ? mq[e][3] : V(0);
    }
    // block totals: the lane that holds a block's last quad (for T >= 64 the same lane for every e)
    auto block_total = [&](int e) {
      const int blk = (tau + T * e) >> 6;
      bs[blk] = tot[e];
      if (blk == N / 256 - 1) bs[-1] = wrap ? tot[e] : V(0);
      if (blk == 0) bs[N / 256] = wrap ? tot[e] : V(0);
    };
    if constexpr (T % 64 == 0) {
      if ((tau & 63) == 63) {
#pragma unroll
        for (int e = 0; e < 4; ++e) block_total(e);
      }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (((tau + T * e) & 63) == 63) block_total(e);
    }
  }
  __syncthreads();

  STAMP(9);
  // ---- CFAR on quads: cells k0 + i, k0 = 4 tau + 4 T e, i = 0..3 ----
  uint32_t word[16];
  {
    const int R = rg.R, G = rg.G;
    const int k00 = 4 * tau;
    const V* pa = pb + (k00 - G);       // P[k - G]        lagging end
    const V* pbq = pb + (k00 - G - R);  // P[k - G - R]    lagging start
    const V* pe = pb + (k00 + G + R);   // P[k + G + R]    + m = P[k + G + R + 1] leading end
    const V* ps = pb + (k00 + G);       // P[k + G]        + m = P[k + G + 1]     leading start
    const V* me = mag + (k00 + G + R);
    const V* ms = mag + (k00 + G);
    const V* mc = mag + k00;
    constexpr int ES = 4 * T;  // cells between a thread's consecutive quads
    const float kA = rg.linear ? rg.div_f * rg.scaler_f : rg.div_f, kB = rg.linear ? 0.0f : rg.scaler_f;
    // block (256 cells) of the two window starts, and whether the window ends in the next block: then
    // the start block's total is added.  A quad never straddles a block, so this is per quad; the
    // "no" case reads the slot that holds 0, which keeps the read unconditional (no divergent branch).
    constexpr int ZS = N / 256 + 1;
    auto cells = [&](auto mode_c, auto group_c) {
      constexpr int MODE = decltype(mode_c)::value;
      constexpr bool GROUP = decltype(group_c)::value;
      int i0[4], i1[4];
      if constexpr ((4 * T) % 256 == 0) {  // a thread's quads sit whole blocks apart: same case for all four
        const int bu0 = (k00 - G - R) >> 8, bu1 = (k00 + G) >> 8;
        const bool z0 = ((k00 - G) >> 8) == bu0, z1 = ((k00 + G + R) >> 8) == bu1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          i0[e] = z0 ? ZS : bu0 + (4 * T / 256) * e;
          i1[e] = z1 ? ZS : bu1 + (4 * T / 256) * e;
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int k0 = k00 + 4 * T * e;
          const int bu0 = (k0 - G - R) >> 8, bu1 = (k0 + G) >> 8;
          i0[e] = ((k0 - G) >> 8) != bu0 ? bu0 : ZS;
          i1[e] = ((k0 + G + R) >> 8) != bu1 ? bu1 : ZS;
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int k0 = k00 + 4 * T * e;
        const V4 Pa = *reinterpret_cast<const V4*>(pa + ES * e), Pb = *reinterpret_cast<const V4*>(pbq + ES * e);
        const V4 Pe = *reinterpret_cast<const V4*>(pe + ES * e), Ps = *reinterpret_cast<const V4*>(ps + ES * e);
        const V4 Me = *reinterpret_cast<const V4*>(me + ES * e), Ms = *reinterpret_cast<const V4*>(ms + ES * e);
        const V4 cut = mq[e];
        const V f0 = bs[i0[e]], f1 = bs[i1[e]];
        V nl = V(0), nr = V(0);
        if constexpr (GROUP) {
          nl = mc[ES * e - 1];
          nr = mc[ES * e + 4];
        }
        if constexpr (!FIXED) {
          // two cells per packed op: sums, combination and threshold of a quad in 14-16 v_pk ops
          const f32x2 kAA = {MODE == 0 ? kA * 0.5f : kA, MODE == 0 ? kA * 0.5f : kA}, kBB = {kB, kB};
          const f32x2 f00 = {f0, f0}, f11 = {f1, f1}, f01 = {f0 + f1, f0 + f1};
          auto half = [&](auto hc) {
            constexpr int h = decltype(hc)::value;
            const f32x2 a = __builtin_shufflevector(Pa, Pa, 2 * h, 2 * h + 1), b = __builtin_shufflevector(Pb, Pb, 2 * h, 2 * h + 1);
            const f32x2 pe2 = __builtin_shufflevector(Pe, Pe, 2 * h, 2 * h + 1), ps2 = __builtin_shufflevector(Ps, Ps, 2 * h, 2 * h + 1);
            const f32x2 me2 = __builtin_shufflevector(Me, Me, 2 * h, 2 * h + 1), ms2 = __builtin_shufflevector(Ms, Ms, 2 * h, 2 * h + 1);
            const f32x2 c2 = __builtin_shufflevector(cut, cut, 2 * h, 2 * h + 1);
            const f32x2 lag = a - b;
            const f32x2 lead = (pe2 - ps2) + (me2 - ms2);
            f32x2 thr2;
            if constexpr (MODE == 0) {
              thr2 = __builtin_elementwise_fma((lag + lead) + f01, kAA, kBB);
            } else {
              const f32x2 lg = lag + f00, ld = lead + f11;
              const f32x2 comb = {MODE == 1 ? fmaxf(lg.x, ld.x) : fminf(lg.x, ld.x),
                                  MODE == 1 ? fmaxf(lg.y, ld.y) : fminf(lg.y, ld.y)};
              thr2 = __builtin_elementwise_fma(comb, kAA, kBB);
            }
            if constexpr (!GROUP) {
              // cut > thr  <=>  thr - cut < 0: the sign bit of the (correctly rounded, never flushed: both
              // operands are normal and differ by >= 1 ulp) difference IS the peak flag
              const f32x2 d = thr2 - c2;
              word[4 * e + 2 * h] = (__float_as_uint(thr2.x) & ~1u) | (__float_as_uint(d.x) >> 31);
              word[4 * e + 2 * h + 1] = (__float_as_uint(thr2.y) & ~1u) | (__float_as_uint(d.y) >> 31);
            } else {
#pragma unroll
              for (int u = 0; u < 2; ++u) {
                const int i = 2 * h + u;
                const bool group_ok = cut[i] > (i == 0 ? nl : cut[i - 1]) && cut[i] > (i == 3 ? nr : cut[i + 1]);
                const float thr = thr2[u];
                const uint32_t peak = (cut[i] > thr) && group_ok;
                word[4 * e + i] = (__float_as_uint(thr) & ~1u) | peak;
              }
            }
          };
          half(std::integral_constant<int, 0>{});
          half(std::integral_constant<int, 1>{});
        } else {
          const V4 lag = (Pa - Pb) + f0;
          const V4 lead = ((Pe - Ps) + (Me - Ms)) + f1;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            bool group_ok = true;
            if constexpr (GROUP) group_ok = cut[i] > (i == 0 ? nl : cut[i - 1]) && cut[i] > (i == 3 ? nr : cut[i + 1]);
            const V sl = CfarMath<V>::side(lag[i], rg), sd = CfarMath<V>::side(lead[i], rg);
            V stat;
            if constexpr (MODE == 0) stat = CfarMath<V>::half_sum(sl, sd);
            else if constexpr (MODE == 1) stat = sl > sd ? sl : sd;
            else stat = sl < sd ? sl : sd;
            word[4 * e + i] = CfarMath<V>::finish(stat, cut[i], group_ok, k0 + i, M, rg);
#ifdef RSP_DBG_OUT  // debugging side builds only: expose the tail's inputs instead of the words
            word[4 * e + i] = (uint32_t)(RSP_DBG_OUT == 1 ? cut[i] : RSP_DBG_OUT == 2 ? Pa[i] : RSP_DBG_OUT == 3 ? Me[i] : Pe[i]);
#endif
          }
          // one quad's loads at a time: hoisting all four quads' 24 LDS reads above the 64-bit threshold arithmetic
          // took the FIXED16 kernel to 142 VGPRs (three workgroups per CU at 4096 points instead of four)
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
#ifdef RSP_COUNT_PATH  // static instruction counts of ONE path (tools/count_insts.sh): CA, no grouping
    cells(I0{}, std::false_type{});
#else
    if (rg.peak_grouping) {
      if (rg.cfar_mode == 0) cells(I0{}, std::true_type{});
      else if (rg.cfar_mode == 1) cells(I1{}, std::true_type{});
      else cells(I2{}, std::true_type{});
    } else {
      if (rg.cfar_mode == 0) cells(I0{}, std::false_type{});
      else if (rg.cfar_mode == 1) cells(I1{}, std::false_type{});
      else cells(I2{}, std::false_type{});
    }
#endif
  }

  STAMP(10);
  // ---- dense words: one 16-byte store per quad (1 KiB per wave-instruction) ----
  if (live && out && rg.send_cut) {  // sendCut = true: 64-bit beat {word, cut}, two 16-byte stores per quad
    char* obase = reinterpret_cast<char*>(out) + ((size_t)frame * N + 4u * (size_t)tau) * 8u;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const V c0 = mq[e][0], c1 = mq[e][1], c2 = mq[e][2], c3 = mq[e][3];
      const u32x4 lo4 = {word[4 * e], bits_of(c0), word[4 * e + 1], bits_of(c1)};
      const u32x4 hi4 = {word[4 * e + 2], bits_of(c2), word[4 * e + 3], bits_of(c3)};
      *reinterpret_cast<u32x4*>(obase + (size_t)(32 * T * e)) = lo4;
      *reinterpret_cast<u32x4*>(obase + (size_t)(32 * T * e) + 16) = hi4;
    }
  } else if (live && out) {
    char* obase = reinterpret_cast<char*>(out);
    const uint32_t ooff = (frame * (uint32_t)N + 4u * (uint32_t)tau) * 4u;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const u32x4 w4 = {word[4 * e], word[4 * e + 1], word[4 * e + 2], word[4 * e + 3]};
      *reinterpret_cast<u32x4*>(obase + (size_t)ooff + (size_t)(16 * T * e)) = w4;
    }
  }
#ifdef RSP_STAMP
  STAMP(11);
  if ((blockIdx.x & 255) == 77 && (threadIdx.x & 63) == 0)
    printf("stamp wg %u wave %u: load %llu p0 %llu x1 %llu p1 %llu x2 %llu p2mag %llu magw %llu scan %llu cells %llu store %llu total %llu\n",
           blockIdx.x, threadIdx.x >> 6, st_[1] - st_[0], st_[2] - st_[1], st_[3] - st_[2], st_[4] - st_[3], st_[5] - st_[4],
           st_[7] - st_[5], st_[8] - st_[7], st_[9] - st_[8], st_[10] - st_[9], st_[11] - st_[10], st_[11] - st_[0]);
#endif
#ifdef RSP_COUNT_PATH
  if (false) {
#else
  if (fcount) {
#endif
    uint32_t any = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) any |= word[j];
    if (any & 1u) {  // rare: ~1 peak per 1000 cells; kept compact (a loop, not 16 unrolled copies)
      uint32_t hits = 0;
#pragma unroll
      for (int j = 0; j < 16; ++j) hits |= (word[j] & 1u) << j;
      while (hits) {
        const int j = __ffs(hits) - 1;
        hits &= hits - 1;
        uint32_t w = word[0];
#pragma unroll
        for (int q = 1; q < 16; ++q) w = (j == q) ? word[q] : w;
        const uint32_t slot = atomicAdd(det_cnt, 1u);
        if (slot < (uint32_t)kFrameDetCap)
          det_stage[slot] = make_uint2((uint32_t)(4 * tau + 4 * T * (j >> 2) + (j & 3)), w);
      }
    }
    // per-frame detection slots (no global atomics): count + first kFrameDetCap peaks
    __syncthreads();
    if (live) {
      const uint32_t cnt = *det_cnt;
      if (tau == 0) fcount[frame] = cnt;
      for (uint32_t i = tau; i < min(cnt, (uint32_t)kFrameDetCap); i += T)
        fdet[(size_t)frame * kFrameDetCap + i] = det_stage[i];
    }
  }
}

// ---------------------------------------------------------------- GOS / ordered-statistic CFAR
// GOSCFARType / GOSCACFARType with cfarAlgorithm = GOS (FftMagCfarChainTester.scala:105-127):
// the per-side statistic is the indexLagg-th / indexLead-th smallest cell of the window.
// The hardware keeps each window sorted with a linear insertion sorter; here every thread keeps
// ONE sorted window in registers: it bitonic-sorts the R cells starting at its first window
// start, then slides it (branch-free delete + insert, cmp/cndmask + med3 per element) over its
// run of consecutive starts, writing the two order statistics of every start to LDS.  The
// lagging window of cell k starts at k - G - R, the leading one at k + G + 1, so a cell needs two
// lookups.  Starts run over [-(G+R), N + G]; cells outside the frame come from the magnitude
// halo (zeros or the wrapped image).

template <typename V> __device__ __forceinline__ V vmin(V a, V b) { return a < b ? a : b; }
template <typename V> __device__ __forceinline__ V vmax(V a, V b) { return a > b ? a : b; }
// fminf / fmaxf quiet their operands first (a v_max x, x each); magnitudes are never NaN, and v_med3 with an
// infinity is the same selection in ONE instruction (the infinities sit in SGPRs)
template <> __device__ __forceinline__ float vmin<float>(float a, float b) { return __builtin_amdgcn_fmed3f(a, b, -__builtin_inff()); }
template <> __device__ __forceinline__ float vmax<float>(float a, float b) { return __builtin_amdgcn_fmed3f(a, b, __builtin_inff()); }
// the same for the selection network, where the compiler folds the med3-with-infinity back into v_min / v_max and
// canonicalises both operands first (3 instructions per selection): the bare instruction
template <typename V> __device__ __forceinline__ V vmin1(V a, V b) { return a < b ? a : b; }
template <typename V> __device__ __forceinline__ V vmax1(V a, V b) { return a > b ? a : b; }
template <> __device__ __forceinline__ float vmin1<float>(float a, float b) {
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
template <> __device__ __forceinline__ float vmax1<float>(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// median of three, a <= c guaranteed by the caller
template <typename V> __device__ __forceinline__ V vmed3(V a, V b, V c) { return vmin(vmax(a, b), c); }
template <> __device__ __forceinline__ float vmed3<float>(float a, float b, float c) {
  return __builtin_amdgcn_fmed3f(a, b, c);
}

// The sorted window lives in ONE vector value (R consecutive VGPRs): every access below has a
// compile-time index except the order-statistic pick, which the compiler then lowers to an indexed
// register read (s_set_gpr_idx_on + v_mov: 3 instructions) -- the index-th register, wave-uniform.
// With a plain C array the same pick became a scratch-memory copy (4x slower) or, blended by hand,
// log2(R) levels of v_bfi (31 instructions per pick at R = 32).
template <typename V, int R> struct WinVec { typedef V type __attribute__((ext_vector_type(R))); };

// Batcher's odd-even merge sort, ascending: 191 compare-exchanges at R = 32 (bitonic: 240), 543 at R = 64 (672);
// every index is a compile-time constant after unrolling
template <typename V, int R>
__device__ __forceinline__ void sort_window(typename WinVec<V, R>::type& s) {
#pragma unroll
  for (int p = 1; p < R; p <<= 1) {
#pragma unroll
    for (int k = p; k >= 1; k >>= 1) {
#pragma unroll
      for (int j = k % p; j <= R - 1 - k; j += 2 * k) {
#pragma unroll
        for (int i = 0; i <= (k - 1 < R - j - k - 1 ? k - 1 : R - j - k - 1); ++i) {
          if ((i + j) / (2 * p) == (i + j + k) / (2 * p)) {
            const V a = s[i + j], b = s[i + j + k];
            s[i + j] = vmin(a, b);
            s[i + j + k] = vmax(a, b);
          }
        }
      }
    }
  }
}

// lane mask of a < b into an SGPR pair / select by such a mask.  Inline asm: the compiler pairs every compare
// with its select through VCC (one register: compare i+1 cannot start before select i has read it) and pads
// each pair with s_nop 1 for the VALU-writes-mask hazard; batches of 8 explicit masks need no padding.
__device__ __forceinline__ unsigned long long cmp_lt_mask(float a, float b) {
  unsigned long long m;
  asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(m) : "v"(a), "v"(b));
  return m;
}
__device__ __forceinline__ unsigned long long cmp_lt_mask(int a, int b) {
  unsigned long long m;
  asm volatile("v_cmp_lt_i32_e64 %0, %1, %2" : "=s"(m) : "v"(a), "v"(b));
  return m;
}
template <typename V>
__device__ __forceinline__ V select_mask(unsigned long long m, V if_set, V if_clear) {
  V r;
  asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(if_clear), "v"(if_set), "s"(m));
  return r;
}

// sorted s: remove one element equal to `old`, insert `nw`, stay sorted
template <typename V, int R>
__device__ __forceinline__ void slide(typename WinVec<V, R>::type& s, V old, V nw) {
  V t[R - 1];
#pragma unroll
  for (int b0 = 0; b0 < R - 1; b0 += 8) {
    unsigned long long m[8];
#pragma unroll
    for (int i = b0; i < b0 + 8 && i < R - 1; ++i) {
      const V a = s[i];
      m[i - b0] = cmp_lt_mask(a, old);
    }
    if (R - 1 - b0 < 3) asm volatile("s_nop 1");  // a short last batch: keep 2 wait states between mask and select
#pragma unroll
    for (int i = b0; i < b0 + 8 && i < R - 1; ++i) {
      const V a = s[i], b = s[i + 1];
      t[i] = select_mask<V>(m[i - b0], a, b);
    }
  }
  s[0] = vmin(t[0], nw);
#pragma unroll
  for (int i = 1; i < R - 1; ++i) s[i] = vmed3(t[i - 1], nw, t[i]);
  s[R - 1] = vmax(t[R - 2], nw);
}

#ifndef RSP_GOS_SPLIT
#define RSP_GOS_SPLIT 1
#endif
struct GosLayout {  // byte offsets inside a frame's LDS, computed on the host
  int32_t frame_bytes, o1_off, o2_off, det_off, run;  // run = consecutive window starts per thread
};

template <typename V, int R>
__device__ __forceinline__ void gos_stage(const V* mag, V* o1, V* o2, int tau, int run, int G,
                                          int idx_lagg, int idx_lead) {
  const int a0 = -(G + R) + run * tau;  // first window start of this thread
  typename WinVec<V, R>::type s;
#pragma unroll
  for (int i = 0; i < R; ++i) s[i] = mag[pad(a0 + i + kHalo)];
  sort_window<V, R>(s);
  const bool two = idx_lagg != idx_lead;
  for (int st = 0; st < run; ++st) {
    const int oi = pad(run * tau + st);
    o1[oi] = s[idx_lagg];
    if (two) o2[oi] = s[idx_lead];
    if (st + 1 < run) {
      const V old = mag[pad(a0 + st + kHalo)], nw = mag[pad(a0 + st + R + kHalo)];
      slide<V, R>(s, old, nw);
    }
  }
}

// k-th smallest (0-based, wave-uniform k) of a bitonic sequence of SZ = 2^n values: a half-cleaner per level leaves
// the SZ/2 smallest (min side) or largest (max side) as a bitonic sequence again, so only the side that holds rank k
// is computed: SZ - 1 min / max operations in all, every index a compile-time constant, the side a scalar branch.
template <int SZ, typename V>
__device__ __forceinline__ V select_bitonic(const V (&x)[SZ], int k) {
  if constexpr (SZ == 1) {
    return x[0];
  } else {
    V h[SZ / 2];
    if (k & (SZ / 2)) {
#pragma unroll
      for (int i = 0; i < SZ / 2; ++i) h[i] = vmax1(x[i], x[i + SZ / 2]);
    } else {
#pragma unroll
      for (int i = 0; i < SZ / 2; ++i) h[i] = vmin1(x[i], x[i + SZ / 2]);
    }
    return select_bitonic<SZ / 2, V>(h, k);
  }
}

// The same statistics with the window SPLIT: the RUN windows of a thread (starts a0 .. a0 + RUN - 1) all contain
// the cells B = [a0 + RUN - 1, a0 + R); only the other RUN - 1 cells D change from start to start (one leaves at
// the front, one enters past B).  B is sorted once, D is kept sorted with the delete + insert slide -- 3 (RUN - 1)
// operations per start instead of 3 R -- and the order statistic is selected from the bitonic sequence
// [B ascending | D descending] with R - 1 min / max.  R = 32, RUN = 17: two 16-element sorts + 16 x 48 + 17 x 31 =
// 1547 operations per 17 starts against 382 + 16 x 94 = 1886 with one 32-cell sorted window.
template <typename V, int R, int RUN>
__device__ __forceinline__ void gos_stage_split(const V* mag, V* o1, V* o2, int tau, int G, int idx_lagg, int idx_lead) {
  constexpr int ND = RUN - 1, NB = R - ND;
  static_assert(NB >= 1 && (R & (R - 1)) == 0, "a common part and a power-of-two window");
  const int a0 = -(G + R) + RUN * tau;  // first window start of this thread
  typename WinVec<V, ND>::type d;
  typename WinVec<V, NB>::type b;
#pragma unroll
  for (int i = 0; i < ND; ++i) d[i] = mag[pad(a0 + i + kHalo)];
#pragma unroll
  for (int i = 0; i < NB; ++i) b[i] = mag[pad(a0 + ND + i + kHalo)];
  sort_window<V, ND>(d);
  sort_window<V, NB>(b);
  const bool two = idx_lagg != idx_lead;
#pragma unroll 1
  for (int st = 0; st < RUN; ++st) {
    V seq[R];  // [B ascending | D descending]
#pragma unroll
    for (int i = 0; i < NB; ++i) seq[i] = b[i];
#pragma unroll
    for (int i = 0; i < ND; ++i) seq[NB + i] = d[ND - 1 - i];
    int k1 = idx_lagg, k2 = idx_lead;
    asm volatile("" : "+s"(k1), "+s"(k2));  // keep the five side branches inside the loop (no 32-way unswitching)
    const int oi = pad(RUN * tau + st);
    o1[oi] = select_bitonic<R, V>(seq, k1);
    if (two) o2[oi] = select_bitonic<R, V>(seq, k2);
    if (st + 1 < RUN) {
      const V old = mag[pad(a0 + st + kHalo)], nw = mag[pad(a0 + st + R + kHalo)];
      slide<V, ND>(d, old, nw);
    }
  }
}

// BIG = the 64-cell window: its sorted window alone is 64 + 63 registers, so it is a kernel of its own -- as one path
// of a common kernel it set the register count (141 + scratch) and with it the occupancy (one 512-thread workgroup
// per CU at 8192 points) of every other window size.
template <int M, bool FIXED, bool BIG, int FX>
__global__ void __launch_bounds__(wg_size(M))
chain1d_gos_kernel(const void* __restrict__ in, uint32_t* __restrict__ out, uint32_t n_frames,
                   ChainRegs rg, GosLayout lay, const void* __restrict__ tw,
                   const int16_t* __restrict__ log_lut, uint32_t* __restrict__ fcount,
                   uint2* __restrict__ fdet) {
  constexpr int N = 1 << M, T = threads_per_frame(M), FPW = frames_per_wg(M);
  using V = typename std::conditional<FIXED, int, float>::type;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int fl = tid / T, tau = tid % T;
  const uint32_t frame = blockIdx.x * FPW + fl;
  const bool live = frame < n_frames;
  unsigned char* fbase = smem + (size_t)fl * lay.frame_bytes;

  V mg[16];
  STAMP_DECL;
  front_end<M, FIXED, V, FX>(in, frame, live, tau, fbase, rg, tw, log_lut,
                         reinterpret_cast<uint32_t*>(smem + (size_t)lay.frame_bytes * FPW), mg STAMP_ARG);

  V* mag = reinterpret_cast<V*>(fbase);  // cell x in [-256, N + 256] at slot pad(x + 256)
  V* o1 = reinterpret_cast<V*>(fbase + lay.o1_off);
  V* o2 = reinterpret_cast<V*>(fbase + lay.o2_off);
  uint32_t* det_cnt = reinterpret_cast<uint32_t*>(fbase + lay.det_off);
  uint2* det_stage = reinterpret_cast<uint2*>(fbase + lay.det_off + 8);
  const bool wrap = rg.edge != 0;
  __syncthreads();  // every thread is done reading the FFT image this overlays
  write_mag<M, V>(mag, kHalo, tau, mg, rg.rev_order != 0);
  if (tau == 0) *det_cnt = 0u;
  __syncthreads();
  for (int h = tau; h < 32; h += T) {  // halos: 32 runs of 16 cells, zeros or the wrapped image
    const int x0 = h < 16 ? -kHalo + 16 * h : N + 16 * (h - 16);
    const int src = h < 16 ? x0 + N : x0 - N;
#pragma unroll
    for (int e = 0; e < 16; ++e) mag[pad(x0 + e + kHalo)] = wrap ? mag[pad(src + e + kHalo)] : V(0);
    if (h == 31) mag[pad(N + kHalo + kHalo)] = wrap ? mag[pad(kHalo + kHalo)] : V(0);
  }
  __syncthreads();
  if constexpr (BIG) {
    if (lay.run == 17 && RSP_GOS_SPLIT) gos_stage_split<V, 64, 17>(mag, o1, o2, tau, rg.G, rg.idx_lagg, rg.idx_lead);
    else gos_stage<V, 64>(mag, o1, o2, tau, lay.run, rg.G, rg.idx_lagg, rg.idx_lead);
  } else {
    switch (rg.R) {
      case 4: gos_stage<V, 4>(mag, o1, o2, tau, lay.run, rg.G, rg.idx_lagg, rg.idx_lead); break;
      case 8: gos_stage<V, 8>(mag, o1, o2, tau, lay.run, rg.G, rg.idx_lagg, rg.idx_lead); break;
      case 16: gos_stage<V, 16>(mag, o1, o2, tau, lay.run, rg.G, rg.idx_lagg, rg.idx_lead); break;
      default:
        if (lay.run == 17 && RSP_GOS_SPLIT) gos_stage_split<V, 32, 17>(mag, o1, o2, tau, rg.G, rg.idx_lagg, rg.idx_lead);
        else gos_stage<V, 32>(mag, o1, o2, tau, lay.run, rg.G, rg.idx_lagg, rg.idx_lead);
        break;
    }
  }
  __syncthreads();

  // cell k = tau + T j: lagging statistic = o1[k] (window start k - G - R), leading = o2[k + 2G + R + 1]
  uint32_t word[16];
  {
    constexpr int JS = T + T / 16;
    const V* pl = o1 + pad(tau);
    const V* pr = o2 + pad(tau + 2 * rg.G + rg.R + 1);
    const V* pm = mag + pad(tau + kHalo);
    const int dl = ((tau & 15) == 0) ? 2 : 1, dr = ((tau & 15) == 15) ? 2 : 1;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const V a = pl[JS * j], b = pr[JS * j];
      V stat;
      if (rg.cfar_mode == 0) stat = CfarMath<V>::half_sum(a, b);
      else if (rg.cfar_mode == 1) stat = a > b ? a : b;
      else stat = a < b ? a : b;
      const V cut = pm[JS * j];
      bool group_ok = true;
      if (rg.peak_grouping) group_ok = cut > pm[JS * j - dl] && cut > pm[JS * j + dr];
      word[j] = CfarMath<V>::finish(stat, cut, group_ok, tau + T * j, M, rg);
    }
  }
  emit_words<M, V>(word, out, frame, live, tau, det_cnt, det_stage, fcount, fdet,
                   rg.send_cut ? mag + pad(tau + kHalo) : nullptr, T + T / 16);
}

// ---------------------------------------------------------------- launcher

// GOS kernel LDS: magnitude with 256-cell halos + one or two order-statistic arrays + staging
template <int M>
static GosLayout gos_layout(const ChainRegs& rg) {
  constexpr int N = 1 << M, T = threads_per_frame(M);
  GosLayout l;
  l.run = (N + 2 * rg.G + rg.R + 1 + T - 1) / T;
  const int mag_bytes = 4 * (pad_slots(N + 2 * kHalo) + 2);
  const int o_bytes = 4 * (pad_slots(l.run * T) + 2);
  l.o1_off = mag_bytes;
  l.o2_off = rg.idx_lagg != rg.idx_lead ? l.o1_off + o_bytes : l.o1_off;
  l.det_off = (l.o2_off + o_bytes + 7) & ~7;
  const int total = l.det_off + 8 + 8 * kFrameDetCap;
  l.frame_bytes = ((total > FrameLds<M>::FFT_BYTES ? total : FrameLds<M>::FFT_BYTES) + 15) & ~15;
  return l;
}

// This file is compiled FOUR times (csrc/Makefile): the fp32 kernels, and -- with RSP_PART_FX = 0 / 1 / 2 -- the
// FIXED16 kernels of ONE path of the fixed-point FFT each (FX, fft_lds.hpp: 0 convergent, 1 floor / half-up,
// 2 stage options): four objects built in parallel instead of one 4-minute compile, and no configuration carries
// another path's registers.
#ifdef RSP_PART_FX
constexpr bool kPartFixed = true;
constexpr int kPartFx = RSP_PART_FX;
#else
constexpr bool kPartFixed = false;
constexpr int kPartFx = 0;
#endif
template <typename F>
static hipError_t with_fx(const Chain1dLaunch&, F f) { return f(std::integral_constant<int, kPartFx>{}); }

template <int M>
static hipError_t launch_gos(const Chain1dLaunch& a) {
  const uint32_t fpw = frames_per_wg(M);
  const uint32_t grid = (a.n_frames + fpw - 1) / fpw;
  const GosLayout lay = gos_layout<M>(a.regs);
  const size_t lds = (size_t)lay.frame_bytes * fpw + (kPartFixed ? FrameLds<M>::ROM_BYTES : 0);
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  auto go = [&](auto k, LdsGrant& g) -> hipError_t {
    hipError_t err = grant_lds(k, lds, a.device, g);
    if (err != hipSuccess) return err;
    hipLaunchKernelGGL(k, dim3(grid), dim3(wg_size(M)), lds, a.stream, a.in, a.out, a.n_frames, a.regs, lay,
                       a.twiddles, a.log_lut, a.frame_count, a.frame_det);
    return hipGetLastError();
  };
  return with_fx(a, [&](auto fx) -> hipError_t {
    constexpr int FX = decltype(fx)::value;
    static LdsGrant g2[2];
    return a.regs.R > 32 ? go(chain1d_gos_kernel<M, kPartFixed, true, FX>, g2[0])
                         : go(chain1d_gos_kernel<M, kPartFixed, false, FX>, g2[1]);
  });
}

template <int M>
static hipError_t launch_quad(const Chain1dLaunch& a) {
  const uint32_t fpw = frames_per_wg(M);
  const uint32_t grid = (a.n_frames + fpw - 1) / fpw;
  return with_fx(a, [&](auto fx) -> hipError_t {
    constexpr int FX = decltype(fx)::value;
    auto go = [&](auto small_c, LdsGrant& granted) -> hipError_t {
      constexpr bool SMALL = decltype(small_c)::value;
      const size_t lds = QuadLds<M, SMALL>::BYTES * fpw + (kPartFixed ? QuadLds<M, SMALL>::ROM_BYTES : 0);
      auto k = chain1d_quad_kernel<M, kPartFixed, FX, SMALL>;
      hipError_t e = grant_lds(k, lds, a.device, granted);
      if (e != hipSuccess) return e;
      hipLaunchKernelGGL(k, dim3(grid), dim3(wg_size(M)), lds, a.stream, a.in, a.out, a.n_frames,
                         a.regs, a.twiddles, a.log_lut, a.frame_count, a.frame_det);
      return hipGetLastError();
    };
    static LdsGrant g2[2];
    return a.regs.R + a.regs.G + 4 <= QuadHalo<true>::MAG ? go(std::true_type{}, g2[0]) : go(std::false_type{}, g2[1]);
  });
}

template <int M>
static hipError_t launch_m(const Chain1dLaunch& a) {
  if (a.regs.algorithm == 1) return launch_gos<M>(a);
  if (quad_tail_supports(M, a.regs) && !a.force_generic_tail) return launch_quad<M>(a);
  const uint32_t fpw = frames_per_wg(M);
  const uint32_t grid = (a.n_frames + fpw - 1) / fpw;
  const size_t lds = FrameLds<M>::BYTES * fpw + (kPartFixed ? FrameLds<M>::ROM_BYTES : 0);
  return with_fx(a, [&](auto fx) -> hipError_t {
    constexpr int FX = decltype(fx)::value;
    static LdsGrant granted;
    auto k = chain1d_kernel<M, kPartFixed, FX>;
    hipError_t e = grant_lds(k, lds, a.device, granted);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3(grid), dim3(wg_size(M)), lds, a.stream, a.in, a.out, a.n_frames,
                       a.regs, a.twiddles, a.log_lut, a.frame_count, a.frame_det);
    return hipGetLastError();
  });
}

// one launch (< 4 GiB of input) of this object's data type / FFT path
#define RSP_CAT2(a, b) a##b
#define RSP_CAT(a, b) RSP_CAT2(a, b)
#ifdef RSP_PART_FX
hipError_t RSP_CAT(launch_chain1d_part_fx, RSP_PART_FX)(const Chain1dLaunch& a) {
#else
hipError_t launch_chain1d_part_f32(const Chain1dLaunch& a) {
#endif
  switch (a.log2n) {
    case 8: return launch_m<8>(a);
    case 9: return launch_m<9>(a);
    case 10: return launch_m<10>(a);
    case 11: return launch_m<11>(a);
    case 12: return launch_m<12>(a);
    case 13: return launch_m<13>(a);
    default: return hipErrorInvalidValue;
  }
}

#ifndef RSP_PART_FX  // everything below is data-type independent: compiled once
hipError_t launch_chain1d_part_fx0(const Chain1dLaunch& a);
hipError_t launch_chain1d_part_fx1(const Chain1dLaunch& a);
hipError_t launch_chain1d_part_fx2(const Chain1dLaunch& a);
static hipError_t launch_chain1d_part_fx(const Chain1dLaunch& a) {
  if (a.regs.keep_lsb_mask | a.regs.expand_mask) return launch_chain1d_part_fx2(a);
  return a.regs.trim_conv ? launch_chain1d_part_fx0(a) : launch_chain1d_part_fx1(a);
}

hipError_t launch_chain1d(const Chain1dLaunch& a0) {
  if (a0.n_frames == 0) return hipSuccess;
  if (a0.log2n < kMinLog2N) return launch_chain1d_small(a0);
  // kernels address a launch's input with 32-bit byte offsets: split into < 4 GiB pieces
  const uint64_t beat = a0.fixed ? 4 : 8;
  uint32_t max_frames = (uint32_t)((0xFFFFFFFFull / (beat << a0.log2n)) & ~63ull);
  if (a0.max_frames_per_launch) {  // whole workgroups only, never zero
    const uint32_t want = (a0.max_frames_per_launch + 63u) & ~63u;
    if (want < max_frames) max_frames = want;
  }
  Chain1dLaunch a = a0;
  for (uint32_t done = 0; done < a0.n_frames; done += max_frames) {
    a.n_frames = a0.n_frames - done < max_frames ? a0.n_frames - done : max_frames;
    a.in = static_cast<const char*>(a0.in) + ((uint64_t)done << a0.log2n) * beat;
    a.out = a0.out ? a0.out + (((uint64_t)done << a0.log2n) << (a0.regs.send_cut ? 1 : 0)) : nullptr;
    a.frame_count = a0.frame_count ? a0.frame_count + done : nullptr;
    a.frame_det = a0.frame_det ? a0.frame_det + (uint64_t)done * kFrameDetCap : nullptr;
    const hipError_t e = a.fixed ? launch_chain1d_part_fx(a) : launch_chain1d_part_f32(a);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

// ---------------------------------------------------------------- detection compaction

// Last-workgroup epilogue shared by both compaction kernels: counters = {found, cursor, ticket}, all
// zero on entry.  Thread 0 of every workgroup has added its share with RETURNING device-scope atomics
// (so they have been performed when it goes on) before it takes a ticket; the workgroup that draws the
// last ticket publishes {found, stored} and re-zeroes the counters for the next launch on this stream.
// No __threadfence: only the counters travel between workgroups, and they are only ever touched by
// device-scope atomics -- an agent-scope fence per workgroup writes back / invalidates L2 on this
// multi-XCD part and cost 180 ns per workgroup (750 us for the 4096 workgroups of a 67 M-cell map).
__device__ __forceinline__ void publish_counts(uint32_t* counters, uint32_t cap, uint32_t* d_count) {
  if (threadIdx.x == 0) {
    const uint32_t ticket = atomicAdd(&counters[2], 1u);
    if (ticket == gridDim.x - 1) {
      const uint32_t found = atomicExch(&counters[0], 0u);
      const uint32_t cursor = atomicExch(&counters[1], 0u);
      atomicExch(&counters[2], 0u);
      d_count[0] = found;
      d_count[1] = cursor < cap ? cursor : cap;
    }
  }
}

// found / cursor shares of a workgroup (thread 0 only): returning atomics, see publish_counts
__device__ __forceinline__ uint32_t reserve_block(uint32_t* counters, uint32_t found, uint32_t entries) {
  uint32_t base = 0u;
  if (found) {
    const uint32_t r = atomicAdd(&counters[0], found);
    asm volatile("" ::"v"(r));  // keep the return value live: the wave waits for the atomic
  }
  if (entries) base = atomicAdd(&counters[1], entries);
  return base;
}

// Per-frame slots written by the chain kernels -> one compact list, frames in ascending order.  One thread per frame,
// 256 frames per workgroup.  PREFIX = true (up to kPrefixFrames frames): every workgroup sums the counts of ALL
// frames before its own (a few KiB of L2 reads, all in flight together) instead of reserving its block of the list with
// device-scope atomics -- no counters, no ticket, a deterministic list, and three atomic round trips (~2.5 us of an
// 8 us launch) off the critical path; the last workgroup publishes {found, stored}.  PREFIX = false: a block of the
// list is reserved with one returning atomic per workgroup (blocks land in completion order).
constexpr uint32_t kPrefixFrames = 16384;

template <bool PREFIX>
__global__ void __launch_bounds__(256)
compact_frames_kernel(const uint32_t* __restrict__ fcount, const uint2* __restrict__ fdet,
                      uint32_t n_frames, const uint32_t* __restrict__ words, int log2n, int word_shift,
                      rsp_detection* __restrict__ list, uint32_t cap,
                      uint32_t* __restrict__ counters, uint32_t* __restrict__ d_count) {
  __shared__ uint32_t wave_tot[4], wave_found[4], wave_pre[4], wave_pref[4];
  __shared__ uint32_t base_sh, ovf_n, ovf_cursor;
  __shared__ uint32_t ovf_frame[256], ovf_base[256];
  const uint32_t f = blockIdx.x * 256 + threadIdx.x;
  const uint32_t found = f < n_frames ? fcount[f] : 0u;
  u32x4 early[8];  // slots 0..15 of the frame, requested before the count is known (stale slots are never used)
  {
    const u32x4* src = reinterpret_cast<const u32x4*>(fdet + (size_t)(f < n_frames ? f : 0) * kFrameDetCap);
#pragma unroll
    for (int j = 0; j < 8; ++j) early[j] = src[j];
  }
  // a frame whose peaks did not fit its slots is re-read from the dense words when there are any
  const bool ovf = found > (uint32_t)kFrameDetCap;
  const uint32_t mine = (ovf && !words) ? (uint32_t)kFrameDetCap : found;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) ovf_n = 0u;
  uint32_t pre = 0, pre_found = 0;  // entries / peaks of the frames before this workgroup's (partial sums per thread)
  if constexpr (PREFIX) {
    const uint32_t before = blockIdx.x * 256u;  // a multiple of 4, all of them < n_frames
    const u32x4* fc4 = reinterpret_cast<const u32x4*>(fcount);
    for (uint32_t i = threadIdx.x; i < before / 4; i += 256) {
      const u32x4 c = fc4[i];
      pre_found += (c.x + c.y) + (c.z + c.w);
      if (words) pre += (c.x + c.y) + (c.z + c.w);
      else pre += min(c.x, (uint32_t)kFrameDetCap) + min(c.y, (uint32_t)kFrameDetCap) + min(c.z, (uint32_t)kFrameDetCap) + min(c.w, (uint32_t)kFrameDetCap);
    }
  }
  uint32_t inc = mine, tot_found = found;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t t = __shfl_up(inc, d);
    if (lane >= d) inc += t;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    tot_found += __shfl_xor(tot_found, d);
    if constexpr (PREFIX) {
      pre += __shfl_xor(pre, d);
      pre_found += __shfl_xor(pre_found, d);
    }
  }
  if (lane == 63) wave_tot[wave] = inc;
  if (lane == 0) {
    wave_found[wave] = tot_found;
    wave_pre[wave] = pre;
    wave_pref[wave] = pre_found;
  }
  __syncthreads();
  uint32_t off = inc - mine;
  for (int w = 0; w < wave; ++w) off += wave_tot[w];
  const uint32_t tot = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
  const uint32_t fnd = wave_found[0] + wave_found[1] + wave_found[2] + wave_found[3];
  uint32_t base;
  if constexpr (PREFIX) {
    const uint32_t wg_base = wave_pre[0] + wave_pre[1] + wave_pre[2] + wave_pre[3];
    base = wg_base + off;
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
      const uint32_t cursor = wg_base + tot;
      d_count[0] = wave_pref[0] + wave_pref[1] + wave_pref[2] + wave_pref[3] + fnd;
      d_count[1] = cursor < cap ? cursor : cap;
    }
  } else {
    if (threadIdx.x == 0) base_sh = reserve_block(counters, fnd, tot);
    __syncthreads();
    base = base_sh + off;
  }
  if (ovf && words) {
    const uint32_t s = atomicAdd(&ovf_n, 1u);
    ovf_frame[s] = f;
    ovf_base[s] = base;
  } else {
    // the first 16 slots were requested together with the count (one memory round trip for all but ~1e-4 of the
    // frames at 5 peaks per frame); later slots four per round (two 16-byte loads)
    auto put = [&](uint32_t i, const u32x4& e01, const u32x4& e23) {
      const uint32_t bins[4] = {e01.x, e01.z, e23.x, e23.z}, wds[4] = {e01.y, e01.w, e23.y, e23.w};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (i + q < mine && base + i + q < cap) {
          rsp_detection d;
          d.frame = f;
          d.bin = bins[q];
          d.doppler = 0;
          d.word = wds[q];
          list[base + i + q] = d;
        }
      }
    };
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (4u * r < mine) put(4u * r, early[2 * r], early[2 * r + 1]);
    const u32x4* src = reinterpret_cast<const u32x4*>(fdet + (size_t)f * kFrameDetCap);
    for (uint32_t i = 16; i < mine; i += 4) put(i, src[i / 2], (i + 2 < mine) ? src[i / 2 + 1] : u32x4{0u, 0u, 0u, 0u});
  }
  __syncthreads();
  // overflow frames (rare: > kFrameDetCap peaks in one frame): the whole workgroup re-reads the frame
  const uint32_t n_ovf = ovf_n;
  for (uint32_t q = 0; q < n_ovf; ++q) {
    if (threadIdx.x == 0) ovf_cursor = 0u;
    __syncthreads();
    const uint32_t of = ovf_frame[q], ob = ovf_base[q];
    const uint32_t* row = words + (((size_t)of << log2n) << word_shift);  // word_shift = 1: 64-bit beats {word, cut}
    for (uint32_t x = threadIdx.x; x < (1u << log2n); x += 256) {
      const uint32_t w = row[(size_t)x << word_shift];
      if (w & 1u) {
        const uint32_t slot = ob + atomicAdd(&ovf_cursor, 1u);
        if (slot < cap) {
          rsp_detection d;
          d.frame = of;
          d.bin = x;
          d.doppler = 0;
          d.word = w;
          list[slot] = d;
        }
      }
    }
    __syncthreads();
  }
  if constexpr (!PREFIX) publish_counts(counters, cap, d_count);
}

hipError_t launch_compact_frames(const uint32_t* fcount, const uint2* fdet, uint32_t n_frames,
                                 const uint32_t* words, int log2n, int word_shift, rsp_detection* list,
                                 uint32_t cap, uint32_t* counters, uint32_t* d_count, hipStream_t stream) {
  if (n_frames == 0) return hipMemsetAsync(d_count, 0, 2 * sizeof(uint32_t), stream);
  if (n_frames <= kPrefixFrames)
    hipLaunchKernelGGL(compact_frames_kernel<true>, dim3((n_frames + 255) / 256), dim3(256), 0, stream,
                       fcount, fdet, n_frames, words, log2n, word_shift, list, cap, counters, d_count);
  else
    hipLaunchKernelGGL(compact_frames_kernel<false>, dim3((n_frames + 255) / 256), dim3(256), 0, stream,
                       fcount, fdet, n_frames, words, log2n, word_shift, list, cap, counters, d_count);
  return hipGetLastError();
}

// Dense words -> compact list of the peak cells (word bit 0, Tester:165).  One pass: a workgroup
// owns 16 384 consecutive cells, every thread loads its 64 words with four 16-byte loads issued
// together (16 KiB in flight per workgroup: the kernel runs at streaming rate, 4 B per cell), counts
// its peaks, the workgroup reserves a block of the list with ONE global atomic and the (rare) peaks
// are written from registers.
constexpr int kCompactCellsPerWg = 16384;

__global__ void __launch_bounds__(256)
compact_kernel(const uint32_t* __restrict__ words, uint64_t n_cells, uint32_t log2_row,
               uint32_t log2_rows_per_frame, uint32_t word_shift, rsp_detection* __restrict__ list, uint32_t cap,
               uint32_t* __restrict__ counters, uint32_t* __restrict__ d_count) {
  // word_shift = 1: 64-bit beats {word, cut} (sendCut): n_cells counts 32-bit WORDS, every other one is a cut
  const uint32_t odd = word_shift ? 0u : 1u;
  __shared__ uint32_t wave_cnt[4];
  __shared__ uint32_t base_sh;
  const uint64_t lo = (uint64_t)blockIdx.x * kCompactCellsPerWg;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // thread t, load j covers cells lo + 1024 j + 4 t .. + 3 (n_cells is a multiple of 4: whole rows of >= 16 cells)
  u32x4 w[16];
  uint32_t mine = 0;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const uint64_t c = lo + 1024u * j + 4u * threadIdx.x;
    w[j] = c < n_cells ? *reinterpret_cast<const u32x4*>(words + c) : u32x4{0u, 0u, 0u, 0u};
  }
#pragma unroll
  for (int j = 0; j < 16; ++j) mine += (w[j].x & 1u) + (w[j].y & odd) + (w[j].z & 1u) + (w[j].w & odd);
  uint32_t inc = mine;  // inclusive scan over the wave
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t t = __shfl_up(inc, d);
    if (lane >= d) inc += t;
  }
  if (lane == 63) wave_cnt[wave] = inc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t t = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    base_sh = reserve_block(counters, t, t);
  }
  __syncthreads();
  if (mine) {  // rare: the thread's words are read again (L2 hits) by a compact loop instead of keeping 64 registers live
    uint32_t slot = base_sh + inc - mine;
    for (int w0 = 0; w0 < wave; ++w0) slot += wave_cnt[w0];
#pragma unroll 1
    for (int j = 0; j < 16; ++j) {
      const uint64_t c = lo + 1024u * j + 4u * threadIdx.x;
      if (c >= n_cells) break;
      const u32x4 v4 = *reinterpret_cast<const u32x4*>(words + c);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint32_t v = v4[q];
        if ((v & 1u) && ((q & 1) == 0 || odd)) {
          if (slot < cap) {
            const uint64_t i = (c + q) >> word_shift;
            rsp_detection d;
            d.bin = (uint32_t)(i & ((1ull << log2_row) - 1ull));
            const uint64_t row = i >> log2_row;
            d.doppler = (uint32_t)(row & ((1ull << log2_rows_per_frame) - 1ull));
            d.frame = (uint32_t)(row >> log2_rows_per_frame);
            d.word = v;
            list[slot] = d;
          }
          ++slot;
        }
      }
    }
  }
}

// {found, cursor} -> d_count, counters re-zeroed: one thread, its own launch.  The dense compaction runs
// thousands of workgroups; a ticket per workgroup on ONE address (publish_counts) serialises at the
// memory side (~13 ns each) and cost more than this ~2 us launch.
__global__ void compact_finalize_kernel(uint32_t* __restrict__ counters, uint32_t cap, uint32_t* __restrict__ d_count,
                                        bool use_found) {
  const uint32_t found = atomicExch(&counters[0], 0u);
  uint32_t cursor = atomicExch(&counters[1], 0u);
  if (use_found) cursor = found;  // lists appended peak by peak (2-D CFAR kernels): every peak found was offered a slot
  d_count[0] = found;
  d_count[1] = cursor < cap ? cursor : cap;
}

hipError_t launch_compact_finalize(uint32_t* counters, uint32_t cap, uint32_t* d_count, bool stored_is_found,
                                   hipStream_t stream) {
  hipLaunchKernelGGL(compact_finalize_kernel, dim3(1), dim3(1), 0, stream, counters, cap, d_count, stored_is_found);
  return hipGetLastError();
}

hipError_t launch_compact(const uint32_t* words, uint64_t n_cells, uint32_t log2_row,
                          uint32_t log2_rows_per_frame, uint32_t word_shift, rsp_detection* list, uint32_t cap,
                          uint32_t* counters, uint32_t* d_count, hipStream_t stream) {
  if (n_cells == 0) return hipMemsetAsync(d_count, 0, 2 * sizeof(uint32_t), stream);
  const uint64_t n_words = n_cells << word_shift;
  const uint64_t blocks = (n_words + kCompactCellsPerWg - 1) / kCompactCellsPerWg;
  hipLaunchKernelGGL(compact_kernel, dim3((uint32_t)blocks), dim3(256), 0, stream, words, n_words,
                     log2_row, log2_rows_per_frame, word_shift, list, cap, counters, d_count);
  hipLaunchKernelGGL(compact_finalize_kernel, dim3(1), dim3(1), 0, stream, counters, cap, d_count, false);
  return hipGetLastError();
}

#endif  // RSP_PART_FX

}  // namespace rsp
