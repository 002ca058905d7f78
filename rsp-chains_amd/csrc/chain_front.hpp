// Shared front end of the fused 1-D chain kernels: frame -> FFT -> magnitude in registers, the
// magnitude image in LDS, and the dense-word / detection-slot epilogue of the per-cell tails.
//
// Replaces the stream wiring
//   cfar.streamNode := AXI4StreamBuffer() := mag.streamNode := AXI4StreamBuffer() := fft.streamNode
// (/root/reference/src/main/scala/FftMagCfarChain.scala:47): a frame enters as 2^M beats from HBM,
// stays in LDS through all three blocks and leaves as 2^M 32-bit words
// (FftMagCfarChainTester.scala:145-151,163-167).  Algorithmic HBM traffic: F32 8 B in + 4 B out per
// cell; FIXED16 4 B in + 4 B out.  Workgroup = frames_per_wg(M) frames x (2^M / 16) threads, 16 cells
// per thread.  The CFAR tails: cfar_quad.hpp (CA / GO / SO on quads, the default), cfar_cell.hpp
// (per-cell tail: CASH, window sizes that are not multiples of 4), cfar_gos.hpp (ordered statistic).
#pragma once
#include <hip/hip_runtime.h>
#include <float.h>

#include "chain_regs.hpp"
#include "fft_lds.hpp"
#include "kernels.hpp"
#include "side_build.hpp"

namespace rsp {

// ---------------------------------------------------------------- LDS layout per frame
// After the FFT the frame's LDS is re-used for the CFAR working set (4-byte slots):
//   mag : cell x in [-16, N+16)   at slot pad(x + 16)    (1-cell halo for peak grouping)
//   pb  : cell x in [-256, N+256] at slot pad(x + 256)   block-relative exclusive prefix
//   bs  : block b in [-16, N/16 + 16) at slot b + 16     block totals (blocks of 256 cells: only b in [-1, N/256];
//                                                        blocks of 16 cells for fp32 windows of at most 16 cells)
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
  static constexpr int BS_HALO = 16;            // block totals kept on either side of the frame (16-cell blocks: 256 cells)
  static constexpr int BS_SLOTS = N / 16 + 2 * BS_HALO + 1;
  static constexpr int MAG_OFF = 0;
  static constexpr int PB_OFF = MAG_OFF + 4 * MAG_SLOTS;
  static constexpr int BS_OFF = PB_OFF + 4 * PB_SLOTS;
  static constexpr int DET_OFF = (BS_OFF + 4 * BS_SLOTS + 7) & ~7;
  static constexpr int CFAR_BYTES = DET_OFF + 8 + 8 * kFrameDetCap;
  static constexpr int FFT_BYTES = 8 * PADN;             // f32x2 per slot (FIXED16 uses 4 B)
  static constexpr int BYTES = ((CFAR_BYTES > FFT_BYTES ? CFAR_BYTES : FFT_BYTES) + 15) & ~15;
  static constexpr int ROM_BYTES = fx_rom_bytes(M);  // FIXED16: LDS copy of the Q2.14 twiddle ROM ({W1, W2} per twiddle: fx_rom_entry), per workgroup
};

// FIXED16: where the twiddle ROM of a chain workgroup lives.  The ROM is read by the FFT only, and the packed exchange
// image of a frame fills the lower half of its region only (4-byte slots), so the ROM OVERLAYS the upper part of the last
// frame's region: every chain kernel's tail starts with a barrier ("done reading the FFT image") that also ends all ROM
// reads before a tail image is written over it.  4096 points, quad kernel: 34.9 KiB per workgroup instead of 34.9 + 17.5 =
// four workgroups per CU instead of three; 8192 points: two instead of one (the GOS kernel at the cfg-4 shape on FIXED16:
// 147 -> 104 us).  The stage-option path (FX = 2) exchanges 8-byte slots that fill the region: its ROM stays behind the frames.
// frame_bytes = the kernel's LDS per frame (a multiple of 16).
template <int M, int FX>
struct FixedRom {
  static constexpr int FPW = frames_per_wg(M);
  static constexpr int IMG = (4 * fft_image_slots(M) + 7) & ~7;  // one frame's packed exchange image
  static constexpr int ROM = fx_rom_bytes(M);
  static constexpr bool OVERLAY = FX != 2;
  static __host__ __device__ constexpr int off(int frame_bytes) {
    const int low = (FPW - 1) * frame_bytes + IMG, high = FPW * frame_bytes - ROM;
    return OVERLAY ? (low > high ? low : high) : FPW * frame_bytes;
  }
  static __host__ __device__ constexpr int total(int frame_bytes) {
    return off(frame_bytes) + ROM > FPW * frame_bytes ? off(frame_bytes) + ROM : FPW * frame_bytes;
  }
};

__device__ __forceinline__ int mag_slot(int x) { return pad(x + 16); }
__device__ __forceinline__ int pb_slot(int x) { return pad(x + kHalo); }

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

// ---------------------------------------------------------------- shared front end
// F32 magnitudes of the thread's 16 bins (logMagMux, mode select = MAG CSR 0).  The mode switch is hoisted
// out of the per-bin loop (a uniform branch per bin costs ~15 SALU each).  JPL and squared magnitude are
// homogeneous, so the power-of-two 1/N scale (net 1/N: FftMagCfarChainTester.scala:77) is applied to the
// magnitude -- bit-identical to scaling the spectrum first -- and two bins share every packed op.
template <int M, typename Hooks>
__device__ __forceinline__ void magnitudes_f32(const f32x2 (&x)[16], int mag_mode, float (&mg)[16], Hooks& hk) {
  const float scale = 1.0f / (float)(1 << M);
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
  if constexpr (kCountPath) {
    jpl_pairs();
  } else if (hk.off(4)) {
#pragma unroll
    for (int e = 0; e < 16; ++e) mg[e] = x[e].x;
  } else if (mag_mode == 2) {
    jpl_pairs();
  } else if (mag_mode == 0) {
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
}

// Frame -> FFT (passes through LDS at fbase) -> magnitude of this thread's 16 bins in registers:
// mg[g * 2^WL + p] is bin (bitrev(p) << (M - WL)) | bitrev(g T + tau).
template <int M, bool FIXED, typename V, int FX, typename Hooks>
__device__ __forceinline__ void front_end(const void* __restrict__ in, uint32_t frame, bool live, int tau,
                                          unsigned char* fbase, const ChainRegs& rg,
                                          const void* __restrict__ tw,
                                          const int16_t* __restrict__ log_lut, uint2* rom,
                                          V (&mg)[16], Hooks& hk) {
  constexpr int N = 1 << M;
  constexpr int W = plan_w(M, 0), LO = plan_lo(M, 0);
  // uniform base (SGPR pair) + one 32-bit per-thread byte offset; the per-register part is a
  // compile-time constant (the launcher keeps one launch's input below 4 GiB).
  // A dead frame (ragged last workgroup) re-reads frame 0 and never stores.
  const char* gbase = reinterpret_cast<const char*>(in);
  if constexpr (!FIXED) {
    f32x2 x[16];
    TwAll<M> twb;
    const uint32_t voff = ((live ? frame : 0u) * (uint32_t)N + (uint32_t)elem_index<M, LO, W>(tau, 0)) * 8u;
    wave_prio(1);
    fft_f32_load<M>([&](int d) { return stream_load(reinterpret_cast<const f32x2*>(gbase + (size_t)voff + (size_t)d * 8u)); },
                    tau, reinterpret_cast<const f32x2*>(tw), twb, x);
    wave_prio(0);
    if (rg.window) {  // pre-FFT window (build extension): one fp32 coefficient per sample
      const float* wt = reinterpret_cast<const float*>(rg.window) + elem_index<M, LO, W>(tau, 0);
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float wv = wt[sample_offset<M>(e)];
        x[e] = x[e] * f32x2{wv, wv};
      }
    }
    fft_f32_passes<M>(tau, reinterpret_cast<f32x2*>(fbase), twb, x, hk);
    magnitudes_f32<M>(x, rg.mag_mode, mg, hk);
  } else {
    const uint32_t* twq = reinterpret_cast<const uint32_t*>(tw);
    const uint32_t voff = ((live ? frame : 0u) * (uint32_t)N + (uint32_t)elem_index<M, LO, W>(tau, 0)) * 4u;
    if constexpr (FX == 0 || FX == 1) {
      // no stage option: the beats stay packed {re[31:16], im[15:0]} (RspChainTesterUtils.scala:105-109) from HBM to the magnitude
      uint32_t z[16];
#pragma unroll
      for (int e = 0; e < 16; ++e)
        z[e] = stream_load(reinterpret_cast<const uint32_t*>(gbase + (size_t)voff + (size_t)sample_offset<M>(e) * 4u));
      if (rg.window) {  // Q1.15 coefficient, product rounded half-up back to 16 bits (spec section 2.1)
        const int16_t* wt = reinterpret_cast<const int16_t*>(rg.window) + elem_index<M, LO, W>(tau, 0);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int wq = wt[sample_offset<M>(e)];
          const int re = ((int)(short)(z[e] >> 16) * wq + (1 << 14)) >> 15, im = ((int)(short)(z[e] & 0xffffu) * wq + (1 << 14)) >> 15;
          z[e] = ((uint32_t)re << 16) | ((uint32_t)im & 0xffffu);
        }
      }
      // twiddle ROM -> LDS once per workgroup (the sample loads above are already in flight)
      fx_rom_fill(rom, twq, N / 2, threadIdx.x, wg_size(M));
      __syncthreads();
      fft_fx_frame_pk<M, FX == 0>(z, tau, fbase, rom, rg, hk);
      if (hk.off(4)) {
#pragma unroll
        for (int e = 0; e < 16; ++e) mg[e] = (int)(z[e] >> 16) & 32767;
      } else if (rg.mag_mode == 2) {
#pragma unroll
        for (int e = 0; e < 16; e += 2) jpl_fx_pair(z[e], z[e + 1], mg[e], mg[e + 1]);
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) mg[e] = mag_fx((int)(short)(z[e] >> 16), (int)(short)(z[e] & 0xffffu), rg, log_lut);
      }
    } else {
      int xr[16], xi[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const uint32_t b = stream_load(reinterpret_cast<const uint32_t*>(gbase + (size_t)voff + (size_t)sample_offset<M>(e) * 4u));
        xr[e] = (int)(short)(b >> 16);
        xi[e] = (int)(short)(b & 0xffffu);
      }
      if (rg.window) {
        const int16_t* wt = reinterpret_cast<const int16_t*>(rg.window) + elem_index<M, LO, W>(tau, 0);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int wq = wt[sample_offset<M>(e)];
          xr[e] = (int)(short)((xr[e] * wq + (1 << 14)) >> 15);
          xi[e] = (int)(short)((xi[e] * wq + (1 << 14)) >> 15);
        }
      }
      fx_rom_fill(rom, twq, N / 2, threadIdx.x, wg_size(M));
      __syncthreads();
      fft_fx_frame<M, FX>(xr, xi, tau, fbase, rom, rg, hk);
      if (hk.off(4)) {
#pragma unroll
        for (int e = 0; e < 16; ++e) mg[e] = xr[e] & 32767;
      } else if (rg.mag_mode == 2) {
#pragma unroll
        for (int e = 0; e < 16; ++e) mg[e] = jpl_fx(xr[e], xi[e]);
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) mg[e] = mag_fx(xr[e], xi[e], rg, log_lut);
      }
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
    for (int j = 0; j < 16; ++j) {
      const u32x2 beat = {word[j], __builtin_bit_cast(uint32_t, cut_lds[cut_stride * j])};
      stream_store(beat, reinterpret_cast<u32x2*>(obase + T * j));
    }
  } else if (live && out) {
    char* obase = reinterpret_cast<char*>(out);
    const uint32_t ooff = (frame * (uint32_t)N + (uint32_t)tau) * 4u;
#pragma unroll
    for (int j = 0; j < 16; ++j)
      stream_store(word[j], reinterpret_cast<uint32_t*>(obase + (size_t)ooff + (size_t)(T * j) * 4u));
  }
  if (!kCountPath && fcount) {
    uint32_t hits = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) hits |= (word[j] & 1u) << j;
    if (hits) {  // rare: ~1 peak per 1000 cells; one LDS atomic per thread with peaks
      uint32_t slot = atomicAdd(det_cnt, (uint32_t)__popc(hits));
      while (hits) {
        const int j = __ffs(hits) - 1;
        hits &= hits - 1;
        uint32_t w = word[0];
#pragma unroll
        for (int q = 1; q < 16; ++q) w = (j == q) ? word[q] : w;
        if (slot < (uint32_t)kFrameDetCap) det_stage[slot] = make_uint2((uint32_t)(tau + T * j), w);
        ++slot;
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

__device__ __forceinline__ uint32_t bits_of(float v) { return __float_as_uint(v); }
__device__ __forceinline__ uint32_t bits_of(int v) { return (uint32_t)v; }

template <typename V> struct Vec4;
template <> struct Vec4<float> { typedef float type __attribute__((ext_vector_type(4))); };
template <> struct Vec4<int> { typedef int type __attribute__((ext_vector_type(4))); };

}  // namespace rsp
