// 2-D range-Doppler chain: range FFT -> Doppler FFT -> magnitude -> 2-D CA-CFAR.
//
// No reference counterpart: every chain under /root/reference/src/main/scala holds ONE 1-D FFT
// (SURVEY F5); this is BASELINE.json configs[2] / configs[4], specified by
// oracle/rsp_oracle.c::orc_rd_f32in.  A channel's map is [doppler d][range r], row-major.
//
//   range_fft_kernel   rows: FFT over r (contiguous in), complex out in tiles of 16 range bins  (8 R + 8 W)
//   doppler_mag_kernel columns: FFT over d for 8 / 16 adjacent range bins per workgroup = one contiguous
//                      block of the tiled map; |.| out in the same tiling                     (8 R + 4 W)
//   cfar2d_walk_kernel (windows of cfg 3 / 5, compile-time) one wave walks a 128-column strip down
//                      the Doppler axis: register ring of rows, running column sums, cross-lane window
//                      sums for the row direction; no LDS memory, no barriers                  (4 R + 4 W)
//   cfar2d_kernel      (any run-time windows) tile + halo in LDS, separable sliding box sums
// = 36 B/cell against the 28 B/cell a fully fused Doppler+CFAR pass would need.  That fusion was designed
// and dropped: the CFAR halo crosses workgroup tiles in range, so the fused workgroup has to
// keep 2 (ref + guard) + 1 = 21 magnitude columns x all Doppler bins next to its FFT image: 64 KiB + 32 KiB at
// 512 Doppler bins (one workgroup per CU), 128 KiB + 64 KiB at 1024 (does not fit 160 KiB).
#include <hip/hip_runtime.h>
#include <float.h>

#include "chain_regs.hpp"
#include "fft_lds.hpp"
#include "kernels.hpp"

namespace rsp {

__device__ __forceinline__ float mag2d(f32x2 z, int mode) {
  const float ar = fabsf(z.x), ai = fabsf(z.y);
  const float u = fmaxf(ar, ai), v = fminf(ar, ai);
  const float jpl = fmaxf(u + v * 0.125f, u * 0.875f + v * 0.5f);  // RspChainTesterUtils.scala:120-127
  if (mode == 2) return jpl;
  if (mode == 0) return z.x * z.x + z.y * z.y;
  return __log2f(fmaxf(jpl, FLT_MIN));
}

// Layout of the two intermediate maps (range spectra, magnitudes) of one channel, chosen per launch (rd_tile()):
// row-major [d][r] up to 4096 range bins; at 8192 the Doppler pass would walk the map at a 64-KiB pitch (HBM banks and
// channels see one power-of-two stride: 264 us at 8 x 8192 x 1024), so the maps are kept in tiles of 16 range bins,
// [r / 16][d][r % 16]: every Doppler workgroup reads ONE contiguous block (190 us), the range pass still writes whole
// 128-B lines (+8 us) and the CFAR walker reads 64-B segments of sequential streams (+5 us).  At 4096 x 512 the tiling
// costs 4 us more than it saves.  RSP_RD_TILE=0 / 16 force one layout for A/B runs.
constexpr uint32_t kTileCols = 16;
#ifndef RSP_RD_MAGTILE
#define RSP_RD_MAGTILE 16
#endif
constexpr uint32_t kMagTileCols = RSP_RD_MAGTILE;  // tile width of the magnitude map
template <uint32_t W = kTileCols>
__host__ __device__ __forceinline__ size_t map_index(uint32_t d, uint32_t r, uint32_t nd, uint32_t nr, uint32_t tile) {
  return tile ? ((size_t)(r / W) * nd + d) * W + (r % W) : (size_t)d * nr + r;
}
static inline uint32_t rd_tile(int log2nr) {
#ifdef RSP_RD_TILE
  return (RSP_RD_TILE) ? kTileCols : 0u;
#else
  return log2nr >= 13 ? kTileCols : 0u;
#endif
}

// ---------------------------------------------------------------- range pass (rows)
// Streaming hints of the 2-D chain, one bit per access kind (side builds A/B them: tools/build_variant_rd.sh x.so
// -DRSP_RD_NT=mask): 1 range-pass loads, 2 its spectrum stores, 4 Doppler-pass loads, 8 its magnitude stores,
// 16 the walker's row loads.  Measured at 8 x 4096 x 512 (base 119-120 us per batch): 1 -> 115-116, 2 -> 129, 4 -> 116-118,
// 8 -> 119, 16 -> 120-121, the walker's word stores (its own run-time switch, below) -> 113.6-114.5; the gains do not add
// (1 + 4 + words: 114.0), and at 8 x 8192 x 1024 every one of them is neutral or slower -- the product keeps only the
// word stores' hint, per launch.
#ifndef RSP_RD_NT
#define RSP_RD_NT 0
#endif
template <int BIT, typename T>
__device__ __forceinline__ T rd_load(const T* p) {
  if constexpr (((RSP_RD_NT) & BIT) != 0) return __builtin_nontemporal_load(p);
  else return *p;
}
template <int BIT, typename T>
__device__ __forceinline__ void rd_store(T v, T* p) {
  if constexpr (((RSP_RD_NT) & BIT) != 0) __builtin_nontemporal_store(v, p);
  else *p = v;
}

// TILED is a template parameter: with the row-major layout the 16 stores of a thread are base + compile-time offsets
// again (a run-time element step cost 7 us of address arithmetic per 16.7 M cells)
template <int M, bool TILED>
__global__ void __launch_bounds__(wg_size(M))
range_fft_kernel(const f32x2* __restrict__ in, f32x2* __restrict__ out, uint32_t n_rows, uint32_t log2nd,
                 const f32x2* __restrict__ tw, const float* __restrict__ win, uint32_t* __restrict__ zero_count) {
  constexpr int N = 1 << M, T = threads_per_frame(M), FPW = frames_per_wg(M);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, fl = tid / T, tau = tid % T;
  const uint32_t row = blockIdx.x * FPW + fl;
  const bool live = row < n_rows;
  if (zero_count && blockIdx.x == 0 && tid == 0) zero_count[0] = zero_count[1] = 0u;  // the detection list's {found, stored}
  f32x2* buf = reinterpret_cast<f32x2*>(smem) + (size_t)fl * fft_image_slots(M);
#ifdef RSP_ABL_RANGE_L2  // ablation builds: every workgroup reads rows 0..63 (L2-resident source)
  const f32x2* src = in + (size_t)(live ? (row & 63) : 0) * N + first_sample<M>(tau);
#else
  const f32x2* src = in + (size_t)(live ? row : 0) * N + first_sample<M>(tau);
#endif
  f32x2 x[16];
  if (win) {  // fast-time window (build extension)
    const float* wsrc = win + first_sample<M>(tau);
    fft_f32_frame<M>([&](int d) { const float wv = wsrc[d]; return rd_load<1>(src + d) * f32x2{wv, wv}; }, tau, buf, tw, x);
  } else {
    fft_f32_frame<M>([&](int d) { return rd_load<1>(src + d); }, tau, buf, tw, x);
  }
  // register (g, p) holds bin (bitrev(p) << (M - WL)) | (g T + tau) (fft_lds.hpp, last pass): consecutive lanes hold
  // consecutive bins, so the spectrum goes straight to HBM in natural order, 512 B per wave-instruction
  constexpr int NP = plan_np(M), WL = plan_w(M, NP - 1);
  const float scale = 1.0f / (float)N;
  if (!live) return;
#ifdef RSP_ABL_RANGE_NOSTORE
  if (log2nd != 77u) return;
#endif
  if constexpr (!TILED) {
    f32x2* dst = out + (size_t)row * N + tau;
#pragma unroll
    for (int g = 0; g < (16 >> WL); ++g) {
#pragma unroll
      for (int p = 0; p < (1 << WL); ++p) rd_store<2>(x[g * (1 << WL) + p] * scale, dst + ((bitrev_c(p, WL) << (M - WL)) + g * T));
    }
  } else {
    // bins c + tau with c a multiple of T >= 16: the tile index splits into a per-lane and a per-register part
    static_assert(T % kTileCols == 0, "a register's bins start on a tile boundary");
    const uint32_t nd = 1u << log2nd, ch = row >> log2nd, d = row & (nd - 1);
    f32x2* dst = out + (((size_t)ch << log2nd) << M) + map_index(d, tau, nd, N, kTileCols);
#pragma unroll
    for (int g = 0; g < (16 >> WL); ++g) {
#pragma unroll
      for (int p = 0; p < (1 << WL); ++p)
        rd_store<2>(x[g * (1 << WL) + p] * scale, dst + ((size_t)((bitrev_c(p, WL) << (M - WL)) + g * T) << log2nd));
    }
  }
}

// ---------------------------------------------------------------- Doppler pass (columns) + magnitude
#ifndef RSP_DOPPLER_COLS10
#define RSP_DOPPLER_COLS10 8
#endif
#ifndef RSP_DOPPLER_COLS9
#define RSP_DOPPLER_COLS9 16
#endif
#ifndef RSP_DOPPLER_XCDMAP
#define RSP_DOPPLER_XCDMAP (C * 8 < 128)
#endif
constexpr int kColsPerWg(int MD) { return MD >= 10 ? (RSP_DOPPLER_COLS10) : MD == 9 ? (RSP_DOPPLER_COLS9) : 16; }
// LDS bytes per column: the padded FFT image + a skew that spreads the columns of a wave over the banks.  The lanes
// of a ds_write_b64's 16-lane group (ds_read_b64: 32-lane group) are the workgroup's columns at one position, so the
// column pitch decides the conflicts: with 8 columns, images 8 banks apart (+32 B) are conflict-free; with 16 columns
// that pitch wraps the banks twice -- 5632 LDS cycles per wave and frame at 512 Doppler bins against 2304 with +8 B
// (2 banks apart; tools/lds_sim.py model, SQ_LDS_BANK_CONFLICT 12.6 M -> per launch confirmed it).
constexpr int kColBytes(int MD) { return 8 * fft_image_slots(MD) + (kColsPerWg(MD) >= 16 ? 8 : 32); }

template <int MD>
__global__ void __launch_bounds__(threads_per_frame(MD) * kColsPerWg(MD))
doppler_mag_kernel(const f32x2* __restrict__ in, float* __restrict__ mag, uint32_t n_ch, uint32_t nr, uint32_t tile,
                   int mag_mode, const f32x2* __restrict__ tw, const float* __restrict__ win) {
  constexpr int ND = 1 << MD, C = kColsPerWg(MD);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // lanes run over the C adjacent range bins first: a wave touches 64 / C rows x (C x 8 B) segments
  const int tid = threadIdx.x, fl = tid % C, tau = tid / C;
  const uint32_t tiles_per_ch = nr / C;  // column groups of C range bins
  // workgroups are dealt round-robin to the 8 XCDs: give each XCD a contiguous run of column tiles, so
  // that the two 8-column tiles sharing a 128-B line meet in the same L2 (-25 % at 1024 Doppler bins;
  // with 16 columns a tile reads whole lines and the plain order is the faster one)
  const uint32_t wg = ((RSP_DOPPLER_XCDMAP) && gridDim.x % 8 == 0) ? (blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8
                                                                   : blockIdx.x;
  const uint32_t ch = wg / tiles_per_ch, r0 = (wg % tiles_per_ch) * C;
  f32x2* buf = reinterpret_cast<f32x2*>(smem + (size_t)fl * kColBytes(MD));
  // element index of (ch, d = 0, r) and the distance between Doppler rows, in each of the two maps
  const size_t col = (size_t)ch * ND * nr + map_index(0, r0 + fl, ND, nr, tile);
  const size_t mcol = (size_t)ch * ND * nr + map_index<kMagTileCols>(0, r0 + fl, ND, nr, tile);
  const uint32_t pitch = tile ? kTileCols : nr, mpitch = tile ? kMagTileCols : nr;
  const f32x2* src = in + col + (size_t)first_sample<MD>(tau) * pitch;
  f32x2 x[16];
  if (win) {  // slow-time window
    const float* wsrc = win + first_sample<MD>(tau);
    fft_f32_frame<MD>([&](int d) { const float wv = wsrc[d]; return rd_load<4>(src + (size_t)d * pitch) * f32x2{wv, wv}; }, tau, buf, tw, x);
  } else {
#ifdef RSP_ABL_DOP_L2  // ablation builds: rows 0..15 only (L2-resident source)
    fft_f32_frame<MD>([&](int d) { return src[(size_t)(d & 15) * pitch]; }, tau, buf, tw, x);
#else
    fft_f32_frame<MD>([&](int d) { return rd_load<4>(src + (size_t)d * pitch); }, tau, buf, tw, x);
#endif
  }
  constexpr int NP = plan_np(MD), WL = plan_w(MD, NP - 1);
  const float scale = 1.0f / (float)ND;
  float* dst = mag + mcol;
#ifdef RSP_ABL_DOP_NOSTORE
  if (mag_mode != 77) dst = nullptr;
  if (mag_mode == 77)
#endif
#pragma unroll
  for (int g = 0; g < (16 >> WL); ++g) {
#pragma unroll
    for (int p = 0; p < (1 << WL); ++p)
      rd_store<8>(mag2d(x[g * (1 << WL) + p] * scale, mag_mode), dst + (size_t)bin_of<MD>(tau, g, p) * mpitch);
  }
}

// 1024 Doppler bins: only 8 column images fit the LDS of a workgroup, and 8 columns are half a 128-B line of either
// map.  This form handles 16 columns with the same 8 images: every thread loads column PAIRS (16-byte loads: whole lines
// per 8 lanes), transforms the even columns, keeps their magnitudes, transforms the odd columns through the same
// images, and stores magnitude pairs (8 B per lane, whole lines per 16 lanes) -- half as many L2 requests per byte.
typedef float f32x4 __attribute__((ext_vector_type(4)));
#ifndef RSP_DOPPLER_PAIR
#define RSP_DOPPLER_PAIR 1
#endif
#ifndef RSP_DOPPLER_PAIR_FROM
#define RSP_DOPPLER_PAIR_FROM 10   // Doppler sizes (log2) from which the paired form is used
#endif
template <int MD>
__global__ void __launch_bounds__(threads_per_frame(MD) * kColsPerWg(MD))
doppler_mag_pair_kernel(const f32x2* __restrict__ in, float* __restrict__ mag, uint32_t n_ch, uint32_t nr, uint32_t tile,
                        int mag_mode, const f32x2* __restrict__ tw, const float* __restrict__ win) {
  constexpr int ND = 1 << MD, C2 = kColsPerWg(MD);  // column PAIRS per workgroup = LDS images
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, fl = tid % C2, tau = tid / C2;
  const uint32_t groups_per_ch = nr / (2 * C2);
  const uint32_t ch = blockIdx.x / groups_per_ch, r0 = (blockIdx.x % groups_per_ch) * (2 * C2) + 2 * fl;
  f32x2* buf = reinterpret_cast<f32x2*>(smem + (size_t)fl * kColBytes(MD));
  const size_t col = (size_t)ch * ND * nr + map_index(0, r0, ND, nr, tile);
  const size_t mcol = (size_t)ch * ND * nr + map_index<kMagTileCols>(0, r0, ND, nr, tile);
  const uint32_t pitch = tile ? kTileCols : nr, mpitch = tile ? kMagTileCols : nr;
  const int s0 = first_sample<MD>(tau);
  f32x4 raw[16];
  {
    const f32x2* src = in + col + (size_t)s0 * pitch;
#pragma unroll
    for (int e = 0; e < 16; ++e) raw[e] = rd_load<4>(reinterpret_cast<const f32x4*>(src + (size_t)sample_offset<MD>(e) * pitch));
    if (win) {  // slow-time window: one coefficient per row, the same for both columns
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float wv = win[s0 + sample_offset<MD>(e)];
        raw[e] = raw[e] * f32x4{wv, wv, wv, wv};
      }
    }
  }
  constexpr int NP = plan_np(MD), WL = plan_w(MD, NP - 1);
  const float scale = 1.0f / (float)ND;
  f32x2 x[16];
  float m0[16];
  {
    int e = 0;
    fft_f32_frame<MD>([&](int) { const f32x4 r = raw[e++]; return f32x2{r.x, r.y}; }, tau, buf, tw, x);
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) m0[e] = mag2d(x[e] * scale, mag_mode);
  __syncthreads();  // the images are rewritten by the second transform's first exchange
  {
    int e = 0;
    fft_f32_frame<MD>([&](int) { const f32x4 r = raw[e++]; return f32x2{r.z, r.w}; }, tau, buf, tw, x);
  }
  float* dst = mag + mcol;
#pragma unroll
  for (int g = 0; g < (16 >> WL); ++g) {
#pragma unroll
    for (int p = 0; p < (1 << WL); ++p) {
      const int e = g * (1 << WL) + p;
      rd_store<8>(f32x2{m0[e], mag2d(x[e] * scale, mag_mode)}, reinterpret_cast<f32x2*>(dst + (size_t)bin_of<MD>(tau, g, p) * mpitch));
    }
  }
}

// ---------------------------------------------------------------- FIXED16 passes (build extension, oracle: orc_rd_fixed)
// The same two passes on the 16-bit FixedPoint data path: beats {re[31:16], im[15:0]} in, the range spectrum kept as
// beats (4 B/cell), both FFTs with the stage-exact arithmetic of fft_fx_frame (Q2.14 ROM in LDS), Q1.15 windows,
// magnitudes (mag_fx) as int32.  FX = the trim path of the fixed-point FFT (0 convergent, 1 floor / half-up), one
// instantiation each: a kernel holding every path of fft_fx_frame carries the registers of the stage-option path
// (154-182 VGPRs, two waves per SIMD), which the 2-D chain never takes (rejected on the host).
template <int M, int FX>
__global__ void __launch_bounds__(wg_size(M))
range_fx_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t n_rows, uint32_t nd, uint32_t tile,
                const uint32_t* __restrict__ twq, ChainRegs rg, uint32_t* __restrict__ zero_count) {
  constexpr int N = 1 << M, T = threads_per_frame(M), FPW = frames_per_wg(M);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, fl = tid / T, tau = tid % T;
  const uint32_t row = blockIdx.x * FPW + fl;
  const bool live = row < n_rows;
  if (zero_count && blockIdx.x == 0 && tid == 0) zero_count[0] = zero_count[1] = 0u;
  unsigned char* fbase = smem + (size_t)fl * fft_image_slots(M) * 4;
  uint2* rom = reinterpret_cast<uint2*>(smem + (((size_t)FPW * fft_image_slots(M) * 4 + 7) & ~size_t(7)));
  const uint32_t* src = in + (size_t)(live ? row : 0) * N + first_sample<M>(tau);
  uint32_t z[16];   // beats stay packed {re[31:16], im[15:0]} from HBM to HBM
#pragma unroll
  for (int e = 0; e < 16; ++e) z[e] = src[sample_offset<M>(e)];
  if (rg.window) {  // Q1.15 coefficient, product rounded half-up back to 16 bits (spec section 2.1)
    const int16_t* wt = reinterpret_cast<const int16_t*>(rg.window) + first_sample<M>(tau);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int wq = wt[sample_offset<M>(e)];
      const int re = ((int)(short)(z[e] >> 16) * wq + (1 << 14)) >> 15, im = ((int)(short)(z[e] & 0xffffu) * wq + (1 << 14)) >> 15;
      z[e] = ((uint32_t)re << 16) | ((uint32_t)im & 0xffffu);
    }
  }
  fx_rom_fill(rom, twq, N / 2, tid, wg_size(M));
  __syncthreads();
  fft_fx_frame_pk<M, FX == 0>(z, tau, fbase, rom, rg);
  constexpr int NP = plan_np(M), WL = plan_w(M, NP - 1);
  if (!live) return;
  const uint32_t ch = row / nd, d = row % nd;
  uint32_t* dst = out + (size_t)ch * nd * N + map_index(d, tau, nd, N, tile);
  const size_t step = tile ? (size_t)nd : 1;
#pragma unroll
  for (int g = 0; g < (16 >> WL); ++g) {
#pragma unroll
    for (int p = 0; p < (1 << WL); ++p) {
      const int e = g * (1 << WL) + p;
      dst[(size_t)((bitrev_c(p, WL) << (M - WL)) + g * T) * step] = z[e];
    }
  }
}

constexpr int kFxCols = 16;  // range bins per workgroup of the FIXED16 Doppler pass
constexpr int kFxColBytes(int MD) { return 4 * fft_image_slots(MD) + (MD >= 10 ? 16 : 8); }  // column skew: as kColBytes, for 4-byte slots

template <int MD, int FX>
__global__ void __launch_bounds__(threads_per_frame(MD) * kFxCols)
doppler_fx_kernel(const uint32_t* __restrict__ in, int32_t* __restrict__ mag, uint32_t nr, uint32_t tile,
                  const uint32_t* __restrict__ twq, const int16_t* __restrict__ win, const int16_t* __restrict__ log_lut,
                  ChainRegs rg) {
  constexpr int ND = 1 << MD, T = threads_per_frame(MD), C = kFxCols;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, fl = tid % C, tau = tid / C;
  const uint32_t groups_per_ch = nr / C;
  const uint32_t ch = blockIdx.x / groups_per_ch, r0 = (blockIdx.x % groups_per_ch) * C;
  unsigned char* fbase = smem + (size_t)fl * kFxColBytes(MD);
  uint2* rom = reinterpret_cast<uint2*>(smem + (((size_t)C * kFxColBytes(MD) + 7) & ~size_t(7)));
  const size_t col = (size_t)ch * ND * nr + map_index(0, r0 + fl, ND, nr, tile);
  const uint32_t pitch = tile ? kTileCols : nr;
  const uint32_t* src = in + col + (size_t)first_sample<MD>(tau) * pitch;
  uint32_t z[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) z[e] = src[(size_t)sample_offset<MD>(e) * pitch];
  if (win) {
    const int16_t* wt = win + first_sample<MD>(tau);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int wq = wt[sample_offset<MD>(e)];
      const int re = ((int)(short)(z[e] >> 16) * wq + (1 << 14)) >> 15, im = ((int)(short)(z[e] & 0xffffu) * wq + (1 << 14)) >> 15;
      z[e] = ((uint32_t)re << 16) | ((uint32_t)im & 0xffffu);
    }
  }
  fx_rom_fill(rom, twq, ND / 2, tid, T * C);
  __syncthreads();
  fft_fx_frame_pk<MD, FX == 0>(z, tau, fbase, rom, rg);
  constexpr int NP = plan_np(MD), WL = plan_w(MD, NP - 1);
  int mg[16];
  if (rg.mag_mode == 2) {
#pragma unroll
    for (int e = 0; e < 16; e += 2) jpl_fx_pair(z[e], z[e + 1], mg[e], mg[e + 1]);
  } else {
#pragma unroll
    for (int e = 0; e < 16; ++e) mg[e] = mag_fx((int)(short)(z[e] >> 16), (int)(short)(z[e] & 0xffffu), rg, log_lut);
  }
  int32_t* dst = mag + (size_t)ch * ND * nr + map_index<kMagTileCols>(0, r0 + fl, ND, nr, tile);
  const uint32_t mpitch = tile ? kMagTileCols : nr;
#pragma unroll
  for (int g = 0; g < (16 >> WL); ++g) {
#pragma unroll
    for (int p = 0; p < (1 << WL); ++p) {
      const int e = g * (1 << WL) + p;
      dst[(size_t)bin_of<MD>(tau, g, p) * mpitch] = mg[e];
    }
  }
}

// ---------------------------------------------------------------- fused detection list
// d_count = {found, stored} is the list's own cursor: the range pass (first launch of the batch) zeroes it, a wave
// that found peaks reserves its entries with ONE returning device-scope atomic on found and folds
// min(found so far, cap) into stored with an atomic max (the maximum over all reservations is min(total, cap)).
// No counters to clean, no finalize launch (4.8 us per batch as its own kernel).  Peaks are ~1e-5 of the cells.
__device__ __forceinline__ uint32_t reserve_peaks(uint32_t* __restrict__ d_count, uint32_t cap, uint32_t n) {
  const uint32_t base = atomicAdd(&d_count[0], n);
  atomicMax(&d_count[1], base + n < cap ? base + n : cap);
  return base;
}
// The entries of a whole WORKGROUP (four waves) in one reservation: a dense scene gives every wave something to append, and
// the reservations of all the launch's waves queue up on one address (~10 ns each: 0.7 ms per 16.7 M cells when every
// 512-cell wave has a peak).  Every thread of the workgroup calls it with its wave's count n (wave-uniform); returns the
// base of this wave's entries.  Two barriers at the very end of the kernels that use it.
__device__ __forceinline__ uint32_t reserve_peaks_wg(uint32_t* __restrict__ d_count, uint32_t cap, uint32_t n, int wave, int lane) {
  __shared__ uint32_t wave_n[4], wg_base;
  if (lane == 0) wave_n[wave] = n;
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t tot = wave_n[0] + wave_n[1] + wave_n[2] + wave_n[3];
    wg_base = tot ? reserve_peaks(d_count, cap, tot) : 0u;
  }
  __syncthreads();
  uint32_t base = wg_base;
  for (int i = 0; i < wave; ++i) base += wave_n[i];
  return base;
}

// ---------------------------------------------------------------- 2-D CA-CFAR
// Training region = (2(ref_r+guard_r)+1) x (2(ref_d+guard_d)+1) box minus the guard box; out-of-map
// range cells read zero (edge 0) or wrap (edge 1), Doppler is cyclic; statistic = sum / count.
constexpr int kTD = 32, kTR = 64;  // output tile: 32 Doppler rows x 64 range bins per workgroup

// SRR/SGR/SRD/SGD >= 0: window half-widths fixed at compile time (loops unroll, LDS offsets become
// immediates: ~2.5x fewer instructions); -1: taken from the run-time arguments.
template <int SRR, int SGR, int SRD, int SGD, typename T = float>
__global__ void __launch_bounds__(256)
cfar2d_kernel(const T* __restrict__ mag, uint32_t* __restrict__ out, uint32_t nd, uint32_t nr,
              int ref_r_rt, int guard_r_rt, int ref_d_rt, int guard_d_rt, int edge, float kA, float kB,
              rsp_detection* __restrict__ det_list, uint32_t det_cap, uint32_t* __restrict__ det_count, uint32_t ch_base,
              int mode, uint32_t tile, ChainRegs rg, int log2nr) {
  // T = float: thresholds kA * sum / count + kB.  T = int32_t (FIXED16 magnitudes): integer sums, the statistic and
  // threshold rules of orc_rd_fixed through CfarMath<int> (rg).
  constexpr bool FX = !std::is_same<T, float>::value;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int ref_r = SRR >= 0 ? SRR : ref_r_rt, guard_r = SGR >= 0 ? SGR : guard_r_rt;
  const int ref_d = SRD >= 0 ? SRD : ref_d_rt, guard_d = SGD >= 0 ? SGD : guard_d_rt;
  const int hr = ref_r + guard_r, hd = ref_d + guard_d;
  const int RW = kTR + 2 * hr, RH = kTD + 2 * hd;  // haloed region
  const int MS = RW | 1;                            // odd row pitch of the magnitude region
  T* m = reinterpret_cast<T*>(smem);        // [RH][MS]
  T* ro = m + RH * MS;                          // outer row sums [RH][kTR + 1]
  // guard-box row sums are only needed for the kTD + 2 guard_d rows around the outputs; keeping that
  // array short is what lets four workgroups share a CU's LDS
  const int ri_first = hd - guard_d, ri_rows = kTD + 2 * guard_d + 1;
  T* ri = ro + (RH + 1) * (kTR + 1);            // [ri_rows][kTR + 1], row dd at index dd - ri_first
  // GO / SO (mode 1 / 2): the same sums over the LAGGING half of each row window (columns c - h .. c - 1)
  T* roL = ri + ri_rows * (kTR + 1);            // [RH][kTR + 1]
  T* riL = roL + (RH + 1) * (kTR + 1);          // [ri_rows][kTR + 1]
  const int tid = threadIdx.x;
  const uint32_t tiles_r = nr / kTR, tiles_d = nd / kTD;
  const uint32_t ch = blockIdx.x / (tiles_r * tiles_d);
  const uint32_t t = blockIdx.x % (tiles_r * tiles_d);
  const int d0 = (int)(t / tiles_r) * kTD, r0 = (int)(t % tiles_r) * kTR;
  const T* map = mag + (size_t)ch * nd * nr;

  // 1. region -> LDS: 128 lanes across a region row (coalesced along r), 2 rows per iteration
  {
    const int rr = tid & 127, half = tid >> 7;
    int r = r0 - hr + rr;
    if (edge) r = (r + (int)nr) & ((int)nr - 1);
    const bool inside = rr < RW && r >= 0 && r < (int)nr;
    // batches of 16 independent loads per thread: a one-load-per-iteration loop serialises on HBM latency
    for (int base = half; base < RH; base += 32) {
      T v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int dd = base + 2 * u;
        const int d = (d0 - hd + dd + (int)nd) & ((int)nd - 1);  // Doppler cyclic (nd is a power of two)
        v[u] = (inside && dd < RH) ? map[map_index<kMagTileCols>((uint32_t)d, (uint32_t)r, nd, nr, tile)] : T(0);
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int dd = base + 2 * u;
        if (rr < RW && dd < RH) m[dd * MS + rr] = v[u];
      }
    }
  }
  __syncthreads();
  // 2. row pass: sliding sums along r; task = (run of 16 output columns, region row), rows on
  // adjacent lanes: the row pitches (RW + 1, kTR + 1) are odd, so a wave's accesses spread over
  // all banks
  for (int task = tid; task < RH * (kTR / 16); task += 256) {
    const int dd = task % RH, c0 = (task / RH) * 16;
    const T* row = m + dd * MS + c0 + hr;  // row[c] = cell at output column c0 + c
    T so = row[0], si = row[0], so2 = T(0), si2 = T(0);  // symmetric windows: two independent chains
#pragma unroll
    for (int k = 1; k <= hr; ++k) { so += row[k]; so2 += row[-k]; }
#pragma unroll
    for (int k = 1; k <= guard_r; ++k) { si += row[k]; si2 += row[-k]; }
    so += so2;
    si += si2;
    T* po = ro + dd * (kTR + 1) + c0;
    const bool want_i = dd >= ri_first && dd < ri_first + ri_rows;
    T* pi = ri + (want_i ? dd - ri_first : 0) * (kTR + 1) + c0;
    if (mode == 0) {
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        po[c] = so;
        if (want_i) pi[c] = si;
        so += row[c + hr + 1] - row[c - hr];
        si += row[c + guard_r + 1] - row[c - guard_r];
      }
    } else {
      T* pol = roL + dd * (kTR + 1) + c0;
      T* pil = riL + (want_i ? dd - ri_first : 0) * (kTR + 1) + c0;
      T sol = so2, sil = si2;  // so2 / si2 = the sums over columns -h .. -1 computed above
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        po[c] = so;
        pol[c] = sol;
        if (want_i) { pi[c] = si; pil[c] = sil; }
        so += row[c + hr + 1] - row[c - hr];
        si += row[c + guard_r + 1] - row[c - guard_r];
        sol += row[c] - row[c - hr];
        sil += row[c] - row[c - guard_r];
      }
    }
  }
  __syncthreads();
  // 3. column pass: thread = (output column, run of 8 output rows); stores are 256 B per wave
  {
    const int c = tid & (kTR - 1), dseg = (tid / kTR) * (kTD / 4);
    const float count = (float)((2 * hr + 1) * (2 * hd + 1) - (2 * guard_r + 1) * (2 * guard_d + 1));
    const float kAc = kA / count;
    const T* co = ro + (dseg + hd) * (kTR + 1) + c;  // co[k (kTR+1)] = outer row sum at output row dseg + k
    const T* ci = ri + (dseg + hd - ri_first) * (kTR + 1) + c;
    T so = co[0], si = ci[0], so2 = T(0), si2 = T(0);
#pragma unroll
    for (int k = 1; k <= hd; ++k) { so += co[k * (kTR + 1)]; so2 += co[-k * (kTR + 1)]; }
#pragma unroll
    for (int k = 1; k <= guard_d; ++k) { si += ci[k * (kTR + 1)]; si2 += ci[-k * (kTR + 1)]; }
    so += so2;
    si += si2;
    uint32_t* dst = out + ((size_t)ch * nd + d0 + dseg) * nr + r0 + c;
    // GO / SO: column sums of the lagging-half row sums + the cells of the CUT's own column above the guard
    const T* col = roL + (dseg + hd) * (kTR + 1) + c;
    const T* cil = riL + (dseg + hd - ri_first) * (kTR + 1) + c;
    const T* mcol = m + (dseg + hd) * MS + c + hr;  // mcol[k MS] = the CUT column at output row dseg + k
    T sol = T(0), sil = T(0), up = T(0);
    if (mode != 0) {
      for (int k = -hd; k <= hd; ++k) sol += col[k * (kTR + 1)];
      for (int k = -guard_d; k <= guard_d; ++k) sil += cil[k * (kTR + 1)];
      for (int k = -hd; k < -guard_d; ++k) up += mcol[k * MS];
    }
    const float kAh = 2.0f * kAc;  // a half holds count / 2 cells
    uint32_t hits = 0u;  // bit j: output row dseg + j of this thread's column is a peak
#pragma unroll
    for (int j = 0; j < kTD / 4; ++j) {
      const T cut = m[(dseg + j + hd) * MS + c + hr];
      T lag = T(0), lead = T(0);
      if (mode != 0) {
        lag = sol - sil + up;
        lead = (so - si) - lag;
        sol += col[(j + hd + 1) * (kTR + 1)] - col[(j - hd) * (kTR + 1)];
        sil += cil[(j + guard_d + 1) * (kTR + 1)] - cil[(j - guard_d) * (kTR + 1)];
        up += mcol[(j - guard_d) * MS] - mcol[(j - hd) * MS];
      }
      uint32_t wd;
      if constexpr (FX) {
        const int sh = rg.div_sum > 0 ? rg.div_sum - 1 : 0;  // a half holds half the cells
        const int stat = mode == 0 ? (int)(so - si) >> rg.div_sum
                       : mode == 1 ? max((int)lag >> sh, (int)lead >> sh) : min((int)lag >> sh, (int)lead >> sh);
        wd = CfarMath<int>::finish(stat, (int)cut, true, r0 + c, log2nr, rg);
      } else {
        float thr = __fmaf_rn((float)(so - si), kAc, kB);
        if (mode != 0) thr = __fmaf_rn(mode == 1 ? fmaxf((float)lag, (float)lead) : fminf((float)lag, (float)lead), kAh, kB);
        wd = (__float_as_uint(thr) & ~1u) | (uint32_t)((float)cut > thr);
      }
      dst[(size_t)j * nr] = wd;
      hits |= (wd & 1u) << j;
      so += co[(j + hd + 1) * (kTR + 1)] - co[(j - hd) * (kTR + 1)];
      si += ci[(j + guard_d + 1) * (kTR + 1)] - ci[(j - guard_d) * (kTR + 1)];
    }
    // fused detection list: ONE reservation per workgroup (a lane scan gives every lane its offset inside its wave's
    // share), the words re-read by the lane that wrote them -- one device atomic per PEAK made a dense scene cost
    // hundreds of microseconds
    if (det_list) {
      const int lane = tid & 63;
      const uint32_t mine = (uint32_t)__popc(hits);
      uint32_t incl = mine;
#pragma unroll
      for (int sh = 1; sh < 64; sh <<= 1) {
        const uint32_t t = __shfl_up(incl, sh);
        if (lane >= sh) incl += t;
      }
      const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
      const uint32_t base = reserve_peaks_wg(det_count, det_cap, total, tid >> 6, lane);
      if (total) {
        uint32_t pos = base + incl - mine;
        while (hits) {
          const int j = __ffs(hits) - 1;
          hits &= hits - 1;
          if (pos < det_cap) det_list[pos] = rsp_detection{ch_base + ch, (uint32_t)(r0 + c), (uint32_t)(d0 + dseg + j), dst[(size_t)j * nr]};
          ++pos;
        }
      }
    }
  }
}

// ---------------------------------------------------------------- 2-D CA-CFAR, compile-time windows
// One WAVE walks a 128-column strip (2 adjacent range bins per lane, 8-byte loads and stores) down
// kWalkSeg Doppler rows.  The last 2 HD + 2 map rows live in a register ring (fully unrolled, so
// ring indices are compile-time); the ring's spare slots are rows already in flight from HBM.
//   vertical:   Vo / Vi = sums over the 2 HD + 1 / 2 GD + 1 rows around the output row, updated by
//               one add and one subtract per row (registers only);
//   horizontal: window sums across lanes from runs of 2 / 4 lanes (DPP wave shifts) + ds_bpermute taps (walk_issue /
//               walk_finish below; no LDS memory, no barriers).
// Lanes LB .. LE own complete windows: 2 (LE - LB + 1) output columns per strip.
template <int CTRL, int ROWMASK, bool BOUND>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROWMASK, 0xf, BOUND));
}

// Row-direction sums of a lane pair's column values v = (column 2l, column 2l + 1), H even:
//   box = (sum of columns c - H .. c + H) for c = 2l and 2l + 1,  lag = (sum of columns c - H .. c - 1)
// from runs of 2 and 4 lanes of the pair sums s = v.x + v.y.  Shifts by one lane are DPP wave shifts (VALU), longer
// ones ds_bpermute (LDS crossbar, no memory): lanes beyond the wave read zero or a wrapped lane, which only reaches
// lanes outside LB .. LE (they own no output).  Split in two so that the walk can issue the cross-lane reads of one
// row before it finishes the previous row (walk_issue holds no arithmetic on a ds_bpermute result): a step then
// never waits for the crossbar round trip it has just started.  (A DPP prefix scan + 8 ds_bpermute taps per step
// measured 33.5 us at 8 x 4096 x 512; these sums 32.2 us; issued one step ahead: see DESIGN 3.2.)
// S = float (fp32 maps) or int32_t (FIXED16 magnitudes: exact integer sums)
template <typename S> struct Pair { typedef S type __attribute__((ext_vector_type(2))); };
template <typename S> struct WalkBox { typename Pair<S>::type box, lag; };
template <int CTRL, typename S>
__device__ __forceinline__ S dpp_s(S v) {
  return __builtin_bit_cast(S, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
template <typename S> __device__ __forceinline__ S lane_up1(S v) { return dpp_s<0x138, S>(v); }  // wave_shr:1: lane l - 1
template <typename S> __device__ __forceinline__ S lane_dn1(S v) { return dpp_s<0x130, S>(v); }  // wave_shl:1: lane l + 1
template <int K, typename S>
__device__ __forceinline__ S lane_at(S v, int lane) {  // value of lane l + K
  if constexpr (K == 0) return v;
  else if constexpr (K == -1) return lane_up1(v);
  else if constexpr (K == 1) return lane_dn1(v);
  else return __builtin_bit_cast(S, __builtin_amdgcn_ds_bpermute(((lane + K) & 63) << 2, __builtin_bit_cast(int, v)));
}
constexpr int lane_runs(int n) { return n <= 0 ? 0 : n / 4 + ((n % 4) >= 2) + (n % 2); }
// the runs a_k[m] = s[m - 2^k + 1 .. m] (k <= 2) that tile lanes l + FIRST .. l + FIRST + N - 1, longest first, into out[I..]
template <int FIRST, int N, int I = 0, typename S, int NT>
__device__ __forceinline__ void lane_run_taps(S s, S a1, S a2, int lane, S (&out)[NT]) {
  if constexpr (N > 0) {
    constexpr int k = N >= 4 ? 2 : N >= 2 ? 1 : 0, last = FIRST + N - 1;
    out[I] = lane_at<last>(k == 2 ? a2 : k == 1 ? a1 : s, lane);
    lane_run_taps<FIRST, N - (1 << k), I + 1>(s, a1, a2, lane, out);
  }
}
template <int H, bool LAG, typename S>
struct WalkTaps {
  static constexpr int h = H / 2, NC = h == 1 ? 1 : lane_runs(2 * h), NM = (h == 1 || !LAG) ? 1 : lane_runs(h - 1);
  S core[NC], mid[NM], ym, sm, yp, x;
};
template <int H, bool LAG, typename S>
__device__ __forceinline__ WalkTaps<H, LAG, S> walk_issue(typename Pair<S>::type v, int lane) {
  static_assert(H % 2 == 0 && H >= 2 && H <= 30, "even half-width");
  constexpr int h = H / 2;
  const S s = v.x + v.y;
  WalkTaps<H, LAG, S> t;
  t.x = v.x;
  if constexpr (h == 1) {  // one-lane shifts only: no crossbar
    t.core[0] = s; t.mid[0] = S(0);
    t.sm = lane_up1(s); t.ym = lane_up1((S)v.y); t.yp = lane_dn1((S)v.x); t.x = lane_dn1(s);
  } else {
    // runs of 2 and 4 lanes ending at this lane, built with wave shifts only: every ds_bpermute depends on VALU results alone
    const S a1 = s + lane_up1(s);
    const S a2 = a1 + lane_up1(lane_up1(a1));
    lane_run_taps<-h + 1, 2 * h>(s, a1, a2, lane, t.core);  // lanes l - h + 1 .. l + h (columns 2l - 2h + 2 .. 2l + 2h + 1)
    if constexpr (LAG) lane_run_taps<-h + 1, h - 1>(s, a1, a2, lane, t.mid);  // lanes l - h + 1 .. l - 1
    else t.mid[0] = S(0);
    t.ym = lane_at<-h>((S)v.y, lane); t.sm = lane_at<-h>(s, lane); t.yp = lane_at<h>((S)v.y, lane);
  }
  return t;
}
template <int H, bool LAG, typename S>
__device__ __forceinline__ WalkBox<S> walk_finish(const WalkTaps<H, LAG, S>& t, typename Pair<S>::type v) {
  typedef typename Pair<S>::type S2;
  constexpr int h = H / 2;
  WalkBox<S> r;
  if constexpr (h == 1) {
    // box(2l) = s[l-1] + s[l] + x[l+1], box(2l+1) = y[l-1] + s[l] + s[l+1]; lag(2l) = s[l-1], lag(2l+1) = y[l-1] + x[l]
    r.box = S2{(t.sm + t.core[0]) + t.yp, (t.ym + t.core[0]) + t.x};
    r.lag = S2{t.sm, t.ym + v.x};
  } else {
    S core = t.core[0], mid = t.mid[0];
#pragma unroll
    for (int k = 1; k < WalkTaps<H, LAG, S>::NC; ++k) core += t.core[k];
#pragma unroll
    for (int k = 1; k < WalkTaps<H, LAG, S>::NM; ++k) mid += t.mid[k];
    r.box = S2{(core + t.sm) - t.yp, core + t.ym};
    // lagging columns: lanes l - h + 1 .. l - 1 whole, plus lane l - h (whole / upper column) and column 2l for c = 2l + 1
    r.lag = S2{mid + t.sm, (mid + t.ym) + v.x};
  }
  return r;
}
#ifndef RSP_WALK_ALTERNATE
#define RSP_WALK_ALTERNATE 1
#endif
constexpr int kWalkRing = 32;
constexpr int kWalkStage = 126;  // detections a wave stages in LDS per segment (2 KiB per wave)
constexpr int walk_lb(int hr) { return (hr + 2) / 2; }
constexpr int walk_le(int hr) { return (126 - hr) / 2; }
constexpr int walk_outw(int hr) { return 2 * (walk_le(hr) - walk_lb(hr) + 1); }

template <int RR, int GR, int RD, int GD, int SEG, int MODE, typename S = float>
__global__ void __launch_bounds__(256)
cfar2d_walk_kernel(const S* __restrict__ mag, uint32_t* __restrict__ out, uint32_t nd, uint32_t nr,
                   uint32_t strips, int edge, float kA, float kB,
                   rsp_detection* __restrict__ det_list, uint32_t det_cap, uint32_t* __restrict__ det_count, uint32_t ch_base,
                   uint32_t tile, ChainRegs rg, int log2nr, uint32_t stream_words) {
  typedef typename Pair<S>::type S2;
  constexpr bool FX = !std::is_same<S, float>::value;  // FIXED16 magnitudes: integer sums, thresholds through CfarMath<int>
  constexpr int HR = RR + GR, HD = RD + GD, SPAN = 2 * HD + 2, RING = kWalkRing;
  constexpr int LB = walk_lb(HR), LE = walk_le(HR), OUTW = walk_outw(HR);
  static_assert(SPAN < RING && SEG % RING == 0, "ring holds the taps plus at least one row in flight");
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t seg_groups = nd / SEG / 4;
  // contiguous run of strips per XCD: neighbouring strips share halo columns and rows through one L2
  const uint32_t blk = (gridDim.x % 8 == 0) ? (blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8 : blockIdx.x;
  const uint32_t sg = blk % seg_groups, strip = (blk / seg_groups) % strips;
  const uint32_t ch = blk / (seg_groups * strips);
  const int d0 = (int)(sg * 4 + w) * SEG;
  const int col = (int)(strip * OUTW) - 2 * LB + 2 * lane;  // even: both columns in or both out of the map
  // buffer addressing: row offset in an SGPR, lane offset constant; a lane outside the map (zero edge)
  // or without a complete window gets an out-of-range offset: its loads return 0, its stores are dropped
  constexpr uint32_t kOob = 0xfffffff0u, kRsrc3 = 0x00020000u;
  const uint32_t map_bytes = nd * nr * 4u;
  const __amdgpu_buffer_rsrc_t rsrc_in = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<S*>(mag) + (size_t)ch * nd * nr, 0, (int)map_bytes, (int)kRsrc3);
  const __amdgpu_buffer_rsrc_t rsrc_out =
      __builtin_amdgcn_make_buffer_rsrc(out + (size_t)ch * nd * nr, 0, (int)map_bytes, (int)kRsrc3);
  const bool readable = edge || (col >= 0 && col < (int)nr);
  const bool owner = lane >= LB && lane <= LE && col < (int)nr;
  const uint32_t voff_in = readable ? (uint32_t)map_index<kMagTileCols>(0, (uint32_t)(col & ((int)nr - 1)), nd, nr, tile) * 4u : kOob;
  const uint32_t row_bytes = (tile ? kMagTileCols : nr) * 4u;  // address step between Doppler rows of the magnitude map
#ifdef RSP_ABL_WALK_NOSTORE  // ablation builds (tools/ablate_rd.sh): never defined in the product
  const uint32_t voff_out = kOob;
#else
  const uint32_t voff_out = owner ? (uint32_t)col * 4u : kOob;
#endif
  const float count = (float)((2 * HR + 1) * (2 * HD + 1) - (2 * GR + 1) * (2 * GD + 1));
  const float kAc = kA / count;
  // stream row p <-> map row d0 - HD + p (Doppler cyclic); odd waves walk their segment upwards (row d0 + SEG - 1 +
  // HD - p): a segment's halo rows are then read at the same time as its neighbour in the workgroup reads them as
  // its own (both at the start or both at the end of their walks), i.e. once from HBM instead of twice
  const bool up = RSP_WALK_ALTERNATE && (w & 1);
  auto load_row = [&](int p) -> S2 {
#ifdef RSP_ABL_WALK_ROW0
    const uint32_t d = (uint32_t)(p & 31);
#else
    const uint32_t d = (uint32_t)((up ? d0 + SEG - 1 + HD - p : d0 - HD + p) & ((int)nd - 1));
#endif
    return __builtin_bit_cast(S2, __builtin_amdgcn_raw_buffer_load_b64(rsrc_in, voff_in, d * row_bytes, ((RSP_RD_NT) & 16) ? 2 : 0));  // aux bit 1 = nt
  };
  // fused detection list: peaks are staged in a wave-private LDS buffer during the walk (LDS atomics
  // only: a vector-memory atomic inside the walk, even in a never-taken branch, makes the wait-count
  // merge drain the row prefetch on every step -- 4x slower, measured) and flushed after it
  __shared__ uint32_t stage_cnt[4];
  __shared__ u32x4 stage[4][kWalkStage];
  const uint32_t det_mask = (det_list && owner) ? 1u : 0u;
  if (lane == 0) stage_cnt[w] = 0u;
  S2 ring[RING];
#pragma unroll
  for (int p = 0; p < RING - 1; ++p) ring[p] = load_row(p);
  S2 vo = ring[0], vi = ring[HD - GD];
#pragma unroll
  for (int p = 1; p <= 2 * HD; ++p) vo += ring[p];
#pragma unroll
  for (int p = HD - GD + 1; p <= HD + GD; ++p) vi += ring[p];
  // GO / SO: the RD rows of the CUT's own column at smaller Doppler than the guard band belong to the lagging half:
  // the first RD rows of the window when walking down, the last RD when walking up
  S2 vup = {S(0), S(0)};
  if constexpr (MODE != 0) {
    if (up) {
#pragma unroll
      for (int p = HD + GD + 1; p <= 2 * HD; ++p) vup += ring[p];
    } else {
#pragma unroll
      for (int p = 0; p < RD; ++p) vup += ring[p];
    }
  }
  const float kAh = 2.0f * kAc;
  constexpr bool LAG = MODE != 0;
  // software pipeline, one step deep: the cross-lane reads of row i + 1 are in flight while row i is finished
  WalkTaps<HR, LAG, S> to = walk_issue<HR, LAG, S>(vo, lane);
  WalkTaps<GR, LAG, S> ti = walk_issue<GR, LAG, S>(vi, lane);
  S2 vo_cur = vo, vi_cur = vi, vup_cur = vup;
  for (int chunk = 0; chunk < SEG / RING; ++chunk) {
#pragma unroll
    for (int u = 0; u < RING; ++u) {
      const int i = chunk * RING + u;
      if (i + RING - 1 < SEG + 2 * HD) ring[(u + RING - 1) % RING] = load_row(i + RING - 1);  // rows past the walk's last window are not read
      const S2 cut = ring[(u + HD) % RING];
      // column sums of row i + 1 and its cross-lane reads (the last step's are never used)
      vo += ring[(u + SPAN - 1) % RING] - ring[u];
      vi += ring[(u + HD + GD + 1) % RING] - ring[(u + HD - GD) % RING];
      if constexpr (LAG) vup += up ? ring[(u + SPAN - 1) % RING] - ring[(u + HD + GD + 1) % RING] : ring[(u + RD) % RING] - ring[u];
      const WalkTaps<HR, LAG, S> to_next = walk_issue<HR, LAG, S>(vo, lane);
      const WalkTaps<GR, LAG, S> ti_next = walk_issue<GR, LAG, S>(vi, lane);
      __builtin_amdgcn_sched_barrier(0);
      const WalkBox<S> bo = walk_finish<HR, LAG, S>(to, vo_cur), bi = walk_finish<GR, LAG, S>(ti, vi_cur);
      u32x2 wd;
      if constexpr (FX) {
        // orc_rd_fixed: CA: training sum >> div_sum; GO / SO: each half's sum >> (div_sum - 1), the greater / smaller
        S2 stat;
        if constexpr (MODE == 0) {
          const S2 sum = bo.box - bi.box;
          stat = S2{sum.x >> rg.div_sum, sum.y >> rg.div_sum};
        } else {
          const int sh = rg.div_sum > 0 ? rg.div_sum - 1 : 0;
          const S2 lag = (bo.lag - bi.lag) + vup_cur, lead = (bo.box - bi.box) - lag;
          const S2 lg = S2{lag.x >> sh, lag.y >> sh}, ld = S2{lead.x >> sh, lead.y >> sh};
          stat = S2{MODE == 1 ? max(lg.x, ld.x) : min(lg.x, ld.x), MODE == 1 ? max(lg.y, ld.y) : min(lg.y, ld.y)};
        }
        wd.x = CfarMath<int>::finish(stat.x, cut.x, true, col, log2nr, rg);
        wd.y = CfarMath<int>::finish(stat.y, cut.y, true, col + 1, log2nr, rg);
      } else {
        float t0, t1;
        if constexpr (MODE == 0) {
          t0 = __fmaf_rn(bo.box.x - bi.box.x, kAc, kB);
          t1 = __fmaf_rn(bo.box.y - bi.box.y, kAc, kB);
        } else {
          // lagging half: columns c - H .. c - 1 of the box sums + the rows of column c above the guard; leading half =
          // the rest of the training region
          const S2 lag = (bo.lag - bi.lag) + vup_cur, lead = (bo.box - bi.box) - lag;
          t0 = __fmaf_rn(MODE == 1 ? fmaxf(lag.x, lead.x) : fminf(lag.x, lead.x), kAh, kB);
          t1 = __fmaf_rn(MODE == 1 ? fmaxf(lag.y, lead.y) : fminf(lag.y, lead.y), kAh, kB);
        }
        wd.x = (__float_as_uint(t0) & ~1u) | (uint32_t)(cut.x > t0);
        wd.y = (__float_as_uint(t1) & ~1u) | (uint32_t)(cut.y > t1);
      }
      const uint32_t d_out = (uint32_t)(up ? d0 + SEG - 1 - i : d0 + i);
      // stream_words (wave-uniform): the words leave with the non-temporal hint, so that they do not push the launch's
      // intermediate maps out of the Infinity Cache -- 119-120 -> 113.6-114.5 us at 8 x 4096 x 512 (201 MB of maps), but
      // 522-531 -> 545 us at 8 x 8192 x 1024 (805 MB: nothing to protect), hence per launch (rd_stream_words())
      if (stream_words) __builtin_amdgcn_raw_buffer_store_b64(wd, rsrc_out, voff_out, d_out * nr * 4u, 2);  // aux bit 1 = nt
      else __builtin_amdgcn_raw_buffer_store_b64(wd, rsrc_out, voff_out, d_out * nr * 4u, 0);
      if ((wd.x | wd.y) & det_mask) {  // rare
        if (wd.x & 1u) {
          const uint32_t sl = atomicAdd(&stage_cnt[w], 1u);
          if (sl < (uint32_t)kWalkStage) stage[w][sl] = u32x4{ch_base + ch, (uint32_t)col, d_out, wd.x};
        }
        if (wd.y & 1u) {
          const uint32_t sl = atomicAdd(&stage_cnt[w], 1u);
          if (sl < (uint32_t)kWalkStage) stage[w][sl] = u32x4{ch_base + ch, (uint32_t)col + 1u, d_out, wd.y};
        }
      }
      to = to_next; ti = ti_next; vo_cur = vo; vi_cur = vi; vup_cur = vup;
      // keep every row's load at the top of its own step: the scheduler would otherwise sink the
      // loads next to their first use, 10 rows later, and drain the prefetch pipeline
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (det_list) {
    const uint32_t n = stage_cnt[w];  // wave-private: this wave's LDS operations execute in order; exact even past kWalkStage
    const uint32_t base = reserve_peaks_wg(det_count, det_cap, n, w, lane);  // once per workgroup that found something
    if (n > 0u && n <= (uint32_t)kWalkStage) {
      for (uint32_t k = lane; k < n; k += 64) {
        if (base + k < det_cap) {
          const u32x4 e = stage[w][k];
          rsp_detection d;
          d.frame = e.x;
          d.bin = e.y;
          d.doppler = e.z;
          d.word = e.w;
          det_list[base + k] = d;
        }
      }
    } else if (n > (uint32_t)kWalkStage) {
      // more peaks than the staging holds (thresholds near the noise floor): every lane reads its own words back and
      // writes its peaks behind those of the lanes before it (a lane scan of the per-lane counts).  (One atomic per PEAK
      // here cost 460 us per 117 k peaks.)
      uint32_t mine = 0u;
      if (owner) {
#pragma unroll 1
        for (int i = 0; i < SEG; ++i) {
          const u32x2 wd = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rsrc_out, voff_out, (uint32_t)(d0 + i) * nr * 4u, 1 /* glc */));
          mine += (wd.x & 1u) + (wd.y & 1u);
        }
      }
      uint32_t incl = mine;
#pragma unroll
      for (int sh = 1; sh < 64; sh <<= 1) {
        const uint32_t t = __shfl_up(incl, sh);
        if (lane >= sh) incl += t;
      }
      uint32_t pos = base + incl - mine;
      if (owner && mine) {
#pragma unroll 1
        for (int i = 0; i < SEG; ++i) {
          const u32x2 wd = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rsrc_out, voff_out, (uint32_t)(d0 + i) * nr * 4u, 1 /* glc */));
          if (wd.x & 1u) {
            if (pos < det_cap) det_list[pos] = rsp_detection{ch_base + ch, (uint32_t)col, (uint32_t)(d0 + i), wd.x};
            ++pos;
          }
          if (wd.y & 1u) {
            if (pos < det_cap) det_list[pos] = rsp_detection{ch_base + ch, (uint32_t)col + 1u, (uint32_t)(d0 + i), wd.y};
            ++pos;
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------- launchers
// the walker's word stores carry the streaming hint when the launch's intermediate maps (spectrum + magnitudes,
// bytes_per_cell of them) fit the 256-MiB Infinity Cache with room to spare
static uint32_t rd_stream_words(uint32_t n_ch, uint32_t nd, uint32_t nr, uint32_t bytes_per_cell) {
  return (uint64_t)n_ch * nd * nr * bytes_per_cell <= (uint64_t)224 << 20 ? 1u : 0u;
}


template <int M>
static hipError_t launch_range_m(const f32x2* in, f32x2* out, uint32_t n_rows, int log2nd, uint32_t tile, const f32x2* tw, const float* win, uint32_t* zero_count,
                                 hipStream_t s, int device) {
  constexpr int fpw = frames_per_wg(M);
  const size_t lds = (size_t)fpw * fft_image_slots(M) * sizeof(f32x2);
  auto go = [&](auto k, LdsGrant& granted) -> hipError_t {
    hipError_t e = grant_lds(k, lds, device, granted);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3((n_rows + fpw - 1) / fpw), dim3(wg_size(M)), lds, s, in, out, n_rows, (uint32_t)log2nd, tw, win, zero_count);
    return hipGetLastError();
  };
  static LdsGrant g2[2];
  return tile ? go(range_fft_kernel<M, true>, g2[0]) : go(range_fft_kernel<M, false>, g2[1]);
}

template <int MD>
static hipError_t launch_doppler_m(const f32x2* in, float* mag, uint32_t n_ch, uint32_t nr, uint32_t tile, int mode,
                                   const f32x2* tw, const float* win, hipStream_t s, int device) {
  if constexpr (MD >= (RSP_DOPPLER_PAIR_FROM) && (RSP_DOPPLER_PAIR)) {  // two columns per LDS image (see doppler_mag_pair_kernel)
    constexpr int C2 = kColsPerWg(MD);
    const size_t lds = (size_t)kColBytes(MD) * C2;
    auto k = doppler_mag_pair_kernel<MD>;
    static LdsGrant granted;
    hipError_t e = grant_lds(k, lds, device, granted);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3(n_ch * (nr / (2 * C2))), dim3(threads_per_frame(MD) * C2), lds, s, in, mag, n_ch, nr, tile, mode, tw, win);
    return hipGetLastError();
  } else {
    constexpr int C = kColsPerWg(MD);
    const size_t lds = (size_t)kColBytes(MD) * C;
    auto k = doppler_mag_kernel<MD>;
    static LdsGrant granted;
    hipError_t e = grant_lds(k, lds, device, granted);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3(n_ch * (nr / C)), dim3(threads_per_frame(MD) * C), lds, s, in, mag, n_ch,
                       nr, tile, mode, tw, win);
    return hipGetLastError();
  }
}

// one chunk of channels through the three kernels; scratch buffers are the chunk's own
static hipError_t launch_rd2d_chunk(const Rd2dLaunch& a, uint32_t ch0, uint32_t n_ch) {
  const uint32_t nr = 1u << a.log2nr, nd = 1u << a.log2nd;
  const size_t map = (size_t)nr * nd;
  hipError_t e;
  const f32x2* in = reinterpret_cast<const f32x2*>(a.in) + (size_t)ch0 * map;
  uint32_t* out = a.out + (size_t)ch0 * map;
  f32x2* x1 = reinterpret_cast<f32x2*>(a.scratch_complex);
  const f32x2* twr = reinterpret_cast<const f32x2*>(a.tw_range);
  const f32x2* twd = reinterpret_cast<const f32x2*>(a.tw_doppler);
  const float* wr = reinterpret_cast<const float*>(a.regs.window);
  const float* wd = reinterpret_cast<const float*>(a.win_doppler);
  const uint32_t rows = n_ch * nd, tile = rd_tile(a.log2nr);
  uint32_t* zero_count = (a.det_count && ch0 == 0) ? a.det_count : nullptr;  // first chunk of the batch
  // count-only call (cap = 0, no list buffer): the kernels count their peaks when handed ANY non-null list pointer and
  // never store through it (every store is behind `slot < cap`)
  rsp_detection* det_list = a.det_count ? (a.det_list ? a.det_list : reinterpret_cast<rsp_detection*>(a.det_count)) : nullptr;
  const uint32_t det_cap = a.det_list ? a.det_cap : 0u;
  switch (a.log2nr) {
    case 8: e = launch_range_m<8>(in, x1, rows, a.log2nd, tile, twr, wr, zero_count, a.stream, a.device); break;
    case 9: e = launch_range_m<9>(in, x1, rows, a.log2nd, tile, twr, wr, zero_count, a.stream, a.device); break;
    case 10: e = launch_range_m<10>(in, x1, rows, a.log2nd, tile, twr, wr, zero_count, a.stream, a.device); break;
    case 11: e = launch_range_m<11>(in, x1, rows, a.log2nd, tile, twr, wr, zero_count, a.stream, a.device); break;
    case 12: e = launch_range_m<12>(in, x1, rows, a.log2nd, tile, twr, wr, zero_count, a.stream, a.device); break;
    case 13: e = launch_range_m<13>(in, x1, rows, a.log2nd, tile, twr, wr, zero_count, a.stream, a.device); break;
    default: return hipErrorInvalidValue;
  }
  if (e != hipSuccess) return e;
  switch (a.log2nd) {
    case 8: e = launch_doppler_m<8>(x1, a.scratch_mag, n_ch, nr, tile, a.regs.mag_mode, twd, wd, a.stream, a.device); break;
    case 9: e = launch_doppler_m<9>(x1, a.scratch_mag, n_ch, nr, tile, a.regs.mag_mode, twd, wd, a.stream, a.device); break;
    case 10: e = launch_doppler_m<10>(x1, a.scratch_mag, n_ch, nr, tile, a.regs.mag_mode, twd, wd, a.stream, a.device); break;
    default: return hipErrorInvalidValue;
  }
  if (e != hipSuccess) return e;
  const int hr = a.regs.R + a.regs.G, hd = a.ref_d + a.guard_d;
  const size_t lds = 4 * ((size_t)(kTD + 2 * hd) * ((kTR + 2 * hr) | 1) +
                          (a.regs.cfar_mode ? 2 : 1) * ((size_t)(kTD + 2 * hd + 1) * (kTR + 1) + (size_t)(kTD + 2 * a.guard_d + 1) * (kTR + 1)));
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  const float kA = a.regs.linear ? a.regs.scaler_f : 1.0f, kB = a.regs.linear ? 0.0f : a.regs.scaler_f;
  const bool spec = a.regs.R == 8 && a.regs.G == 2 && a.ref_d == 8 && a.guard_d == 2;  // cfg 3 / cfg 5
  if (spec && !a.force_tiled_cfar) {
    const uint32_t strips = (nr + walk_outw(10) - 1) / walk_outw(10);
    // rows per wave.  Every segment reads 2 HD halo rows beside its own, but with the waves of a workgroup walking
    // towards / away from each other those are L2 hits (8.07 B/cell of HBM traffic at 64 rows): the choice is about
    // parallelism and latency.  32 rows: 29 us at 8 x 4096 x 512 (same as 64), 137 against 150 us at 8 x 8192 x 1024;
    // 128 rows: 35 / 148 us.
#ifndef RSP_WALK_SEG
#define RSP_WALK_SEG 32
#endif
    constexpr uint32_t SEG = RSP_WALK_SEG;
    const dim3 grid(n_ch * strips * (nd / SEG / 4));
#define RSP_WALK(MODE)                                                                                              \
  hipLaunchKernelGGL((cfar2d_walk_kernel<8, 2, 8, 2, SEG, MODE>), grid, dim3(256), 0, a.stream, a.scratch_mag, out, \
                     nd, nr, strips, a.regs.edge, kA, kB, det_list, det_cap, a.det_count, ch0, tile, a.regs, a.log2nr, \
                     rd_stream_words(n_ch, nd, nr, 12))
    if (a.regs.cfar_mode == 0) RSP_WALK(0);
    else if (a.regs.cfar_mode == 1) RSP_WALK(1);
    else RSP_WALK(2);
#undef RSP_WALK
    return hipGetLastError();
  }
  auto k = cfar2d_kernel<-1, -1, -1, -1, float>;
  static LdsGrant granted;
  e = grant_lds(k, lds, a.device, granted);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k, dim3(n_ch * (nr / kTR) * (nd / kTD)), dim3(256), lds, a.stream, a.scratch_mag, out,
                     nd, nr, a.regs.R, a.regs.G, a.ref_d, a.guard_d, a.regs.edge, kA, kB, det_list, det_cap,
                     a.det_count, ch0, a.regs.cfar_mode, tile, a.regs, a.log2nr);
  return hipGetLastError();
}

// ---- FIXED16 data path: three launches of the same shape
template <int M>
static hipError_t launch_range_fx(const uint32_t* in, uint32_t* out, uint32_t n_rows, uint32_t nd, uint32_t tile,
                                  const uint32_t* twq, const ChainRegs& rg, uint32_t* zero_count, hipStream_t s, int device) {
  constexpr int fpw = frames_per_wg(M);
  const size_t lds = (((size_t)fpw * fft_image_slots(M) * 4 + 7) & ~size_t(7)) + fx_rom_bytes(M);
  auto go = [&](auto k, LdsGrant& granted) -> hipError_t {
    hipError_t e = grant_lds(k, lds, device, granted);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3((n_rows + fpw - 1) / fpw), dim3(wg_size(M)), lds, s, in, out, n_rows, nd, tile, twq, rg, zero_count);
    return hipGetLastError();
  };
  static LdsGrant g2[2];
  return rg.trim_conv ? go(range_fx_kernel<M, 0>, g2[0]) : go(range_fx_kernel<M, 1>, g2[1]);
}
template <int MD>
static hipError_t launch_doppler_fx(const uint32_t* in, int32_t* mag, uint32_t n_ch, uint32_t nr, uint32_t tile,
                                    const uint32_t* twq, const int16_t* win, const int16_t* log_lut, const ChainRegs& rg,
                                    hipStream_t s, int device) {
  const size_t lds = (((size_t)kFxColBytes(MD) * kFxCols + 7) & ~size_t(7)) + fx_rom_bytes(MD);
  auto go = [&](auto k, LdsGrant& granted) -> hipError_t {
    hipError_t e = grant_lds(k, lds, device, granted);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3(n_ch * (nr / kFxCols)), dim3(threads_per_frame(MD) * kFxCols), lds, s, in, mag, nr, tile, twq, win,
                       log_lut, rg);
    return hipGetLastError();
  };
  static LdsGrant g2[2];
  return rg.trim_conv ? go(doppler_fx_kernel<MD, 0>, g2[0]) : go(doppler_fx_kernel<MD, 1>, g2[1]);
}

static hipError_t launch_rd2d_chunk_fx(const Rd2dLaunch& a, uint32_t ch0, uint32_t n_ch) {
  const uint32_t nr = 1u << a.log2nr, nd = 1u << a.log2nd;
  const size_t map = (size_t)nr * nd;
  hipError_t e;
  const uint32_t* in = reinterpret_cast<const uint32_t*>(a.in) + (size_t)ch0 * map;
  uint32_t* out = a.out + (size_t)ch0 * map;
  uint32_t* x1 = reinterpret_cast<uint32_t*>(a.scratch_complex);
  int32_t* mag = reinterpret_cast<int32_t*>(a.scratch_mag);
  const uint32_t* twr = reinterpret_cast<const uint32_t*>(a.tw_range);
  const uint32_t* twd = reinterpret_cast<const uint32_t*>(a.tw_doppler);
  const uint32_t rows = n_ch * nd, tile = rd_tile(a.log2nr);
  uint32_t* zero_count = (a.det_count && ch0 == 0) ? a.det_count : nullptr;
  // count-only call (cap = 0, no list buffer): the kernels count their peaks when handed ANY non-null list pointer and
  // never store through it (every store is behind `slot < cap`)
  rsp_detection* det_list = a.det_count ? (a.det_list ? a.det_list : reinterpret_cast<rsp_detection*>(a.det_count)) : nullptr;
  const uint32_t det_cap = a.det_list ? a.det_cap : 0u;
  switch (a.log2nr) {
    case 8: e = launch_range_fx<8>(in, x1, rows, nd, tile, twr, a.regs, zero_count, a.stream, a.device); break;
    case 9: e = launch_range_fx<9>(in, x1, rows, nd, tile, twr, a.regs, zero_count, a.stream, a.device); break;
    case 10: e = launch_range_fx<10>(in, x1, rows, nd, tile, twr, a.regs, zero_count, a.stream, a.device); break;
    case 11: e = launch_range_fx<11>(in, x1, rows, nd, tile, twr, a.regs, zero_count, a.stream, a.device); break;
    case 12: e = launch_range_fx<12>(in, x1, rows, nd, tile, twr, a.regs, zero_count, a.stream, a.device); break;
    case 13: e = launch_range_fx<13>(in, x1, rows, nd, tile, twr, a.regs, zero_count, a.stream, a.device); break;
    default: return hipErrorInvalidValue;
  }
  if (e != hipSuccess) return e;
  ChainRegs rd = a.regs;  // the Doppler FFT has no window of its own in the register file
  rd.window = nullptr;
  const int16_t* wd = reinterpret_cast<const int16_t*>(a.win_doppler);
  switch (a.log2nd) {
    case 8: e = launch_doppler_fx<8>(x1, mag, n_ch, nr, tile, twd, wd, a.log_lut, rd, a.stream, a.device); break;
    case 9: e = launch_doppler_fx<9>(x1, mag, n_ch, nr, tile, twd, wd, a.log_lut, rd, a.stream, a.device); break;
    case 10: e = launch_doppler_fx<10>(x1, mag, n_ch, nr, tile, twd, wd, a.log_lut, rd, a.stream, a.device); break;
    default: return hipErrorInvalidValue;
  }
  if (e != hipSuccess) return e;
  const int hr = a.regs.R + a.regs.G, hd = a.ref_d + a.guard_d;
  const size_t lds = 4 * ((size_t)(kTD + 2 * hd) * ((kTR + 2 * hr) | 1) +
                          (a.regs.cfar_mode ? 2 : 1) * ((size_t)(kTD + 2 * hd + 1) * (kTR + 1) + (size_t)(kTD + 2 * a.guard_d + 1) * (kTR + 1)));
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  const bool spec = a.regs.R == 8 && a.regs.G == 2 && a.ref_d == 8 && a.guard_d == 2;  // cfg 3 / cfg 5 windows: the strip walker
  if (spec && !a.force_tiled_cfar) {
    const uint32_t strips = (nr + walk_outw(10) - 1) / walk_outw(10);
    constexpr uint32_t SEG = RSP_WALK_SEG;
    const dim3 grid(n_ch * strips * (nd / SEG / 4));
#define RSP_WALK(MODE)                                                                                                   \
  hipLaunchKernelGGL((cfar2d_walk_kernel<8, 2, 8, 2, SEG, MODE, int32_t>), grid, dim3(256), 0, a.stream, mag, out, nd, nr, \
                     strips, a.regs.edge, 0.f, 0.f, det_list, det_cap, a.det_count, ch0, tile, a.regs, a.log2nr, \
                     rd_stream_words(n_ch, nd, nr, 8))
    if (a.regs.cfar_mode == 0) RSP_WALK(0);
    else if (a.regs.cfar_mode == 1) RSP_WALK(1);
    else RSP_WALK(2);
#undef RSP_WALK
    return hipGetLastError();
  }
  auto k = cfar2d_kernel<-1, -1, -1, -1, int32_t>;
  static LdsGrant granted;
  e = grant_lds(k, lds, a.device, granted);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k, dim3(n_ch * (nr / kTR) * (nd / kTD)), dim3(256), lds, a.stream, mag, out,
                     nd, nr, a.regs.R, a.regs.G, a.ref_d, a.guard_d, a.regs.edge, 0.f, 0.f, det_list, det_cap,
                     a.det_count, ch0, a.regs.cfar_mode, tile, a.regs, a.log2nr);
  return hipGetLastError();
}

// Channels are independent: the batch runs in chunks whose intermediates (range spectrum 8 B/cell + magnitude map
// 4 B/cell) fit the 256 MiB Infinity Cache together with the streams passing by, and every chunk reuses the SAME
// scratch memory, so the corner turn and the magnitude map are re-read from cache instead of HBM.
hipError_t launch_rd2d(const Rd2dLaunch& a) {
  if (a.n_ch == 0) return a.det_count ? hipMemsetAsync(a.det_count, 0, 2 * sizeof(uint32_t), a.stream) : hipSuccess;
  const uint32_t per = rd2d_chunk_channels(a.log2nr, a.log2nd, a.n_ch, a.chunk_bytes);
  for (uint32_t c0 = 0; c0 < a.n_ch; c0 += per) {
    const uint32_t n = a.n_ch - c0 < per ? a.n_ch - c0 : per;
    hipError_t e = a.fixed ? launch_rd2d_chunk_fx(a, c0, n) : launch_rd2d_chunk(a, c0, n);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

}  // namespace rsp
