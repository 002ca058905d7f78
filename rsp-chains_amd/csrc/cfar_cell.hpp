// Per-cell CFAR tail of the fused 1-D chain (the round-1 kernel): CASH and window sizes that are not
// multiples of 4; rsp_chain_set_option(RSP_OPT_FORCE_GENERIC_TAIL) selects it for A/B runs.  Magnitudes and
// block-relative prefix sums in padded LDS images (x + x / 16), one cell per lane and step.  CFAR sliding
// sums come from block-relative prefix sums held in LDS (blocks of 256 cells = one 16-lane row of the
// scan), so a window sum is a difference of two nearby prefixes plus at most one block total: exact for
// integers, no long-range cancellation for fp32.
#pragma once
#include "chain_front.hpp"

namespace rsp {

template <int M, bool FIXED, int FX>
__global__ void __launch_bounds__(wg_size(M))
chain1d_kernel(const void* __restrict__ in, uint32_t* __restrict__ out, uint32_t n_frames,
               ChainRegs rg, const void* __restrict__ tw, const int16_t* __restrict__ log_lut,
               uint32_t* __restrict__ fcount, uint2* __restrict__ fdet) {
  constexpr int N = 1 << M, T = threads_per_frame(M), FPW = frames_per_wg(M);
  using L = FrameLds<M>;
  using V = typename std::conditional<FIXED, int, float>::type;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int fl = tid / T, tau = tid % T;
  const uint32_t frame = blockIdx.x * FPW + fl;
  const bool live = frame < n_frames;  // dead frames still walk every barrier
  unsigned char* fbase = smem + fl * L::BYTES;

  V mg[16];
  SideHooks hk;
  hk.init(rg);
  front_end<M, FIXED, V, FX>(in, frame, live, tau, fbase, rg, tw, log_lut,
                             reinterpret_cast<uint2*>(smem + (FIXED ? FixedRom<M, FX>::off(L::BYTES) : 0)), mg, hk);

  // ---- magnitudes to LDS in natural bin order ----
  V* mag = reinterpret_cast<V*>(fbase + L::MAG_OFF);
  V* pb = reinterpret_cast<V*>(fbase + L::PB_OFF);
  V* bs = reinterpret_cast<V*>(fbase + L::BS_OFF) + L::BS_HALO;  // bs[-16] .. bs[N/16 + 15]
  // Block length of the block-relative prefixes: 256 cells, or -- fp32, every summed run (window, CASH sub-window) at
  // most 16 cells long -- 16 cells = the thread's own chunk.  A window sum is a difference of two prefixes and its
  // rounding error scales with the prefix, not with the window: with 256-cell blocks an 8-cell window needed a 2.5x
  // looser tolerance than every other geometry.  A run of <= 16 cells crosses at most one 16-cell block edge.
  const int run_len = rg.cfar_mode == 3 ? rg.sub_window : rg.R;
  const bool short_blocks = !FIXED && run_len <= 16;
  const int sh = short_blocks ? 4 : 8;
  uint32_t* det_cnt = reinterpret_cast<uint32_t*>(fbase + L::DET_OFF);
  uint2* det_stage = reinterpret_cast<uint2*>(fbase + L::DET_OFF + 8);
  const bool wrap = rg.edge != 0;
  __syncthreads();  // every thread is done reading the FFT image this overlays
  write_mag<M, V>(mag, 16, tau, mg, rg.rev_order != 0);
  if (tau == 0) *det_cnt = 0u;
  __syncthreads();

  // ---- block-relative exclusive prefix sums: thread owns cells 16 tau .. 16 tau + 15 ----
  if (!hk.off(2)) {
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
    const V exc = short_blocks ? V(0) : row_shr<1>(inc);
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
    if (short_blocks) {  // one block per thread: its total, and the halo images of the first / last 16 blocks
      bs[tau] = acc;
      if (tau < 16) bs[tau + T] = wrap ? acc : V(0);
      if (tau >= T - 16) bs[tau - T] = wrap ? acc : V(0);
    } else if ((tau & 15) == 15) {
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
        if (T % 256 == 0 || short_blocks) {  // block of cell k is a per-thread constant + j T / block (T is a multiple of 16)
          bu0 = (xu0 >> sh) + j * (T >> sh);
          bu1 = (xu1 >> sh) + j * (T >> sh);
          c0 = (xv0 >> sh) != (xu0 >> sh);
          c1 = (xv1 >> sh) != (xu1 >> sh);
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
            // consecutive sub-windows share a boundary: its prefix and block index are carried, not looked up again
            V pu = pb[pb_slot(a)];
            int bu = a >> sh;
            for (int s0 = 0; s0 + rg.sub_window <= R; s0 += rg.sub_window) {
              const int v = a + s0 + rg.sub_window, bv = v >> sh;
              const V pv = pb[pb_slot(v)];
              V ss = pv - pu;
              if (bv != bu) ss += bs[bu];
              b = first ? ss : (ss > b ? ss : b);
              first = false;
              pu = pv;
              bu = bv;
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
    if constexpr (kCountPath) {
      cells(I0{}, std::false_type{});
    } else if (hk.off(1)) {
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

}  // namespace rsp
