// Launchers of the fused 1-D chain kernels: FFT -> magnitude -> CFAR, one launch per batch of frames.
// Kernels: chain_front.hpp (shared front end), cfar_quad.hpp / cfar_cell.hpp / cfar_gos.hpp (the tails).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include "cfar_cell.hpp"
#include "cfar_gos.hpp"
#include "cfar_quad.hpp"

namespace rsp {

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
  const size_t lds = kPartFixed ? (size_t)FixedRom<M, kPartFx>::total(lay.frame_bytes) : (size_t)lay.frame_bytes * fpw;
  auto go = [&](auto k, LdsGrant& g) -> hipError_t {
    hipError_t err = grant_lds(k, lds, a.device, g);
    if (err != hipSuccess) return err;
    hipExtLaunchKernelGGL(k, dim3(grid), dim3(wg_size(M)), lds, a.stream, a.ev_start, a.ev_stop, 0, a.in, a.out, a.n_frames, a.regs, lay,
                       a.twiddles, a.log_lut, a.frame_count, a.frame_det);
    return hipGetLastError();
  };
  return with_fx(a, [&](auto fx) -> hipError_t {
    constexpr int FX = decltype(fx)::value;
    static LdsGrant g3[3];
    if (a.regs.R <= 32) return go(chain1d_gos_kernel<M, kPartFixed, 0, FX>, g3[0]);
    // 64-cell window: run = ceil((N + 2 G + 65) / T) starts per thread; the split path is instantiated for the run of
    // the usual guard sizes at each frame size (gos_big_run), the one-window path (scratch-memory picks) serves the rest
    const bool split = lay.run == gos_big_run(M) && RSP_GOS_SPLIT;
    if (split) return go(chain1d_gos_kernel<M, kPartFixed, 1, FX>, g3[1]);
    if constexpr (M <= 11) return go(chain1d_gos_kernel<M, kPartFixed, 2, FX>, g3[2]);
    return hipErrorInvalidValue;
  });
}

template <int M>
static hipError_t launch_quad(const Chain1dLaunch& a) {
  const uint32_t fpw = frames_per_wg(M);
  const uint32_t grid = (a.n_frames + fpw - 1) / fpw;
  return with_fx(a, [&](auto fx) -> hipError_t {
    constexpr int FX = decltype(fx)::value;
    auto go = [&](auto small_c, auto short_c, LdsGrant& granted) -> hipError_t {
      constexpr bool SMALL = decltype(small_c)::value, SHORTW = decltype(short_c)::value;
      using L = QuadLds<M, SMALL, SHORTW>;
      const size_t lds = kPartFixed ? (size_t)FixedRom<M, FX>::total(L::BYTES) : (size_t)L::BYTES * fpw;
      auto k = chain1d_quad_kernel<M, kPartFixed, FX, SMALL, SHORTW>;
      hipError_t e = grant_lds(k, lds, a.device, granted);
      if (e != hipSuccess) return e;
      hipExtLaunchKernelGGL(k, dim3(grid), dim3(wg_size(M)), lds, a.stream, a.ev_start, a.ev_stop, 0, a.in, a.out, a.n_frames,
                         a.regs, a.twiddles, a.log_lut, a.frame_count, a.frame_det);
      return hipGetLastError();
    };
    static LdsGrant g3[3];
    // fp32 windows of at most 16 cells: prefixes relative to 16-cell blocks (QuadLds: SHORTW)
    if constexpr (!kPartFixed) {
      if (a.regs.R <= 16) return go(std::true_type{}, std::true_type{}, g3[2]);
    }
    return a.regs.R + a.regs.G + 4 <= QuadHalo<true>::MAG ? go(std::true_type{}, std::false_type{}, g3[0])
                                                           : go(std::false_type{}, std::false_type{}, g3[1]);
  });
}

template <int M>
static hipError_t launch_m(const Chain1dLaunch& a) {
  if (a.regs.algorithm == 1) return launch_gos<M>(a);
  if (quad_tail_supports(M, a.regs) && !a.force_generic_tail) return launch_quad<M>(a);
  const uint32_t fpw = frames_per_wg(M);
  const uint32_t grid = (a.n_frames + fpw - 1) / fpw;
  const size_t lds = kPartFixed ? (size_t)FixedRom<M, kPartFx>::total(FrameLds<M>::BYTES) : (size_t)FrameLds<M>::BYTES * fpw;
  return with_fx(a, [&](auto fx) -> hipError_t {
    constexpr int FX = decltype(fx)::value;
    static LdsGrant granted;
    auto k = chain1d_kernel<M, kPartFixed, FX>;
    hipError_t e = grant_lds(k, lds, a.device, granted);
    if (e != hipSuccess) return e;
    hipExtLaunchKernelGGL(k, dim3(grid), dim3(wg_size(M)), lds, a.stream, a.ev_start, a.ev_stop, 0, a.in, a.out, a.n_frames,
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
    case 14: return launch_m<14>(a);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace rsp
