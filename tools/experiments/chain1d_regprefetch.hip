// EXPERIMENT (round 3, measured slower than the plain kernel; not part of librspchain.so -- build a side library
// with tools/build_experiment.sh regprefetch [-DRSP_PIPE_ISSUE=0|1 -DRSP_PIPE_WPC=3|4], run tools/ab_experiment.sh).
//
// Software-pipelined form of the fused 1-D chain (F32, CA / GO / SO on quads): PERSISTENT workgroups that have
// their NEXT frame group's samples in flight -- in registers -- while they finish the current one.
//
// Why: the plain kernel (cfar_quad.hpp) loads, computes, then stores.  A workgroup issues its 32 KiB of loads once,
// at the top of its life, waits out the whole (loaded) HBM latency with nothing to do, and a CU's four workgroups
// keep on average one of them in that phase: ~32 KiB in flight per CU, ~4 TB/s with nothing re-read (round 2,
// DESIGN.md 3.1).  Here a workgroup walks frame groups g = blockIdx.x, + gridDim.x, ... and issues the loads of
// group g' = g + gridDim.x into a second register set while group g is still being worked on:
//   ISSUE = 0  at the top of the iteration: a whole iteration of prefetch distance, 32 more live VGPRs through
//              the FFT (three workgroups per CU)
//   ISSUE = 1  behind the magnitudes, when the FFT's 32 sample registers are dead: the prefetch set takes their
//              place (no extra registers: four workgroups per CU), the CFAR tail and the word stores run under
//              the loads
// The word stores are never waited for: the wait for the prefetch sits in front of them (PipeHooks::before_stores).
//
// Measured and dropped (tools/experiments/chain1d_ldsdma.hip): landing the next group in a second LDS region by
// LDS-DMA (global_load_lds_dwordx4).  Two 66-KiB workgroups per CU then: 51 us per 4096 x 4096 against 46.4 us
// for the plain kernel on the same box -- with no loads and no stores at all that structure still takes 38.5 us:
// at two waves per SIMD the chain's dependent LDS / barrier phases are latency-bound.  Occupancy, not bytes in
// flight, was the scarcer resource; registers are the only landing zone that does not cost a resident frame.
#include <hip/hip_runtime.h>

#include "../../rsp-chains_amd/csrc/cfar_quad.hpp"

namespace rsp {

// the persistent loop re-uses the frame's LDS: the previous group's tail may still be reading its CFAR images when
// a fast wave reaches the next group's first exchange write -- one barrier in front of that write (behind pass 0,
// whose arithmetic covers the skew between the waves)
struct PipeHooks : SideHooks {
  static constexpr bool kSerialQuads = true;
  f32x2 (&nx)[16];  // the prefetch registers
  __device__ __forceinline__ explicit PipeHooks(f32x2 (&n)[16]) : nx(n) {}
  // The prefetch has to have arrived before the next iteration reads it, and the wait for it must not catch the
  // word stores: the compiler waits vmcnt(0) for a load result whenever loads AND stores are pending (it treats the
  // counter as out of order across the two kinds).  So the wait sits IN FRONT of the word stores -- an empty asm
  // that "uses" the 16 registers: all that is outstanding there is the prefetch itself and the previous group's
  // stores, a whole iteration old.
  __device__ __forceinline__ void before_stores() {
#pragma unroll
    for (int e = 0; e < 16; ++e) asm volatile("" : "+v"(nx[e]));
  }
  template <int P>
  __device__ __forceinline__ void before_exchange() {
    if constexpr (P == 1) __syncthreads();
  }
};

// HAS_OUT: dense words are written (out != NULL).  A compile-time fact, because the counted wait depends on it: with
// the word stores behind a run-time test the compiler has to assume the path without them, on which the prefetch
// loads are the youngest operations, and waits vmcnt(0) -- for the stores too -- at the end of every iteration.
template <int M, bool SMALL, int ISSUE, int WPC, bool HAS_OUT>
__global__ void __launch_bounds__(wg_size(M), WPC * wg_size(M) / 256)
chain1d_pipe_kernel(const void* __restrict__ in, uint32_t* __restrict__ out, uint32_t n_frames, uint32_t n_groups,
                    ChainRegs rg, const void* __restrict__ tw, uint32_t* __restrict__ fcount,
                    uint2* __restrict__ fdet) {
  constexpr int N = 1 << M, T = threads_per_frame(M), FPW = frames_per_wg(M);
  using L = QuadLds<M, SMALL>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, fl = tid / T;
  const char* gbase = reinterpret_cast<const char*>(in);
  const uint32_t first_byte = (uint32_t)first_sample<M>(tid % T) * 8u;

  // the thread's 16 samples of group g (uniform base + one 32-bit per-thread offset, as the plain front end);
  // frames past the end of the batch (ragged last group) re-read the batch's first frame and are never stored
  auto load_group = [&](f32x2 (&dst)[16], uint32_t g) {
    const uint32_t frame = g * FPW + fl;
    const uint32_t voff = (frame < n_frames ? frame : 0u) * (uint32_t)(N * 8) + first_byte;
#pragma unroll
    for (int e = 0; e < 16; ++e)
      dst[e] = *reinterpret_cast<const f32x2*>(gbase + (size_t)voff + (size_t)sample_offset<M>(e) * 8u);
  };

  if constexpr (HAS_OUT) __builtin_assume(out != nullptr);
  else out = nullptr;
  f32x2 nx[16];
  PipeHooks hk(nx);
  hk.init(rg);
  // Every pass's base twiddles.  Loop-invariant, and re-reading them per group is not an option: a load issued
  // beside the prefetch would have the compiler wait vmcnt(0) at its first use (in pass 1) and drain the
  // prefetch there.  They are made to ARRIVE here (an empty asm that "uses" every register).
  TwAll<M> twb;
  twb.load(tid % T, reinterpret_cast<const f32x2*>(tw));
  twb.touch();
  uint32_t g = blockIdx.x;
  if (g < n_groups) load_group(nx, g);
  hk.before_stores();  // the first group has ARRIVED when the loop is entered: a wait left pending into the loop would be
                       // merged into its top and wait, every iteration, for the previous group's word stores
  for (; g < n_groups; g += gridDim.x) {
    const uint32_t frame = g * FPW + fl;
    const bool live = FPW == 1 || frame < n_frames;  // one frame per workgroup: the loop bound says it all
    const bool has_next = g + gridDim.x < n_groups;
    // Everything the body derives from the thread's position (a score of LDS and global addresses) is loop-
    // invariant; hoisted, it stays live across the whole loop and the kernel spills at four workgroups per CU.
    // An opaque copy of the position per iteration keeps those values as short-lived as in the plain kernel.
    int tau = tid % T;
    asm volatile("" : "+v"(tau));
    unsigned char* fbase = smem + fl * L::BYTES;
    f32x2 x[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) x[e] = nx[e];
    if (ISSUE == 0 && has_next) load_group(nx, g + gridDim.x);
    fft_f32_passes<M>(tau, reinterpret_cast<f32x2*>(fbase), twb, x, hk);
    float mg[16];
    magnitudes_f32<M>(x, rg.mag_mode, mg, hk);
    if (ISSUE == 1 && has_next) load_group(nx, g + gridDim.x);
    quad_tail<M, false, SMALL>(fbase, mg, tau, frame, live, rg, out, fcount, fdet, hk);
  }
}

static int resident_groups(int device, int per_cu) {
  static int cus[kMaxDevices] = {};
  const int d = device >= 0 && device < kMaxDevices ? device : 0;
  if (!cus[d]) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || n <= 0) n = 256;
    cus[d] = n;
  }
  return per_cu * cus[d];
}

static bool chain1d_pipe_supports(const Chain1dLaunch& a) {
  return !a.fixed && a.log2n == 12 && quad_tail_supports(a.log2n, a.regs) && !a.force_generic_tail &&
         !a.regs.window && !a.regs.send_cut;
}

template <int M, int ISSUE, int WPC>
static hipError_t launch_pipe_m(const Chain1dLaunch& a) {
  const uint32_t fpw = frames_per_wg(M);
  const uint32_t n_groups = (a.n_frames + fpw - 1) / fpw;
  auto go = [&](auto small_c, LdsGrant* granted2) -> hipError_t {
    constexpr bool SMALL = decltype(small_c)::value;
    const size_t lds = QuadLds<M, SMALL>::BYTES * fpw;
    LdsGrant& granted = granted2[a.out ? 1 : 0];
    const uint32_t resident = (uint32_t)resident_groups(a.device, WPC);
    const dim3 grid(n_groups < resident ? n_groups : resident);
    auto run = [&](auto k) -> hipError_t {
      hipError_t e = grant_lds(k, lds, a.device, granted);
      if (e != hipSuccess) return e;
      hipLaunchKernelGGL(k, grid, dim3(wg_size(M)), lds, a.stream, a.in, a.out, a.n_frames, n_groups, a.regs,
                         a.twiddles, a.frame_count, a.frame_det);
      return hipGetLastError();
    };
    return a.out ? run(chain1d_pipe_kernel<M, SMALL, ISSUE, WPC, true>) : run(chain1d_pipe_kernel<M, SMALL, ISSUE, WPC, false>);
  };
  static LdsGrant g2[2][2];  // per instantiation: [halo size][HAS_OUT]
  return a.regs.R + a.regs.G + 4 <= QuadHalo<true>::MAG ? go(std::true_type{}, g2[0]) : go(std::false_type{}, g2[1]);
}

#ifndef RSP_PIPE_ISSUE
#define RSP_PIPE_ISSUE 1
#endif
#ifndef RSP_PIPE_WPC
#define RSP_PIPE_WPC 4
#endif

// the hook compact.hip's launch_chain1d looks for (weak symbol; absent from the product library)
extern "C" int rsp_experiment_chain1d(const Chain1dLaunch* a, hipError_t* err) {
  if (!chain1d_pipe_supports(*a)) return 0;
  *err = launch_pipe_m<12, RSP_PIPE_ISSUE, RSP_PIPE_WPC>(*a);
  return 1;
}

}  // namespace rsp
