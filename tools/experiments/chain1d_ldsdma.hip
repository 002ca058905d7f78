// EXPERIMENT (round 3, measured slower than the plain kernel; not part of librspchain.so -- build a side library
// with tools/build_experiment.sh ldsdma [-DRSP_PIPE_NOLOAD -DRSP_PIPE_NOSTORE], run tools/ab_experiment.sh).
// 4096 x 4096, same box: 51.0 us against 46.4 us plain; without its loads 43.1, without its stores 46.4, without
// both 38.5 us -- two workgroups per CU cannot hide the chain's dependent LDS / barrier phases.
//
// Software-pipelined form of the fused 1-D chain (F32, CA / GO / SO on quads): PERSISTENT workgroups that
// have their NEXT frame group's samples in flight while they transform the current one.
//
// Why: the plain kernel (cfar_quad.hpp) loads, computes, then stores; a workgroup issues its 32 KiB of loads
// once, at the top of its life, and four workgroups per CU keep on average one of them in its load phase --
// ~32 KiB in flight per CU, which at the loaded HBM latency is the ~4 TB/s measured in round 2 with nothing
// re-read.  Here a workgroup owns TWO LDS regions:
//   land  FPW frames x 2^M beats, unpadded -- the destination of LDS-DMA loads (global_load_lds_dwordx4: no
//         VGPRs, 1 KiB per wave-instruction, laid down in lane order)
//   work  the FFT image / CFAR images of the plain kernel
// and walks frame groups g = blockIdx.x, + gridDim.x, ...:
//   wait for DMA(g) | samples land -> registers | pass 0 | exchange 1: write, BARRIER, issue DMA(g') into land |
//   rest of the FFT | magnitude | quad CFAR tail | 16-byte word stores, never waited for
// The vector-memory counter retires in order, so the wait at the top is counted: the word stores of the
// previous group are younger than the DMA and stay in flight (s_waitcnt vmcnt(4): four stores per thread).
// Two 66-KiB workgroups per CU: one frame group in flight per workgroup ALL the time (64 KiB per CU) beside
// two being transformed.  Barriers inside the loop are raw s_barrier + lgkmcnt(0): a __syncthreads() may
// carry a vector-memory wait, which would drain the DMA.
#include <hip/hip_runtime.h>

#include "../../rsp-chains_amd/csrc/cfar_quad.hpp"

namespace rsp {

// one LDS-DMA piece: 64 lanes x 16 B from sbase + voff (per lane) to LDS byte address lds_dst + 16 lane.
// M0 (the destination base) is compiler-reserved and not preserved around the statement: written and
// restored inside it (cdna_hip_programming.md, LDS-DMA recipe).
__device__ __forceinline__ void glds16(uint32_t voff, const void* sbase, uint32_t lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(sbase), "s"(lds_dst)
      : "memory");
}

typedef __attribute__((address_space(3))) unsigned char lds_byte;

template <int M>
struct PipeHooks {
  static constexpr int N = 1 << M, FPW = frames_per_wg(M), NW = wg_size(M) / 64;
  static constexpr int LAND_BYTES = FPW * N * 8;
  static constexpr int PIECES = LAND_BYTES / 1024;  // 1 KiB per wave-instruction
  static_assert(PIECES % NW == 0, "whole pieces per wave");

  const char* in;        // batch base (uniform)
  uint32_t land_lds;     // LDS byte address of the landing region (uniform)
  uint32_t next_group;   // group to prefetch behind exchange 1's barrier (>= n_groups: none)
  uint32_t n_groups, n_frames;

  // every lane's 16 bytes of piece p of group g; frames past the end of the batch (ragged last group)
  // re-read the batch's first frame and are never stored
  __device__ __forceinline__ void issue(uint32_t g) const {
#ifdef RSP_PIPE_NOLOAD  // side build (tools/build_variant_pipe.sh): no loads -- the kernel's compute + store floor
    return;
#endif
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < PIECES / NW; ++k) {
      const uint32_t piece = (uint32_t)(wave + NW * k);
      uint32_t byte = piece * 1024u;  // offset of the piece inside the group
      uint64_t gbyte = (uint64_t)g * LAND_BYTES + byte;
      if constexpr (FPW > 1) {
        const uint32_t frame = g * FPW + byte / (N * 8u);
        if (frame >= n_frames) gbyte = byte % (N * 8u);
      }
      glds16((uint32_t)lane * 16u, in + gbyte, land_lds + byte);
    }
  }

  static __device__ __forceinline__ void barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
  template <int P>
  __device__ __forceinline__ void after_exchange_barrier() {
    // every thread read its samples out of `land` before it wrote exchange 1: the region is free
    if constexpr (P == 1) {
      if (next_group < n_groups) issue(next_group);
    }
  }
  template <int P> __device__ __forceinline__ void before_exchange() {}
  __device__ __forceinline__ void before_stores() {}
  static constexpr bool kSerialQuads = false;
  __device__ __forceinline__ bool off(int) const { return false; }
  __device__ __forceinline__ void stamp(int) {}
  __device__ __forceinline__ void report() const {}
};

template <int M, bool SMALL>
__global__ void __launch_bounds__(wg_size(M))
chain1d_pipe_kernel(const void* __restrict__ in, uint32_t* __restrict__ out, uint32_t n_frames, uint32_t n_groups,
                    ChainRegs rg, const void* __restrict__ tw, uint32_t* __restrict__ fcount,
                    uint2* __restrict__ fdet) {
  constexpr int N = 1 << M, T = threads_per_frame(M), FPW = frames_per_wg(M);
  using L = QuadLds<M, SMALL>;
  using HK = PipeHooks<M>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, fl = tid / T, tau = tid % T;
  unsigned char* land = smem;
  unsigned char* fbase = smem + HK::LAND_BYTES + fl * L::BYTES;

  HK hk;
  hk.in = reinterpret_cast<const char*>(in);
  hk.land_lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_byte*)land);
  hk.n_groups = n_groups;
  hk.n_frames = n_frames;

  uint32_t g = blockIdx.x;
  if (g < n_groups) hk.issue(g);
  // all passes' base twiddles: loop-invariant.  They are made to ARRIVE here (an empty asm that "uses" every
  // register): left pending into the loop, the compiler's own wait for them would sit in front of pass 0 of EVERY
  // iteration and there also wait for the previous group's word stores.
  TwAll<M> twb;
  twb.load(tau, reinterpret_cast<const f32x2*>(tw));
  twb.touch();
  bool first = true;
  for (; g < n_groups; g += gridDim.x) {
    // this wave's pieces of group g have landed (the previous group's 4 word stores per thread may still be in
    // flight: they are younger), then everybody's
    if (first || !out) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    first = false;
    HK::barrier();
    const uint32_t frame = g * FPW + fl;
    const bool live = frame < n_frames;
    f32x2 x[16];
    {
      const f32x2* src = reinterpret_cast<const f32x2*>(land) + fl * N + first_sample<M>(tau);
#pragma unroll
      for (int e = 0; e < 16; ++e) x[e] = src[sample_offset<M>(e)];
    }
    hk.next_group = g + gridDim.x;
    fft_f32_passes<M>(tau, reinterpret_cast<f32x2*>(fbase), twb, x, hk);
    float mg[16];
    magnitudes_f32<M>(x, rg.mag_mode, mg, hk);
#ifdef RSP_PIPE_NOSTORE  // side build: no word stores (the run-time test keeps the tail's arithmetic alive)
    quad_tail<M, false, SMALL>(fbase, mg, tau, frame, live, rg, rg.R == 12345 ? out : nullptr, fcount, fdet, hk);
#else
    quad_tail<M, false, SMALL>(fbase, mg, tau, frame, live, rg, out, fcount, fdet, hk);
#endif
    // the next group's first barrier (top of the loop) also separates this tail's last LDS reads from the
    // next FFT's first exchange write
  }
}

static int pipe_grid(int device, uint32_t n_groups) {
  static int cus[kMaxDevices] = {};
  const int d = device >= 0 && device < kMaxDevices ? device : 0;
  if (!cus[d]) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || n <= 0) n = 256;
    cus[d] = n;
  }
  const uint32_t resident = 2u * (uint32_t)cus[d];  // two 66-KiB workgroups per CU
  return (int)(n_groups < resident ? n_groups : resident);
}

static bool chain1d_pipe_supports(const Chain1dLaunch& a) {
  return !a.fixed && a.log2n == 12 && quad_tail_supports(a.log2n, a.regs) && !a.force_generic_tail &&
         !a.regs.window && !a.regs.send_cut;
}

template <int M>
static hipError_t launch_pipe_m(const Chain1dLaunch& a) {
  const uint32_t fpw = frames_per_wg(M);
  const uint32_t n_groups = (a.n_frames + fpw - 1) / fpw;
  auto go = [&](auto small_c, LdsGrant& granted) -> hipError_t {
    constexpr bool SMALL = decltype(small_c)::value;
    const size_t lds = PipeHooks<M>::LAND_BYTES + QuadLds<M, SMALL>::BYTES * fpw;
    auto k = chain1d_pipe_kernel<M, SMALL>;
    hipError_t e = grant_lds(k, lds, a.device, granted);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3(pipe_grid(a.device, n_groups)), dim3(wg_size(M)), lds, a.stream, a.in, a.out, a.n_frames,
                       n_groups, a.regs, a.twiddles, a.frame_count, a.frame_det);
    return hipGetLastError();
  };
  static LdsGrant g2[2];
  return a.regs.R + a.regs.G + 4 <= QuadHalo<true>::MAG ? go(std::true_type{}, g2[0]) : go(std::false_type{}, g2[1]);
}

// the hook compact.hip's launch_chain1d looks for (weak symbol; absent from the product library)
extern "C" int rsp_experiment_chain1d(const Chain1dLaunch* a, hipError_t* err) {
  if (!chain1d_pipe_supports(*a)) return 0;
  *err = launch_pipe_m<12>(*a);
  return 1;
}

}  // namespace rsp
