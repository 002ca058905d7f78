// EXPERIMENT (round 3, measured slower than the plain kernel; not part of librspchain.so -- build a side library with
// tools/build_experiment.sh regprefetch [-DRSP_PIPE_ISSUE=0|1 -DRSP_PIPE_WPC=3|4], run tools/ab_experiment.sh on the GPU box).
//
// Software-pipelined form of the fused 1-D chain (F32, CA / GO / SO on quads, 4096 points): PERSISTENT workgroups that
// have their NEXT frame's samples in flight -- in registers -- while they finish the current one.
//   ISSUE = 1 (default)  the loads of frame g + gridDim.x are issued behind the magnitudes of frame g, when the FFT's 32
//                        sample registers are dead, and waited for in front of frame g's word stores: the CFAR tail
//                        runs under the loads, the stores are never waited for (PipeHooks::before_stores)
//   ISSUE = 0            at the top of the iteration (a whole iteration of prefetch distance, 32 more live VGPRs)
// What it took to make the compiler keep the prefetch in flight (each item measured or read off the ISA):
//   * the loads are inline asm into registers PINNED to v[96:127]; plain loads were spilled to scratch on arrival or
//     moved mid-tail by the register allocator (a move is a use: it waits), and a kernel that names AGPRs gets its 128
//     registers split 64 / 64;
//   * the compiler waits vmcnt(0) for any load result while stores are pending (it treats the counter as out of order
//     across the two kinds), so the one wait sits in front of the stores, where only the prefetch is outstanding;
//   * no vector-memory load inside the loop: twiddles come from two small LDS tables (pass 0: W^low, squared up to
//     W^8low -- 2-8 ulp, about twice the product kernel's threshold error), scratch must be ZERO (a reload is a load);
//   * position-derived addresses are recomputed per iteration (an opaque copy of tau), peak grouping and sendCut are
//     left to the plain kernel, the tail handles one quad at a time: 128 VGPRs, no scratch, four workgroups per CU.
// Result (4096 x 4096, same box, tools/ab_experiment.sh): 50.0-51.5 us against 45.8-46.6 us for the plain kernel;
// three workgroups per CU with the compiler's own (draining) waits: 52-55 us.  The plain kernel's compute alone
// (ablation mask 96) is 32.5 us at four workgroups per CU: its instruction issue -- VALU plus LDS transfers, which
// serialise on a SIMD (tools/valubench.hip) -- is the bound, HBM latency costs it 7.7 us and the stores 5.6 us, and the
// hardware's own multiplexing of four independent workgroups hides that better than one in-order instruction stream
// per wave that also has to carry the next frame.
#include <hip/hip_runtime.h>

#include "../../rsp-chains_amd/csrc/cfar_quad.hpp"

namespace rsp {

// the prefetch registers: v[96:127], the top of the 128 a wave has at four workgroups per CU
#define RSP_PIN_0 "{v[96:97]}"
#define RSP_PIN_1 "{v[98:99]}"
#define RSP_PIN_2 "{v[100:101]}"
#define RSP_PIN_3 "{v[102:103]}"
#define RSP_PIN_4 "{v[104:105]}"
#define RSP_PIN_5 "{v[106:107]}"
#define RSP_PIN_6 "{v[108:109]}"
#define RSP_PIN_7 "{v[110:111]}"
#define RSP_PIN_8 "{v[112:113]}"
#define RSP_PIN_9 "{v[114:115]}"
#define RSP_PIN_10 "{v[116:117]}"
#define RSP_PIN_11 "{v[118:119]}"
#define RSP_PIN_12 "{v[120:121]}"
#define RSP_PIN_13 "{v[122:123]}"
#define RSP_PIN_14 "{v[124:125]}"
#define RSP_PIN_15 "{v[126:127]}"
#define RSP_PIN(e) RSP_PIN_##e

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
    asm volatile("s_waitcnt vmcnt(0)"
                 : "=" RSP_PIN(0)(nx[0]), "=" RSP_PIN(1)(nx[1]), "=" RSP_PIN(2)(nx[2]), "=" RSP_PIN(3)(nx[3]), "=" RSP_PIN(4)(nx[4]),
                   "=" RSP_PIN(5)(nx[5]), "=" RSP_PIN(6)(nx[6]), "=" RSP_PIN(7)(nx[7]), "=" RSP_PIN(8)(nx[8]), "=" RSP_PIN(9)(nx[9]),
                   "=" RSP_PIN(10)(nx[10]), "=" RSP_PIN(11)(nx[11]), "=" RSP_PIN(12)(nx[12]), "=" RSP_PIN(13)(nx[13]),
                   "=" RSP_PIN(14)(nx[14]), "=" RSP_PIN(15)(nx[15])
                 : "0"(nx[0]), "1"(nx[1]), "2"(nx[2]), "3"(nx[3]), "4"(nx[4]), "5"(nx[5]), "6"(nx[6]), "7"(nx[7]), "8"(nx[8]),
                   "9"(nx[9]), "10"(nx[10]), "11"(nx[11]), "12"(nx[12]), "13"(nx[13]), "14"(nx[14]), "15"(nx[15]));
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

  // The thread's 16 samples of group g, loaded by inline asm INTO ACCUMULATION REGISTERS (gfx950: one 512-entry file,
  // a load may name an AGPR as its destination).  Two reasons: (1) left to the compiler, the register allocator
  // defragments its VGPRs in the middle of the CFAR tail by MOVING the 32 long-lived prefetch registers (a copy is a
  // use: it waits for the loads and drains the prefetch; with plain loads it also spilled nine of the sixteen pairs
  // straight to scratch); (2) the compiler does not count asm loads, so no wait of its own can catch them: the one
  // wait is the asm statement in PipeHooks::before_stores.  The 32 registers are PINNED to the top of the kernel's
  // 128 (constraint "{v[96:97]}" ...) in the load and in the wait statement, so that the allocator has no reason to
  // move them in between -- a move there would read registers whose loads have not landed, which is why the build
  // recipe greps the ISA for it (tools/build_experiment.sh).  (AGPR destinations would be the clean form, but a kernel
  // that names AGPRs gets its register budget split 64 / 64 by this compiler.)
  // Offsets: 13-bit signed immediates, so the 16 loads use 8 scalar bases 4096 B apart.
  static_assert(FPW == 1 && M == 12, "one 4096-point frame per workgroup: uniform base address, samples 256 apart");
  auto load_group = [&](f32x2 (&dst)[16], uint32_t g) {
    const char* fb = gbase + (size_t)g * (N * 8);
#define RSP_LOAD2(e, e1)                                                                                       \
    asm volatile("global_load_dwordx2 %0, %2, %3\n\tglobal_load_dwordx2 %1, %2, %3 offset:2048"                  \
                 : "=" RSP_PIN(e)(dst[e]), "=" RSP_PIN(e1)(dst[e1])                                             \
                 : "v"(first_byte), "s"(fb + (size_t)((e) / 2) * 4096))
    RSP_LOAD2(0, 1); RSP_LOAD2(2, 3); RSP_LOAD2(4, 5); RSP_LOAD2(6, 7); RSP_LOAD2(8, 9); RSP_LOAD2(10, 11); RSP_LOAD2(12, 13);
    RSP_LOAD2(14, 15);
  };

  if constexpr (HAS_OUT) __builtin_assume(out != nullptr);
  else out = nullptr;
  f32x2 nx[16];
  PipeHooks hk(nx);
  hk.init(rg);
  // Twiddles without registers held across the loop and without vector-memory loads inside it (a load beside the
  // prefetch would have the compiler wait vmcnt(0) at its first use): pass 0 reads ONE value per thread, W^low, from a
  // 2-KiB LDS copy and squares it up to W^2low, W^4low, W^8low (6 packed operations; 2-8 ulp instead of the table's
  // 0.5); pass 1's table (16 entries x 32 B at 4096 points) is an LDS copy too.
  static_assert(plan_np(M) == 3 && plan_lo(M, 1) == 4, "written for the 16 x 16 x 16 plan");
  f32x2* tw1_lds = reinterpret_cast<f32x2*>(smem + (size_t)L::BYTES * FPW);   // pass 1: 16 entries x {W, W^2, W^4, W^8}
  f32x2* tw0_lds = tw1_lds + 64;                                             // pass 0: W^low, low < 256
  const f32x2* twf = reinterpret_cast<const f32x2*>(tw);
  if (tid < 4 * 16) tw1_lds[tid] = twf[tw_table_offset(M, 1) + tid];
  static_assert(wg_size(M) == (1 << plan_lo(M, 0)), "one pass-0 twiddle per thread");
  tw0_lds[tid] = twf[tw_table_offset(M, 0) + 4 * tid];
  __syncthreads();
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
    TwAll<M> twb;
    const f32x2 w_low = tw0_lds[tau];
    twb.p0.w[0][0] = w_low;
    twb.p0.w[0][1] = cmul(w_low, w_low);
    twb.p0.w[0][2] = cmul(twb.p0.w[0][1], twb.p0.w[0][1]);
    twb.p0.w[0][3] = cmul(twb.p0.w[0][2], twb.p0.w[0][2]);
    {
      const f32x4* e1 = reinterpret_cast<const f32x4*>(tw1_lds + 4 * (tau & 15));
      const f32x4 lo = e1[0], hi = e1[1];
      twb.p1.w[0][0] = {lo.x, lo.y};
      twb.p1.w[0][1] = {lo.z, lo.w};
      twb.p1.w[0][2] = {hi.x, hi.y};
      twb.p1.w[0][3] = {hi.z, hi.w};
    }
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
         !a.regs.window && !a.regs.send_cut && !a.regs.peak_grouping;
}

template <int M, int ISSUE, int WPC>
static hipError_t launch_pipe_m(const Chain1dLaunch& a) {
  const uint32_t fpw = frames_per_wg(M);
  const uint32_t n_groups = (a.n_frames + fpw - 1) / fpw;
  auto go = [&](auto small_c, LdsGrant* granted2) -> hipError_t {
    constexpr bool SMALL = decltype(small_c)::value;
    const size_t lds = QuadLds<M, SMALL>::BYTES * fpw + 512 + 2048;
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
