// Side-build scaffolding of the chain kernels, in ONE place.  Nothing here exists in the product
// library: every hook below is an empty inline function unless the file is compiled with one of
//   -DRSP_ABLATE      phase ablation (tools/ablate.sh, tools/pmc_ablate.sh): a run-time mask, read from
//                     ChainRegs::sub_window (unused by the CA-family kernels; the API fills it from
//                     RSP_ABLATE_MASK), switches phases off -- bit 0 butterflies, 1 CFAR cells, 2 scan,
//                     3 FFT exchanges, 4 magnitude; quad kernel also: 5 loads from 64 L2-resident frames
//                     instead of the batch, 6 no word stores (5 + 6 = the kernel's compute alone)
//   -DRSP_STAMP       s_memtime at the phase boundaries of a few workgroups, printed by one wave each
//                     (tools/stamp.sh): where a workgroup's lifetime goes while the chip is loaded
//   -DRSP_COUNT_PATH  ONE path of the run-time mode switches (CA, no peak grouping, JPL, no detection
//                     slots), so that static instruction counts equal dynamic ones (tools/count_insts.sh)
//   -DRSP_DBG_OUT=n   FIXED16 quad tail: expose one of its inputs instead of the words (debugging)
// The kernels take the hooks as a policy object (`SideHooks`), so a side build changes no kernel source.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "chain_regs.hpp"

namespace rsp {

// what every hook policy offers by default
struct PlainHooks {
  static constexpr bool kSerialQuads = false;  // quad tail, F32: schedule the four quads of a thread one after the other
  static __device__ __forceinline__ void barrier() { __syncthreads(); }       // the barrier inside an FFT exchange
  __device__ __forceinline__ void before_stores() {}                            // quad tail: in front of the word stores
  template <int P> __device__ __forceinline__ void before_exchange() {}         // in front of exchange P's LDS writes
  template <int P> __device__ __forceinline__ void after_exchange_barrier() {}  // behind the barrier of exchange P
};

struct SideHooks : PlainHooks {
#ifdef RSP_ABLATE
  int mask = 0;
  __device__ __forceinline__ void init(const ChainRegs& rg) { mask = rg.sub_window; }
  __device__ __forceinline__ bool off(int bit) const { return (mask & (1 << bit)) != 0; }
#else
  __device__ __forceinline__ void init(const ChainRegs&) {}
  __device__ __forceinline__ bool off(int) const { return false; }
#endif
#ifdef RSP_STAMP
  uint64_t st[12] = {};  // wave-uniform (SGPR) values
  __device__ __forceinline__ void stamp(int i) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    st[i] = __builtin_amdgcn_s_memtime();
  }
  __device__ __forceinline__ void report() const {
    if ((blockIdx.x & 255) == 77 && (threadIdx.x & 63) == 0)
      printf("stamp wg %u wave %u: load %llu p0 %llu x1 %llu p1 %llu x2 %llu p2mag %llu magw %llu scan %llu cells %llu store %llu total %llu\n",
             blockIdx.x, threadIdx.x >> 6, st[1] - st[0], st[2] - st[1], st[3] - st[2], st[4] - st[3], st[5] - st[4],
             st[7] - st[5], st[8] - st[7], st[9] - st[8], st[10] - st[9], st[11] - st[10], st[11] - st[0]);
  }
#else
  __device__ __forceinline__ void stamp(int) {}
  __device__ __forceinline__ void report() const {}
#endif
};

// the stand-in of kernels that take no hooks (2-D chain)
struct NoHooks : PlainHooks {
  __device__ __forceinline__ bool off(int) const { return false; }
  __device__ __forceinline__ void stamp(int) {}
};

#ifdef RSP_COUNT_PATH
constexpr bool kCountPath = true;
#else
constexpr bool kCountPath = false;
#endif

// A/B switches of kept optimisations (tools/build_variant.sh x.so -DRSP_NO_NT ...): the product builds with all three on
#ifdef RSP_NO_NT
constexpr bool kStreamHints = false;   // non-temporal frame loads / word stores
#else
constexpr bool kStreamHints = true;
#endif
#ifdef RSP_NO_PRIO
constexpr bool kWavePrio = false;      // s_setprio around a workgroup's frame loads and word stores
#else
constexpr bool kWavePrio = true;
#endif
#ifdef RSP_EXCHANGE_READ2
constexpr bool kExchangeB64 = false;   // FFT exchange reads as ds_read_b64 (asm) instead of the compiler's ds_read2_b64 pairing
#else
constexpr bool kExchangeB64 = true;
#endif

}  // namespace rsp
