// In-LDS FFT building blocks for gfx950 (wave64, 160 KiB LDS/CU).
//
// Replaces the reference's SDF-FFT block (AXI4FFTBlock, constructed at
// /root/reference/src/main/scala/FftMagCfarChain.scala:33 from
// FFTParams.fixed(...) :78-90; generator sources are an empty submodule).
//
// Formulation: an N = 2^M point decimation-in-frequency FFT computed in place
// over the sample index i (M bits).  The index is cut into 2..4 bit-fields of
// width <= 4, most significant first; one *pass* does the radix-2^w butterflies
// of one field entirely in registers (16 samples per thread), passes exchange
// through LDS.  The result lands bit-reversed (sample i holds bin bitrev_M(i))
// and is consumed in that order by the magnitude stage, i.e. the reorder the
// reference gets from useBitReverse = true costs nothing here.
//
//  * F32: a pass is a 2^w-point DFT with constant twiddles, then one multiply
//    per sample by W_{2^(lo+w)}^(low*q) built from <= 4 table reads.
//  * FIXED16: the same butterfly graph evaluated radix-2 stage by radix-2 stage
//    with a Q2.14 twiddle per butterfly and a 1-bit trim per stage -- exactly the
//    dataflow of the SDF pipeline, so results are bit-identical to the spec
//    (oracle/rsp_oracle.c orc_fft_fixed) whatever the pass split.
//
// LDS image: sample i lives at slot i + (i >> 4) (one pad slot per 16), which
// keeps the three access shapes used here (unit stride, 16-sample runs at a
// 256-sample pitch, 16 consecutive samples per lane) at <= 2-way bank conflicts.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "chain_regs.hpp"
#include "side_build.hpp"

namespace rsp {

typedef float f32x2 __attribute__((ext_vector_type(2)));

__host__ __device__ constexpr int plan_np(int M) { return M <= 8 ? 2 : (M <= 12 ? 3 : 4); }
// field widths per pass, most significant field first
__host__ __device__ constexpr int plan_w(int M, int p) {
  switch (M) {
    case 8: return p < 2 ? 4 : 0;
    case 9: return p < 3 ? 3 : 0;
    case 10: return p == 0 ? 4 : (p < 3 ? 3 : 0);
    case 11: return p < 2 ? 4 : (p == 2 ? 3 : 0);
    case 12: return p < 3 ? 4 : 0;
    case 13: return p == 0 ? 4 : (p < 4 ? 3 : 0);
    case 14: return p < 2 ? 4 : (p < 4 ? 3 : 0);
    default: return 0;
  }
}
__host__ __device__ constexpr int plan_lo(int M, int p) {
  int lo = M;
  for (int j = 0; j <= p; ++j) lo -= plan_w(M, j);
  return lo;
}
__host__ __device__ constexpr int threads_per_frame(int M) { return (1 << M) / 16; }
__host__ __device__ constexpr int frames_per_wg(int M) {
  return threads_per_frame(M) >= 256 ? 1 : 256 / threads_per_frame(M);
}
__host__ __device__ constexpr int wg_size(int M) { return threads_per_frame(M) * frames_per_wg(M); }
__host__ __device__ constexpr int pad_slots(int n) { return n + (n >> 4); }

__device__ __forceinline__ int pad(int i) { return i + (i >> 4); }

// The image carries one more pad slot per 512 samples (M >= 10): the last pass assigns chunks to
// threads in bit-reversed order (below), so a 32-lane group then reads samples 2^(M-5) apart, whose
// slots 2^(M-5) (1 + 1/16) j would share banks; one slot per 512 samples spreads exactly those
// (and leaves the other exchanges' access shapes conflict-free at 2048 / 4096 points: tools/lds_sim.py).
// ONE layout for every exchange of a frame: a thread rewrites exactly the slots of the samples it
// read (in place), which is what lets consecutive exchanges share the image with one barrier each.
__host__ __device__ constexpr int fft_image_slots(int M) { return pad_slots(1 << M) + (M >= 10 ? (1 << M) >> 9 : 0); }
template <int M>
__host__ __device__ constexpr int padx(int i) {
  return i + (i >> 4) + (M >= 10 ? (i >> 9) : 0);
}

// sample index of register e (0..15) of thread tau in the pass that owns field [LO, LO+W)
template <int M, int LO, int W>
__device__ __forceinline__ int elem_index(int tau, int e) {
  constexpr int T = threads_per_frame(M);
  const int g = e >> W, r = e & ((1 << W) - 1);
  const int c = g * T + tau;
  const int low = c & ((1 << LO) - 1), high = c >> LO;
  return (high << (LO + W)) | (r << LO) | low;
}

// LDS slot of that sample = slot_base(tau, g) + slot_delta(r): the padding is additive across the
// field bits (no carry can cross them), so the per-register part is a compile-time constant
// that folds into the ds_read/ds_write offset field.  LAST = the exchange in front of the last pass.
// In the last pass (LO = 0) thread tau takes chunk bitrev(g T + tau) instead of chunk g T + tau: its
// outputs are then bins (bitrev(p) << (M - W)) | (g T + tau) -- consecutive lanes hold consecutive bins,
// and what follows the FFT (magnitudes to LDS, spectra to HBM) is written lane-contiguously.
template <int M, int LO, int W, bool LAST = false>
__device__ __forceinline__ int slot_base(int tau, int g) {
  constexpr int T = threads_per_frame(M);
  int c = g * T + tau;
  if constexpr (LAST && LO == 0) c = (int)(__brev((unsigned)c) >> (32 - (M - W)));
  const int i0 = ((c >> LO) << (LO + W)) | (c & ((1 << LO) - 1));
  return padx<M>(i0);
}
template <int M, int LO, int W>
__host__ __device__ constexpr int slot_delta(int r) {
  return padx<M>(r << LO);
}

__host__ __device__ constexpr int bitrev_c(int x, int bits) {
  int r = 0;
  for (int i = 0; i < bits; ++i) r |= ((x >> i) & 1) << (bits - 1 - i);
  return r;
}

// ---------------------------------------------------------------- F32 pieces

// (ax wx - ay wy, ax wy + ay wx) as v_pk_mul_f32 + v_pk_fma_f32.  The second op reads
// {ay, ay} x {-wy, wx}: both broadcasts, the swap and the one-sided negation are operand modifiers
// (op_sel / op_sel_hi / neg_lo).  Written as inline asm because the compiler does not fold a
// half-negated swap into the modifiers: it materialises {-wy, wx} with v_xor + v_mov for every
// twiddle, 60 extra VALU instructions per thread and frame at 4096 points (tools/count_insts.sh).
__device__ __forceinline__ f32x2 cmul(f32x2 a, f32x2 w) {
  const f32x2 axx = {a.x, a.x};
  const f32x2 t = axx * w;
  f32x2 r;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"
      : "=v"(r)
      : "v"(a), "v"(w), "v"(t));
  return r;
}

// d * exp(-2 pi i k16 / 16), k16 (0..15) a compile-time constant after unrolling.  -i is ONE packed
// multiply (the re/im swap is an op_sel modifier, the constant pair {1, -1} sits in SGPRs); every
// other twiddle, the W8 family included, is multiply + fma = 2 packed ops.  (Spelling the W8 ones as
// (x + y) * H, (y - x) * H compiles to 4 instructions: two packed adds, a move and a multiply.)
__device__ __forceinline__ f32x2 mul_w16(f32x2 d, int k16) {
  constexpr float C1 = 0.92387953251128675613f, S1 = 0.38268343236508977173f, H = 0.70710678118654752440f;
  constexpr float C[16] = {1.0f, C1, H, S1, 0.0f, -S1, -H, -C1, -1.0f, -C1, -H, -S1, 0.0f, S1, H, C1};
  constexpr float S[16] = {0.0f, S1, H, C1, 1.0f, C1, H, S1, 0.0f, -S1, -H, -C1, -1.0f, -C1, -H, -S1};
  if (k16 == 0) return d;
  const f32x2 dyx = {d.y, d.x};
  if (k16 == 4) return dyx * f32x2{1.0f, -1.0f};
  const f32x2 ss = {S[k16], -S[k16]}, cc = {C[k16], C[k16]};
  return __builtin_elementwise_fma(dyx, ss, d * cc);
}

// 2^W-point DIF DFT on x[g*2^W .. ), output q at position bitrev_W(q)
template <int W>
__device__ __forceinline__ void dft_dif(f32x2 (&x)[16], int g) {
  if constexpr (W == 4) {
    // 16 points as two radix-4 stages (= the radix-2 stages fused in pairs, same output positions):
    // the +-i rotations inside a radix-4 butterfly are fma(swap(d1), {+-1, -+1}, d0) -- one packed op
    // each instead of a rotation plus an add -- and only 9 of the 16 outputs of the first stage carry
    // a twiddle.  81 packed ops per 16 points against 110 for four radix-2 stages.
    const f32x2 pm = {1.0f, -1.0f}, mp = {-1.0f, 1.0f};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const f32x2 x0 = x[g * 16 + r], x4 = x[g * 16 + r + 4], x8 = x[g * 16 + r + 8], x12 = x[g * 16 + r + 12];
      const f32x2 s0 = x0 + x8, d0 = x0 - x8, s1 = x4 + x12, d1 = x4 - x12;
      const f32x2 d1s = {d1.y, d1.x};
      x[g * 16 + r] = s0 + s1;
      x[g * 16 + r + 4] = mul_w16(s0 - s1, 2 * r);
      x[g * 16 + r + 8] = mul_w16(__builtin_elementwise_fma(d1s, pm, d0), r);       // (d0 - i d1) W^r
      x[g * 16 + r + 12] = mul_w16(__builtin_elementwise_fma(d1s, mp, d0), 3 * r);  // (d0 + i d1) W^3r
    }
#pragma unroll
    for (int b = 0; b < 16; b += 4) {
      const f32x2 y0 = x[g * 16 + b], y1 = x[g * 16 + b + 1], y2 = x[g * 16 + b + 2], y3 = x[g * 16 + b + 3];
      const f32x2 s0 = y0 + y2, d0 = y0 - y2, s1 = y1 + y3, d1 = y1 - y3;
      const f32x2 d1s = {d1.y, d1.x};
      x[g * 16 + b] = s0 + s1;
      x[g * 16 + b + 1] = s0 - s1;
      x[g * 16 + b + 2] = __builtin_elementwise_fma(d1s, pm, d0);
      x[g * 16 + b + 3] = __builtin_elementwise_fma(d1s, mp, d0);
    }
    return;
  }
#pragma unroll
  for (int st = 0; st < W; ++st) {
    const int bl = W - 1 - st;
#pragma unroll
    for (int r0 = 0; r0 < (1 << W); ++r0) {
      if (r0 & (1 << bl)) continue;
      const int r1 = r0 | (1 << bl);
      const int k16 = (r0 & ((1 << bl) - 1)) << (3 - bl);
      const f32x2 a = x[g * (1 << W) + r0], b = x[g * (1 << W) + r1];
      x[g * (1 << W) + r0] = a + b;
      x[g * (1 << W) + r1] = mul_w16(a - b, k16);
    }
  }
}

// Base twiddles of pass P: W_{2^(LO+W)}^(low 2^j), j < W, for each of the thread's 16 >> W groups.
// ROM layout (built by the host, rspchain_api.cpp get_rom): per pass with LO > 0 a table indexed by
// `low` (2^LO entries) of 4 consecutive f32x2 {W^low, W^2low, W^4low, W^8low} = 32 bytes, tables of
// the passes back to back.  A lane reads its 32 bytes with two 16-byte loads and consecutive lanes read
// consecutive entries: 16 lines per wave for pass 0 at 4096 points, where indexing one W_N^k table by
// k = low 2^j touched 60 (as many L2 requests as the frame itself; TCP_TCC_READ_REQ, tools/pmc_tail.sh).
// ALL passes' base twiddles are loaded at the top of the kernel, together with the frame: a load issued
// between two passes would queue behind the streaming traffic of the other workgroups and stall the
// pass for its whole (loaded) L2 latency (phase stamps, tools/stamp.sh).
__host__ __device__ constexpr int tw_table_entries(int M, int P) {  // f32x2 entries of pass P's table
  return plan_lo(M, P) > 0 ? (4 << plan_lo(M, P)) : 0;
}
__host__ __device__ constexpr int tw_table_offset(int M, int P) {
  int o = 0;
  for (int p = 0; p < P; ++p) o += tw_table_entries(M, p);
  return o;
}
__host__ __device__ constexpr int tw_table_total(int M) { return tw_table_offset(M, plan_np(M)); }

template <int M, int P>
struct TwBase {
  static constexpr int W = plan_w(M, P), G = 16 >> W;
  f32x2 w[G][4];
};

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Streaming accesses: a frame's samples are read once and its words written once per launch, so both carry the
// non-temporal hint (they need not stay in L2 / the Infinity Cache): 44.2-44.6 -> 42.2-43.6 us at cfg 2, 78-79 -> 75.1-75.4
// us at cfg 4 (same-box A/B, three rounds, against a side build without the hints: -DRSP_NO_NT).
template <typename T>
__device__ __forceinline__ T stream_load(const T* p) {
  if constexpr (kStreamHints) return __builtin_nontemporal_load(p);
  else return *p;
}
template <typename T>
__device__ __forceinline__ void stream_store(T v, T* p) {
  if constexpr (kStreamHints) __builtin_nontemporal_store(v, p);
  else *p = v;
}
// a workgroup's frame loads / word stores issue ahead of the resident waves' arithmetic: 41.2-41.3 -> 40.8-40.9 us at
// cfg 2 (same-box A/B, three rounds)
__device__ __forceinline__ void wave_prio(int p) {
  if constexpr (kWavePrio) {
    if (p) __builtin_amdgcn_s_setprio(3);
    else __builtin_amdgcn_s_setprio(0);
  }
}


template <int M, int P>
__device__ __forceinline__ void load_tw(int tau, const f32x2* __restrict__ tw, TwBase<M, P>& b) {
  constexpr int W = plan_w(M, P), LO = plan_lo(M, P), G = 16 >> W, T = threads_per_frame(M);
  if constexpr (LO > 0) {
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int low = (g * T + tau) & ((1 << LO) - 1);
      const f32x4* e = reinterpret_cast<const f32x4*>(tw + tw_table_offset(M, P) + 4 * low);
      const f32x4 lo = e[0];
      b.w[g][0] = {lo.x, lo.y};
      b.w[g][1] = {lo.z, lo.w};
      if constexpr (W >= 3) {
        const f32x4 hi = e[1];
        b.w[g][2] = {hi.x, hi.y};
        b.w[g][3] = {hi.z, hi.w};
      }
    }
  }
}

// One register pass of the F32 FFT: 2^W-point DFTs, then x[p] *= W_{2^(LO+W)}^(low q), q = bitrev(p),
// the 2^W - 1 twiddles built from the W base ones with products <= 3 deep.
template <int M, int P>
__device__ __forceinline__ void pass_f32(f32x2 (&x)[16], const TwBase<M, P>& b) {
  constexpr int W = plan_w(M, P), LO = plan_lo(M, P), G = 16 >> W;
#pragma unroll
  for (int g = 0; g < G; ++g) {
    dft_dif<W>(x, g);
    if constexpr (LO > 0) {
      f32x2 w[1 << W];
      w[1] = b.w[g][0];
      if constexpr (W >= 2) { w[2] = b.w[g][1]; w[3] = cmul(w[1], w[2]); }
      if constexpr (W >= 3) {
        w[4] = b.w[g][2];
        w[5] = cmul(w[1], w[4]); w[6] = cmul(w[2], w[4]); w[7] = cmul(w[3], w[4]);
      }
      if constexpr (W >= 4) {
        w[8] = b.w[g][3];
#pragma unroll
        for (int q = 1; q < 8; ++q) w[8 + q] = cmul(w[q], w[8]);
      }
#pragma unroll
      for (int p = 1; p < (1 << W); ++p) {
        const int q = bitrev_c(p, W);
        x[g * (1 << W) + p] = cmul(x[g * (1 << W) + p], w[q]);
      }
    }
  }
}

// the base twiddles of every pass of an M-stage frame (passes without twiddles hold nothing)
template <int M>
struct TwAll {
  TwBase<M, 0> p0;
  TwBase<M, 1> p1;
  TwBase<M, 2> p2;
  TwBase<M, 3> p3;
  __device__ __forceinline__ void load(int tau, const f32x2* __restrict__ tw) {
    load_tw<M, 0>(tau, tw, p0);
    load_tw<M, 1>(tau, tw, p1);
    if constexpr (plan_np(M) > 2) load_tw<M, 2>(tau, tw, p2);
    if constexpr (plan_np(M) > 3) load_tw<M, 3>(tau, tw, p3);
  }
  // an empty asm statement that reads and "rewrites" every loaded register: the loads have completed behind it
  __device__ __forceinline__ void touch() {
    auto t = [](auto& b) {
#pragma unroll
      for (auto& g : b.w) {
#pragma unroll
        for (auto& v : g) asm volatile("" : "+v"(v));
      }
    };
    t(p0); t(p1); t(p2); t(p3);
  }
  template <int P>
  __device__ __forceinline__ const TwBase<M, P>& get() const {
    if constexpr (P == 0) return p0;
    else if constexpr (P == 1) return p1;
    else if constexpr (P == 2) return p2;
    else return p3;
  }
};

// Whole F32 FFT of one frame through the LDS image `buf`, in two halves so that a caller may touch the
// samples between them (pre-FFT window):
//   fft_f32_load    all passes' base twiddles, then the thread's 16 samples: load(d) returns the sample at
//                   the thread's first sample index + d (d is a compile-time constant after unrolling, so
//                   the caller's address arithmetic folds into immediates);
//   fft_f32_passes  the register passes and their LDS exchanges.  On return x[g 2^WL + p] holds,
//                   unscaled, bin (bitrev(p) << (M - WL)) | (g T + tau), WL = width of the last pass.
// `hk` = the kernel's hook policy (side_build.hpp): the workgroup barrier of the exchanges, a call-out behind
// each exchange's barrier (the pipelined kernel issues its next frame's loads there), and the side builds'
// ablation switches and phase stamps -- all empty / __syncthreads() in the plain product kernels.
template <int M, typename Load>
__device__ __forceinline__ void fft_f32_load(Load load, int tau, const f32x2* __restrict__ tw, TwAll<M>& twb,
                                             f32x2 (&x)[16]) {
  twb.load(tau, tw);
  constexpr int W = plan_w(M, 0), LO = plan_lo(M, 0);
#pragma unroll
  for (int e = 0; e < 16; ++e) x[e] = load(elem_index<M, LO, W>(0, e) - elem_index<M, LO, W>(0, 0));
}

// The 16 reads of an exchange as ds_read_b64.  Left to itself the compiler pairs neighbouring 8-byte reads into
// ds_read2_b64, which the LDS serves at HALF the rate per byte (tools/valubench.hip, four waves per SIMD: ds_read_b64
// 6.2 ticks per 512 B, ds_read2_b64 24.1 per 1024 B; MI355X_MICROARCH.md: 256 B/clk on 64 banks vs 128 B/clk on 32), and
// a volatile access does not stop it.  Two asm blocks of eight reads with immediate offsets; the second ends with the
// wait for all sixteen and carries the first eight as in/out operands, so nothing can touch a register before its data
// has arrived (the compiler does not count LDS operations issued by inline asm).
typedef __attribute__((address_space(3))) unsigned char lds_byte_t;
template <int M, int LO, int W, bool LAST, int H>
__device__ __forceinline__ void lds_read8_b64(int tau, const f32x2* buf, f32x2 (&x)[16]) {
  static_assert(W == 3 || W == 4, "field widths of the plans");
  constexpr int G = W == 4 ? 0 : H, R0 = W == 4 ? 8 * H : 0, E = 8 * H;
  const uint32_t a = (uint32_t)(uintptr_t)(const lds_byte_t*)(buf + slot_base<M, LO, W, LAST>(tau, G));
#define RSP_OFF(j) "n"(8 * slot_delta<M, LO, W>(R0 + j))
  if constexpr (H == 0) {
    asm volatile("ds_read_b64 %0, %8 offset:%9\n\tds_read_b64 %1, %8 offset:%10\n\tds_read_b64 %2, %8 offset:%11\n\t"
                 "ds_read_b64 %3, %8 offset:%12\n\tds_read_b64 %4, %8 offset:%13\n\tds_read_b64 %5, %8 offset:%14\n\t"
                 "ds_read_b64 %6, %8 offset:%15\n\tds_read_b64 %7, %8 offset:%16"
                 : "=&v"(x[E]), "=&v"(x[E + 1]), "=&v"(x[E + 2]), "=&v"(x[E + 3]), "=&v"(x[E + 4]), "=&v"(x[E + 5]), "=&v"(x[E + 6]),
                   "=&v"(x[E + 7])
                 : "v"(a), RSP_OFF(0), RSP_OFF(1), RSP_OFF(2), RSP_OFF(3), RSP_OFF(4), RSP_OFF(5), RSP_OFF(6), RSP_OFF(7)
                 : "memory");
  } else {
    asm volatile("ds_read_b64 %0, %16 offset:%17\n\tds_read_b64 %1, %16 offset:%18\n\tds_read_b64 %2, %16 offset:%19\n\t"
                 "ds_read_b64 %3, %16 offset:%20\n\tds_read_b64 %4, %16 offset:%21\n\tds_read_b64 %5, %16 offset:%22\n\t"
                 "ds_read_b64 %6, %16 offset:%23\n\tds_read_b64 %7, %16 offset:%24\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(x[E]), "=&v"(x[E + 1]), "=&v"(x[E + 2]), "=&v"(x[E + 3]), "=&v"(x[E + 4]), "=&v"(x[E + 5]), "=&v"(x[E + 6]),
                   "=&v"(x[E + 7]), "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7])
                 : "v"(a), RSP_OFF(0), RSP_OFF(1), RSP_OFF(2), RSP_OFF(3), RSP_OFF(4), RSP_OFF(5), RSP_OFF(6), RSP_OFF(7)
                 : "memory");
  }
#undef RSP_OFF
}
// the largest immediate an exchange read needs (ds offsets are 16 bits)
template <int M, int LO, int W>
__host__ __device__ constexpr int lds_read_max_off() { return 8 * slot_delta<M, LO, W>((1 << W) - 1); }

// one exchange through the LDS image + the register pass behind it
template <int M, int P, typename Hooks>
__device__ __forceinline__ void fft_f32_exchange(int tau, f32x2* buf, const TwAll<M>& twb, f32x2 (&x)[16], Hooks& hk) {
  constexpr int NP = plan_np(M);
  constexpr int W0 = plan_w(M, P - 1), LO0 = plan_lo(M, P - 1);
  constexpr int W1 = plan_w(M, P), LO1 = plan_lo(M, P);
  constexpr bool LAST = P == NP - 1;
  hk.template before_exchange<P>();
  if (!hk.off(3)) {
#pragma unroll
    for (int g = 0; g < (16 >> W0); ++g) {
      f32x2* b0 = buf + slot_base<M, LO0, W0, LAST>(tau, g);
#pragma unroll
      for (int r = 0; r < (1 << W0); ++r) b0[slot_delta<M, LO0, W0>(r)] = x[g * (1 << W0) + r];
    }
  }
  Hooks::barrier();
  hk.template after_exchange_barrier<P>();
  if (!hk.off(3)) {
    if constexpr (kExchangeB64 && lds_read_max_off<M, LO1, W1>() < 65536) {
      lds_read8_b64<M, LO1, W1, LAST, 0>(tau, buf, x);
      lds_read8_b64<M, LO1, W1, LAST, 1>(tau, buf, x);
    } else {
#pragma unroll
      for (int g = 0; g < (16 >> W1); ++g) {
        const f32x2* b1 = buf + slot_base<M, LO1, W1, LAST>(tau, g);
#pragma unroll
        for (int r = 0; r < (1 << W1); ++r) x[g * (1 << W1) + r] = b1[slot_delta<M, LO1, W1>(r)];
      }
    }
  }
  hk.stamp(2 * P + 1);
  if (!hk.off(0)) pass_f32<M, P>(x, twb.template get<P>());
  hk.stamp(2 * P + 2);
}

template <int M, typename Hooks>
__device__ __forceinline__ void fft_f32_passes(int tau, f32x2* buf, const TwAll<M>& twb, f32x2 (&x)[16], Hooks& hk) {
  constexpr int NP = plan_np(M);
  hk.stamp(1);
  if (!hk.off(0)) pass_f32<M, 0>(x, twb.template get<0>());
  hk.stamp(2);
  fft_f32_exchange<M, 1>(tau, buf, twb, x, hk);
  if constexpr (NP > 2) fft_f32_exchange<M, 2>(tau, buf, twb, x, hk);
  if constexpr (NP > 3) fft_f32_exchange<M, 3>(tau, buf, twb, x, hk);
}

template <int M, typename Load>
__device__ __forceinline__ void fft_f32_frame(Load load, int tau, f32x2* buf,
                                              const f32x2* __restrict__ tw, f32x2 (&x)[16]) {
  TwAll<M> twb;
  NoHooks hk;
  fft_f32_load<M>(load, tau, tw, twb, x);
  fft_f32_passes<M>(tau, buf, twb, x, hk);
}

// sample offset (from the thread's first sample) of register e before pass 0
template <int M>
__host__ __device__ constexpr int sample_offset(int e) {
  return elem_index<M, plan_lo(M, 0), plan_w(M, 0)>(0, e) - elem_index<M, plan_lo(M, 0), plan_w(M, 0)>(0, 0);
}
// first sample index of thread tau in pass 0 (its other 15 are at compile-time offsets from it)
template <int M>
__device__ __forceinline__ int first_sample(int tau) {
  return elem_index<M, plan_lo(M, 0), plan_w(M, 0)>(tau, 0);
}

// bin held in register (g, p) after fft_f32_frame
template <int M>
__device__ __forceinline__ int bin_of(int tau, int g, int p) {
  constexpr int NP = plan_np(M), WL = plan_w(M, NP - 1), T = threads_per_frame(M);
  return (bitrev_c(p, WL) << (M - WL)) | (g * T + tau);
}

// ---------------------------------------------------------------- FIXED16 pieces

// arithmetic >> n with the FFT trim type (oracle: trim_shift).
// Generic form: bias 0 (floor) or 2^(n-1); convergent clears the LSB on an exact tie.
__device__ __forceinline__ int trim_n(int x, int n, int bias, int conv) {
  const int t = x + bias;
  int r = t >> n;
  const int tie = ((t & ((1 << n) - 1)) == 0) ? conv : 0;
  return r & ~tie;
}
// Convergent (the default) in 3 ops: round-half-even(x / 2^n) = (x + 2^(n-1) - 1 + ((x >> n) & 1)) >> n
template <int NBITS>
__device__ __forceinline__ int trim_conv(int x) {
  return (x + ((1 << (NBITS - 1)) - 1) + ((x >> NBITS) & 1)) >> NBITS;
}
__device__ __forceinline__ int wrap16(int x) { return (int)(short)x; }

// One register pass of the fixed-point FFT: radix-2 stages of field [LO, LO+W), each with its own Q2.14 twiddle,
// (a+b) trimmed by 1 bit, (a-b)*W by 15 bits -- the dataflow of the SDF pipeline.  The twiddle ROM is an LDS copy:
// for a given stage and butterfly the index is (per-thread low << s) + compile-time constant, so every lookup is
// one ds_read with an immediate offset.
// The samples stay PACKED, {re[31:16], im[15:0]} (the beat format), through passes and exchanges.
// A radix-2 butterfly of the SDF pipeline in ~21 VALU operations instead of 26 (and nothing around the exchanges):
//   sum    a + b is 17 bits wide, (a + b) / 2 is not: floor((a + b) / 2) = (a & b) + ((a ^ b) >> 1) per 16-bit lane
//          (v_and, v_xor, v_pk_ashrrev_i16, v_pk_add_u16), the rounding bit of the trim from (a ^ b) & 1
//   diff   (a - b) W = a W - b W never forms the 17-bit difference: Re = a . {wr, -wi} - b . {wr, -wi},
//          Im = a . {wi, wr} - b . {wi, wr} with v_dot2_i32_i16 (exact: |a . w| <= 2^30), the 15-bit trim on the
//          32-bit results as before, the two 16-bit results packed by one v_perm_b32 (= the wrap to 16 bits)
// and no unpacking / packing around the LDS exchanges.  The LDS ROM holds both operand forms of a twiddle,
// rom[k] = {W1 = {wr, -wi}, W2 = {wi, wr}} (8 bytes; fx_rom_entry).  Bit-identical to pass_fx (same integers).
typedef short s16x2 __attribute__((ext_vector_type(2)));

__host__ __device__ __forceinline__ uint2 fx_rom_entry(uint32_t w) {  // w = (wr << 16) | (wi & 0xffff), the global ROM's first format
  const uint32_t wr = w >> 16, wi = w & 0xffffu;
  return make_uint2((wr << 16) | ((0u - wi) & 0xffffu), (wi << 16) | wr);
}
// LDS slot of twiddle k: one pad entry per 16 and one more per 256.  A stage's lookups differ between lanes by
// low << s (low = the thread's low index bits): unpadded, the lanes of a wave then share 16 >> s bank pairs (s < 4) or
// ONE (s >= 4) -- 68 % of the FIXED16 kernel's LDS cycles were bank conflicts (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE).
// With the pads sixteen consecutive values of low land in sixteen different bank pairs for every s <= 7.  The slot is
// additive over disjoint bit fields (no carries), so slot(low << s) + slot(constant part) is the address and the
// constant part stays an instruction's immediate offset.
__host__ __device__ constexpr int fx_rom_slot(int k) { return k + (k >> 4) + (k >> 8); }
__host__ __device__ constexpr int fx_rom_slots(int M) { return fx_rom_slot((1 << M) / 2) + 1; }
__host__ __device__ constexpr int fx_rom_bytes(int M) { return 8 * fx_rom_slots(M); }
// global ROM -> LDS, by all the workgroup's threads.  The global ROM holds N/2 words {wr, wi} (what the small-frame
// kernel and the stage-option path read) followed by the same N/2 twiddles as fx_rom_entry pairs (rspchain_api.cpp
// get_rom), so the fill is a copy.
__device__ __forceinline__ void fx_rom_fill(uint2* rom, const uint32_t* __restrict__ twq, int half, int tid, int nthreads) {
  const uint2* __restrict__ src = reinterpret_cast<const uint2*>(twq + half);
  for (int i = tid; i < half; i += nthreads) rom[fx_rom_slot(i)] = src[i];
}
__device__ __forceinline__ int fx_rom_wr(uint2 e) { return (int)(short)(e.y & 0xffffu); }
__device__ __forceinline__ int fx_rom_wi(uint2 e) { return (int)(short)(e.y >> 16); }

// The butterfly's lower output for the twiddles W = 1 and W = -j (ROM entries 0 and N/4: {16384, 0} and {0, -16384},
// exact in Q2.14): (a - b) 2^14 trimmed by 15 bits is (a - b) / 2 trimmed by one, so the product is the SUM's
// arithmetic on a - b = a + ~b + 1, per 16-bit lane:
//   floor((a - b) / 2) = (a & ~b) + ((a ^ ~b) >> 1) + ((a ^ ~b) & 1),  dropped bit = (a ^ b) & 1
// 8 operations instead of 15 and no ROM read; 22 of the 32 butterflies of a 4-stage last pass (the last stage is all
// W = 1, the one before it half 1 half -j, ...).  (a - b)(-j) = (a.im - b.im) + j (b.re - a.re) is the same on the
// lane-swapped operands {a.im, b.re} - {b.im, a.re} (-x 2^14 trims like (b - a) 2^14: same integer).
// The two 32-bit sums of a butterfly's product, pr = a . w1 - b . w1 + c and pi = a . w2 - b . w2 + c, as four
// v_dot2_i32_i16 (VOP3P: the addend is an operand).  Through __builtin_amdgcn_sdot2 this compiler picks the
// accumulate-in-place form (v_dot2c) and pays a v_mov 0 per chain; spelled here that is two operations less per
// butterfly.  The compiler does not track dot-instruction hazards across inline asm, so the block keeps them itself:
// the second dot of a chain reads the first's result as its addend (same opcode, SrcC: no wait states), and the
// s_nop leaves the three wait states a dot result needs before any other VALU instruction may read it.
__device__ __forceinline__ void fx_dot_pair(uint32_t a, uint32_t b, uint32_t w1, uint32_t w2, uint32_t n1, uint32_t n2,
                                            int c, int& pr, int& pi) {
  asm("v_dot2_i32_i16 %0, %3, %6, %8\n\t"
      "v_dot2_i32_i16 %1, %3, %7, %8\n\t"
      "v_dot2_i32_i16 %0, %2, %4, %0\n\t"
      "v_dot2_i32_i16 %1, %2, %5, %1\n\t"
      "s_nop 1"
      : "=&v"(pr), "=&v"(pi)
      : "v"(a), "v"(b), "v"(w1), "v"(w2), "v"(n1), "v"(n2), "s"(c));
}

template <bool CONV>
__device__ __forceinline__ uint32_t fx_half_diff(uint32_t a, uint32_t b, uint32_t half_mask) {
  const s16x2 one2 = {1, 1};
  const uint32_t xn = ~(a ^ b);
  const s16x2 f = __builtin_bit_cast(s16x2, a & ~b) + (__builtin_bit_cast(s16x2, xn) >> one2) +
                  __builtin_bit_cast(s16x2, xn & 0x00010001u);
  const uint32_t fb = __builtin_bit_cast(uint32_t, f);
  const uint32_t rb = CONV ? (~xn & fb & 0x00010001u) : (~xn & half_mask);
  return __builtin_bit_cast(uint32_t, f + __builtin_bit_cast(s16x2, rb));
}

template <int M, int P, bool CONV>
__device__ __forceinline__ void pass_fx_pk(uint32_t (&z)[16], int tau, const uint2* tw, const ChainRegs& rg) {
  constexpr int W = plan_w(M, P), LO = plan_lo(M, P), G = 16 >> W, T = threads_per_frame(M);
  const s16x2 one2 = {1, 1};
  // floor / half-up (CONV = false): the sum's rounding bit is (a ^ b) & 1 for half-up, 0 for floor
  const uint32_t half_mask = rg.trim_bias1 ? 0x00010001u : 0u;
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const int low = (g * T + tau) & ((1 << LO) - 1);
#pragma unroll
    for (int st = 0; st < W; ++st) {
      const int bl = W - 1 - st;
      const int s = M - 1 - (LO + bl);  // radix-2 stage number of this bit
      const uint2* tws = tw + fx_rom_slot(low << s);
#pragma unroll
      for (int r0 = 0; r0 < (1 << W); ++r0) {
        if (r0 & (1 << bl)) continue;
        const int r1 = r0 | (1 << bl);
        const int jj = r0 & ((1 << bl) - 1);
        const int ia = g * (1 << W) + r0, ib = g * (1 << W) + r1;
        const uint32_t a = z[ia], b = z[ib];
        const uint32_t x = a ^ b;
        const s16x2 f = __builtin_bit_cast(s16x2, a & b) + (__builtin_bit_cast(s16x2, x) >> one2);  // floor((a + b) / 2)
        const uint32_t fb = __builtin_bit_cast(uint32_t, f);
        const uint32_t rb = CONV ? (x & fb & 0x00010001u) : (x & half_mask);  // convergent: a tie goes to the even neighbour
        z[ia] = __builtin_bit_cast(uint32_t, f + __builtin_bit_cast(s16x2, rb));
        if (LO == 0 && jj == 0) {  // W = 1 (compile-time after unrolling: low = 0 in the pass of the last stages)
          z[ib] = fx_half_diff<CONV>(a, b, half_mask);
          continue;
        }
        if (LO == 0 && (jj << s) == (1 << (M - 2))) {  // W = -j
          z[ib] = fx_half_diff<CONV>(__builtin_amdgcn_alignbit(a, b, 16), __builtin_amdgcn_alignbit(b, a, 16), half_mask);
          continue;
        }
        const uint2 w = tws[fx_rom_slot((jj << LO) << s)];
        // a . w + b . (-w) per component (negating the twiddle is exact, |w| <= 2^14; negating a sample would not be),
        // the trim's rounding constant as the first addend; the negations are shared by the butterflies of a twiddle
        const s16x2 zero2 = {0, 0};
        const uint32_t n1 = __builtin_bit_cast(uint32_t, zero2 - __builtin_bit_cast(s16x2, w.x));
        const uint32_t n2 = __builtin_bit_cast(uint32_t, zero2 - __builtin_bit_cast(s16x2, w.y));
        int pr, pi;
        fx_dot_pair(a, b, w.x, w.y, n1, n2, CONV ? 16383 : rg.trim_bias15, pr, pi);
        // the 16-bit results are bits [30:15] of the rounded sums = the high halves of twice them (mod 2^32: the
        // wrap to 16 bits drops bit 31 anyway), so no shift right: one v_add_lshl per component, one v_perm for both.
        // Convergent: (y + bit 15 of y) >> 15 with y = x + 2^14 - 1 (bit 15 of y and of x differ only where the
        // increment cannot carry into bit 15).
        uint32_t tr, ti;
        if constexpr (CONV) {
          tr = ((uint32_t)pr + __builtin_amdgcn_ubfe((uint32_t)pr, 15u, 1u)) << 1;
          ti = ((uint32_t)pi + __builtin_amdgcn_ubfe((uint32_t)pi, 15u, 1u)) << 1;
        } else {
          tr = (uint32_t)pr << 1;
          ti = (uint32_t)pi << 1;
        }
        z[ib] = __builtin_amdgcn_perm(tr, ti, 0x07060302u);  // {tr[31:16], ti[31:16]}
      }
    }
  }
}

// The same pass with the per-stage options of FFTParams.fixed that no reference configuration uses
// (expandLogic(s) = 1: the stage keeps its (w+1)-bit results; keepMSBorLSB(s) = false: it drops the MSB
// instead of the LSB) -- spec: docs/FIXED_POINT_SPEC.md section 3, oracle: orc_fft_fixed_ex.  Generic and
// slow on purpose (64-bit products, run-time shifts); values stay below 2^29 and live in 32-bit registers.
__device__ __forceinline__ long long trim_var(long long x, int n, const ChainRegs& rg) {
  if (n <= 0) return x;
  const long long half = rg.trim_bias1 ? (1ll << (n - 1)) : 0ll;  // trim_bias1 != 0 <=> not floor
  const long long t = x + half;
  long long r = t >> n;
  if (rg.trim_conv && (t & ((1ll << n) - 1ll)) == 0) r &= ~1ll;
  return r;
}
__device__ __forceinline__ int wrap_bits(long long x, int bits) {
  return (int)((x << (64 - bits)) >> (64 - bits));
}

template <int M, int P>
__device__ __forceinline__ void pass_fx_opt(int (&xr)[16], int (&xi)[16], int tau, const uint2* tw,
                                            const ChainRegs& rg) {
  constexpr int W = plan_w(M, P), LO = plan_lo(M, P), G = 16 >> W, T = threads_per_frame(M);
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const int low = (g * T + tau) & ((1 << LO) - 1);
#pragma unroll
    for (int st = 0; st < W; ++st) {
      const int bl = W - 1 - st;
      const int s = M - 1 - (LO + bl);  // radix-2 stage number of this bit
      const int grow = (int)((rg.expand_mask >> s) & 1u);
      const int lsb = !grow && ((rg.keep_lsb_mask >> s) & 1u);
      const int sh_sum = (grow || lsb) ? 0 : 1, sh_prod = 14 + sh_sum;
      const int wout = 16 + __popc(rg.expand_mask & ((2u << s) - 1u));
      const uint2* tws = tw + fx_rom_slot(low << s);
#pragma unroll
      for (int r0 = 0; r0 < (1 << W); ++r0) {
        if (r0 & (1 << bl)) continue;
        const int r1 = r0 | (1 << bl);
        const int jj = r0 & ((1 << bl) - 1);
        const uint2 w = tws[fx_rom_slot((jj << LO) << s)];
        const long long wr = fx_rom_wr(w), wi = fx_rom_wi(w);
        const int ia = g * (1 << W) + r0, ib = g * (1 << W) + r1;
        const long long sr = (long long)xr[ia] + xr[ib], si = (long long)xi[ia] + xi[ib];
        const long long dr = (long long)xr[ia] - xr[ib], di = (long long)xi[ia] - xi[ib];
        const long long pr = dr * wr - di * wi, pi = dr * wi + di * wr;
        xr[ia] = wrap_bits(trim_var(sr, sh_sum, rg), wout);
        xi[ia] = wrap_bits(trim_var(si, sh_sum, rg), wout);
        xr[ib] = wrap_bits(trim_var(pr, sh_prod, rg), wout);
        xi[ib] = wrap_bits(trim_var(pi, sh_prod, rg), wout);
      }
    }
  }
}


// Whole FIXED16 FFT of one frame through the LDS image at fbase (4-byte slots {re[31:16], im[15:0]}; 8-byte slots when
// a stage option grows the word), twiddle ROM already in LDS at rom.  xr / xi hold the thread's 16 samples (first
// sample first_sample(tau), offsets sample_offset(e)) on entry and its 16 bins on return, as fft_f32_frame.
// FX selects the path at compile time -- 0: convergent trim (3-op closed form), 1: floor / half-up, 2: stage options;
// -1: by the register snapshot at run time.  One kernel holding all three paths carries the registers of the widest
// (the stage-option path: 180 VGPRs against 77-140 for the others), i.e. half the occupancy for every configuration.
// The packed form: z[e] = {re[31:16], im[15:0]} (the beat format) in and out, trim path CONV (convergent) or the
// floor / half-up pair -- what front_end and the 2-D kernels call when no stage option is set.
template <int M, bool CONV, typename Hooks = NoHooks>
__device__ __forceinline__ void fft_fx_frame_pk(uint32_t (&z)[16], int tau, unsigned char* fbase, const uint2* rom,
                                                const ChainRegs& rg, Hooks hk = Hooks{}) {
  constexpr int NP = plan_np(M);
  uint32_t* buf = reinterpret_cast<uint32_t*>(fbase);
  if (!hk.off(0)) pass_fx_pk<M, 0, CONV>(z, tau, rom, rg);
  auto exchange = [&](auto pc) {
    constexpr int P = decltype(pc)::value;
    constexpr int W0 = plan_w(M, P - 1), LO0 = plan_lo(M, P - 1);
    constexpr int W1 = plan_w(M, P), LO1 = plan_lo(M, P);
    constexpr bool LAST = P == NP - 1;
    if (!hk.off(3)) {
#pragma unroll
    for (int g = 0; g < (16 >> W0); ++g) {
      uint32_t* b0 = buf + slot_base<M, LO0, W0, LAST>(tau, g);
#pragma unroll
      for (int r = 0; r < (1 << W0); ++r) b0[slot_delta<M, LO0, W0>(r)] = z[g * (1 << W0) + r];
    }
    __syncthreads();
#pragma unroll
    for (int g = 0; g < (16 >> W1); ++g) {
      const uint32_t* b1 = buf + slot_base<M, LO1, W1, LAST>(tau, g);
#pragma unroll
      for (int r = 0; r < (1 << W1); ++r) z[g * (1 << W1) + r] = b1[slot_delta<M, LO1, W1>(r)];
    }
    }
    if (!hk.off(0)) pass_fx_pk<M, P, CONV>(z, tau, rom, rg);
  };
  exchange(std::integral_constant<int, 1>{});
  if constexpr (NP > 2) exchange(std::integral_constant<int, 2>{});
  if constexpr (NP > 3) exchange(std::integral_constant<int, 3>{});
}

template <int M, int FX = -1, typename Hooks = NoHooks>
__device__ __forceinline__ void fft_fx_frame(int (&xr)[16], int (&xi)[16], int tau, unsigned char* fbase,
                                             const uint2* rom, const ChainRegs& rg, Hooks hk = Hooks{}) {
  constexpr int NP = plan_np(M);
  auto run = [&](auto conv_c) {
    constexpr bool CONV = decltype(conv_c)::value;
    uint32_t z[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) z[e] = ((uint32_t)xr[e] << 16) | ((uint32_t)xi[e] & 0xffffu);  // folds away behind a beat's unpacking
    fft_fx_frame_pk<M, CONV>(z, tau, fbase, rom, rg, hk);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      xr[e] = (int)(short)(z[e] >> 16);
      xi[e] = (int)(short)(z[e] & 0xffffu);
    }
  };
  // expandLogic / keepMSBorLSB = false somewhere: generic stages, 8-byte exchange slots (words grow past 16 bits)
  auto run_opt = [&]() {
    uint2* wbuf = reinterpret_cast<uint2*>(fbase);
    pass_fx_opt<M, 0>(xr, xi, tau, rom, rg);
    auto exchange = [&](auto pc) {
      constexpr int P = decltype(pc)::value;
      constexpr int W0 = plan_w(M, P - 1), LO0 = plan_lo(M, P - 1);
      constexpr int W1 = plan_w(M, P), LO1 = plan_lo(M, P);
      constexpr bool LAST = P == NP - 1;
#pragma unroll
      for (int g = 0; g < (16 >> W0); ++g) {
        uint2* b0 = wbuf + slot_base<M, LO0, W0, LAST>(tau, g);
#pragma unroll
        for (int r = 0; r < (1 << W0); ++r) {
          const int e = g * (1 << W0) + r;
          b0[slot_delta<M, LO0, W0>(r)] = make_uint2((uint32_t)xr[e], (uint32_t)xi[e]);
        }
      }
      __syncthreads();
#pragma unroll
      for (int g = 0; g < (16 >> W1); ++g) {
        const uint2* b1 = wbuf + slot_base<M, LO1, W1, LAST>(tau, g);
#pragma unroll
        for (int r = 0; r < (1 << W1); ++r) {
          const uint2 b = b1[slot_delta<M, LO1, W1>(r)];
          xr[g * (1 << W1) + r] = (int)b.x;
          xi[g * (1 << W1) + r] = (int)b.y;
        }
      }
      pass_fx_opt<M, P>(xr, xi, tau, rom, rg);
    };
    exchange(std::integral_constant<int, 1>{});
    if constexpr (NP > 2) exchange(std::integral_constant<int, 2>{});
    if constexpr (NP > 3) exchange(std::integral_constant<int, 3>{});
    // the 2 x 16-bit stream to the magnitude block carries the 16 MSBs of a grown word
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      xr[e] = (int)(short)(xr[e] >> rg.growth);
      xi[e] = (int)(short)(xi[e] >> rg.growth);
    }
  };
  // convergent (the default trim) has a 3-op closed form; floor / half-up share the generic one
  if constexpr (FX == 0) run(std::true_type{});
  else if constexpr (FX == 1) run(std::false_type{});
  else if constexpr (FX == 2) run_opt();
  else if (rg.keep_lsb_mask | rg.expand_mask) run_opt();
  else if (rg.trim_conv) run(std::true_type{});
  else run(std::false_type{});
}

// ---------------------------------------------------------------- FIXED16 magnitude and CFAR arithmetic
__device__ __forceinline__ int jpl_fx(int re, int im) {
  const int ar = re < 0 ? -re : re, ai = im < 0 ? -im : im;
  const int u = max(ar, ai), v = min(ar, ai);
  const int m = max(u + (v >> 3), ((7 * u) >> 3) + (v >> 1));
  return min(m, 32767);
}

// jpl_fx of TWO packed bins at once (z = {re[31:16], im[15:0]}): the halves are regrouped into {re0, re1} / {im0, im1}
// and everything runs on unsigned 16-bit lanes -- |-32768| = 0x8000 is a valid u16, the sums stay below 2^16, and
// (7 u) >> 3 = u - ((u + 7) >> 3) avoids the 19-bit product.  19 operations per pair against 2 x (14 + 2 to unpack).
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void jpl_fx_pair(uint32_t z0, uint32_t z1, int& m0, int& m1) {
  const s16x2 zero2 = {0, 0};
  const s16x2 re = __builtin_bit_cast(s16x2, __builtin_amdgcn_perm(z1, z0, 0x07060302u));  // {z0.hi, z1.hi} = {re0 (lo), re1 (hi)}
  const s16x2 im = __builtin_bit_cast(s16x2, __builtin_amdgcn_perm(z1, z0, 0x05040100u));  // {z0.lo, z1.lo}
  const u16x2 ar = __builtin_bit_cast(u16x2, __builtin_elementwise_max(re, zero2 - re));
  const u16x2 ai = __builtin_bit_cast(u16x2, __builtin_elementwise_max(im, zero2 - im));
  const u16x2 u = __builtin_elementwise_max(ar, ai), v = __builtin_elementwise_min(ar, ai);
  const u16x2 k1 = {1, 1}, k3 = {3, 3}, k7 = {7, 7}, kmax = {32767, 32767};
  const u16x2 t1 = u + (v >> k3);
  const u16x2 t2 = (u - ((u + k7) >> k3)) + (v >> k1);
  const uint32_t m = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_elementwise_max(t1, t2), kmax));
  m0 = (int)(m & 0xffffu);
  m1 = (int)(m >> 16);
}

__device__ __forceinline__ int mag_fx(int re, int im, const ChainRegs& rg,
                                      const int16_t* __restrict__ log_lut) {
  if (rg.mag_mode == 2) return jpl_fx(re, im);
  if (rg.mag_mode == 0) {
    const long long s = ((long long)re * re + (long long)im * im) >> rg.bp_data;
    return (int)(s > 32767 ? 32767 : s);
  }
  int x = jpl_fx(re, im);
  x = x < 1 ? 1 : x;
  const int e = 31 - __clz(x);
  const int lw = rg.lut_w;
  unsigned f = e >= lw ? ((unsigned)x >> (e - lw)) : ((unsigned)x << (lw - e));
  f &= (1u << lw) - 1u;
  return (e - rg.bp_data) * (1 << rg.bp_log) + (int)log_lut[f];
}

// ---------------------------------------------------------------- CFAR arithmetic

template <typename V>
struct CfarMath;

template <>
struct CfarMath<float> {
  static __device__ __forceinline__ float side(float sum, const ChainRegs& rg) { return sum * rg.div_f; }
  static __device__ __forceinline__ float half_sum(float a, float b) { return 0.5f * (a + b); }
  // thr = stat * A + B with (A, B) = (scaler, 0) linear or (1, scaler) log: both exact
  static __device__ __forceinline__ uint32_t finish(float stat, float cut, bool group_ok, int k,
                                                    int log2n, const ChainRegs& rg) {
    (void)k; (void)log2n;
    const float A = rg.linear ? rg.scaler_f : 1.0f, B = rg.linear ? 0.0f : rg.scaler_f;
    const float thr = __fmaf_rn(stat, A, B);
    const uint32_t peak = (cut > thr) && group_ok;
    return (__float_as_uint(thr) & ~1u) | peak;
  }
};

template <>
struct CfarMath<int> {
  static __device__ __forceinline__ int side(int sum, const ChainRegs& rg) { return sum >> rg.div_sum; }
  static __device__ __forceinline__ int half_sum(int a, int b) { return (a + b) >> 1; }
  // The domain is the (uniform) logOrLinearMode register: a scalar branch, one domain's arithmetic per cell.
  // Ranges: magnitudes are 16-bit, so cut / thr products below fit 32 bits; stat * scaler needs 64.
  static __device__ __forceinline__ uint32_t finish(int stat, int cut, bool group_ok, int k,
                                                    int log2n, const ChainRegs& rg) {
    int thr;
    if (!rg.linear) {
      const int lg = ((stat << rg.log_shl) >> rg.log_shr) + rg.log_scaler;
      thr = min(max(lg, rg.tmin), rg.tmax);
    } else if (rg.fast32) {  // (uniform) the product is exact in 32 bits: the same integers as the 64-bit form below
      const int l32 = (__mul24(stat, (int)rg.scaler_raw) << rg.lin_shl) >> rg.lin_shr;  // arithmetic: floor, as the spec's trim_shift
      thr = min(max(l32, rg.tmin), rg.tmax);
    } else {
      const long long prod = ((long long)stat * (long long)rg.scaler_raw) << rg.lin_shl;
      const long long lin64 = prod >> rg.lin_shr;
      thr = lin64 > (long long)rg.tmax ? rg.tmax : (lin64 < (long long)rg.tmin ? rg.tmin : (int)lin64);
    }
    // cut * 2^bp_thr > thr * 2^bp_in, both sides within 31 bits (16-bit values, shifts <= 15)
    const uint32_t peak = ((cut * (1 << rg.bp_thr)) > (thr * (1 << rg.bp_in))) && group_ok;
    return ((uint32_t)thr << (log2n + 1)) | ((uint32_t)k << 1) | peak;
  }
};


}  // namespace rsp
