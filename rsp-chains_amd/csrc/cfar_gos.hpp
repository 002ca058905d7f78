// GOS / ordered-statistic CFAR tail of the fused 1-D chain (cfg 4).
// GOSCFARType / GOSCACFARType with cfarAlgorithm = GOS (FftMagCfarChainTester.scala:105-127):
// the per-side statistic is the indexLagg-th / indexLead-th smallest cell of the window.
// The hardware keeps each window sorted with a linear insertion sorter; here every thread keeps
// ONE sorted window in registers: it bitonic-sorts the R cells starting at its first window
// start, then slides it (branch-free delete + insert, cmp/cndmask + med3 per element) over its
// run of consecutive starts, writing the two order statistics of every start to LDS.  The
// lagging window of cell k starts at k - G - R, the leading one at k + G + 1, so a cell needs two
// lookups.  Starts run over [-(G+R), N + G]; cells outside the frame come from the magnitude
// halo (zeros or the wrapped image).

#pragma once
#include "cfar_quad.hpp"

namespace rsp {

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
// every index is a compile-time constant after unrolling.  P0 > 1 runs only the merge phases p = P0, 2 P0, ...: it
// MERGES s[0, P0) and s[P0, R), both sorted (P0 a power of two; the network for R elements is the one for the next
// power of two without the comparators that touch an index >= R, whose inputs would be +infinity and stay put).
template <typename V, int R, int P0 = 1, typename S>
__device__ __forceinline__ void sort_net(S& s) {
#pragma unroll
  for (int p = P0; p < R; p <<= 1) {
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
template <typename V, int R>
__device__ __forceinline__ void sort_window(typename WinVec<V, R>::type& s) { sort_net<V, R>(s); }

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
template <typename V, int R, typename S>
__device__ __forceinline__ void slide_s(S& s, V old, V nw) {
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
template <typename V, int R>
__device__ __forceinline__ void slide(typename WinVec<V, R>::type& s, V old, V nw) { slide_s<V, R>(s, old, nw); }

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
  for (int i = 0; i < R; ++i) s[i] = mag[a0 + i];
  sort_window<V, R>(s);
  const bool two = idx_lagg != idx_lead;
  for (int st = 0; st < run; ++st) {
    const int oi = run * tau + st;
    o1[oi] = s[idx_lagg];
    if (two) o2[oi] = s[idx_lead];
    if (st + 1 < run) {
      const V old = mag[a0 + st], nw = mag[a0 + st + R];
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

// k-th smallest (0-based, K a compile-time constant) of the union of two ASCENDING sequences b[NB] and d[ND]: the
// smallest v = max(b[i], d[j]) over the splits i + j = K - 1 (b[-1] = d[-1] = -infinity, i.e. i = -1 contributes d[K]
// and j = -1 b[K]): b[0..i] and d[0..j] are K + 1 values <= v, and at the split that takes exactly the K + 1 smallest
// values v IS the statistic.  At most min(NB, ND) + 1 terms: for two 16-cell parts and K = 24 eight max + seven min
// = 15 operations, against the 31 of the half-cleaner cascade above.
template <int K, int NB, int ND, typename V, typename VB, typename VD>
__device__ __forceinline__ V select_split_k(const VB& b, const VD& d) {
  constexpr int ILO = K - ND > -1 ? K - ND : -1, IHI = NB - 1 < K ? NB - 1 : K, NT = IHI - ILO + 1;
  static_assert(K >= 0 && K < NB + ND && NT >= 1, "0 <= K < NB + ND");
  V t[NT];
#pragma unroll
  for (int i = ILO; i <= IHI; ++i) {
    const int j = K - 1 - i;
    // (indices clamped in the branches not taken: they are compile-time dead but still type-checked)
    if (i < 0) t[i - ILO] = d[K < ND ? K : 0];
    else if (j < 0) t[i - ILO] = b[K < NB ? K : 0];
    else t[i - ILO] = vmax1<V>(b[i], d[j < 0 ? 0 : j]);
  }
#pragma unroll
  for (int w = 1; w < NT; w <<= 1) {  // pairwise: a tree log2(NT) deep instead of a chain of NT dependent minima
#pragma unroll
    for (int i = 0; i + w < NT; i += 2 * w) t[i] = vmin1<V>(t[i], t[i + w]);
  }
  return t[0];
}

// The same statistics with the window SPLIT: the RUN windows of a thread (starts a0 .. a0 + RUN - 1) all contain
// the cells B = [a0 + RUN - 1, a0 + R); only the other RUN - 1 cells D change from start to start (one leaves at
// the front, one enters past B).  B is sorted once, D is kept sorted with the delete + insert slide -- 3 (RUN - 1)
// operations per start instead of 3 R -- and the order statistic is selected from the bitonic sequence
// [B ascending | D descending] with R - 1 min / max.  R = 32, RUN = 17: two 16-element sorts + 16 x 48 + 17 x 31 =
// 1547 operations per 17 starts against 382 + 16 x 94 = 1886 with one 32-cell sorted window.
// KC >= 0: indexLagg = indexLead = KC known at compile time (the launcher instantiates the usual choices, R / 2 and
// 3 R / 4): the selection is select_split_k on the two sorted parts, 15 operations at R = 32, K = 24 -- 1275 per 17
// starts.  (As a switch over a run-time index INSIDE the loop the same selection lost its gain to the compare chain.)
template <typename V, int R, int RUN, int KC = -1>
__device__ __forceinline__ void gos_stage_split(const V* mag, V* o1, V* o2, int tau, int G, int idx_lagg, int idx_lead) {
  constexpr int ND = RUN - 1, NB = R - ND;
  static_assert(NB >= 1 && (R & (R - 1)) == 0, "a common part and a power-of-two window");
  const int a0 = -(G + R) + RUN * tau;  // first window start of this thread
  typename WinVec<V, ND>::type d;
  typename WinVec<V, NB>::type b;
#pragma unroll
  for (int i = 0; i < ND; ++i) d[i] = mag[a0 + i];
#pragma unroll
  for (int i = 0; i < NB; ++i) b[i] = mag[a0 + ND + i];
  sort_window<V, ND>(d);
  sort_window<V, NB>(b);
  const bool two = idx_lagg != idx_lead;
  if constexpr (KC >= 0) {
    // the cell that leaves and the cell that enters are read ONE step ahead: the slide is a chain of asm statements
    // the scheduler cannot hoist a load over, so a read issued inside the step would expose its LDS latency 16 times
    V old = mag[a0], nw = mag[a0 + R];
#pragma unroll 1
    for (int st = 0; st < RUN; ++st) {
      const V old_n = mag[a0 + st + 1], nw_n = mag[a0 + st + 1 + R];  // inside the 256-cell halo
      o1[RUN * tau + st] = select_split_k<KC, NB, ND, V>(b, d);
      if (st + 1 < RUN) slide<V, ND>(d, old, nw);
      old = old_n;
      nw = nw_n;
    }
    return;
  }
#pragma unroll 1
  for (int st = 0; st < RUN; ++st) {
    V seq[R];  // [B ascending | D descending]
#pragma unroll
    for (int i = 0; i < NB; ++i) seq[i] = b[i];
#pragma unroll
    for (int i = 0; i < ND; ++i) seq[NB + i] = d[ND - 1 - i];
    int k1 = idx_lagg, k2 = idx_lead;
    asm volatile("" : "+s"(k1), "+s"(k2));  // keep the five side branches inside the loop (no 32-way unswitching)
    const int oi = RUN * tau + st;
    o1[oi] = select_bitonic<R, V>(seq, k1);
    if (two) o2[oi] = select_bitonic<R, V>(seq, k2);
    if (st + 1 < RUN) {
      const V old = mag[a0 + st], nw = mag[a0 + st + R];
      slide<V, ND>(d, old, nw);
    }
  }
}

// R = 32, compile-time index: the thread's 17 window starts as TWO runs that share the sorted middle of their windows.
// With m[j] = cell a0 + j: starts 0..7 all contain [7, 32), starts 8..16 all contain [16, 40); both contain C = [16, 32).
//   sort C (16 cells) once; run A: common part = merge(C, sort [7, 16)) -- 25 cells --, sliding part 7 cells;
//                           run B: common part = merge(C, sort [32, 40)) -- 24 cells --, sliding part 8 cells.
// A slide costs 3 cells-in-the-sliding-part operations, so halving that part halves the dominant term: 15 slides of 19 /
// 22 operations instead of 16 of 46; the sorts and merges cost 458 instead of 252.  ~1020 operations per 17 starts
// against 1243 for one run of 17 (gos_stage_split).  Selection as there: select_split_k on the two sorted parts.
template <typename V, int KC>
__device__ __forceinline__ void gos_stage_split2(const V* mag, V* o1, int tau, int G) {
  constexpr int R = 32, RUN = 17;
  const V* m = mag + (-(G + R) + RUN * tau);  // m[j] = cell a0 + j, a0 the thread's first window start
  V* o = o1 + RUN * tau;
  V c[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) c[i] = m[16 + i];
  sort_net<V, 16>(c);
  {  // starts 0..7
    V b[25], d[7];
#pragma unroll
    for (int i = 0; i < 16; ++i) b[i] = c[i];
    {
      V e[9];
#pragma unroll
      for (int i = 0; i < 9; ++i) e[i] = m[7 + i];
      sort_net<V, 9>(e);
#pragma unroll
      for (int i = 0; i < 9; ++i) b[16 + i] = e[i];
    }
    sort_net<V, 25, 16>(b);
#pragma unroll
    for (int i = 0; i < 7; ++i) d[i] = m[i];
    sort_net<V, 7>(d);
    V old = m[0], nw = m[R];
#pragma unroll 1
    for (int st = 0; st < 8; ++st) {
      const V old_n = m[st + 1], nw_n = m[st + 1 + R];  // one step ahead (see gos_stage_split)
      o[st] = select_split_k<KC, 25, 7, V>(b, d);
      if (st + 1 < 8) slide_s<V, 7>(d, old, nw);
      old = old_n;
      nw = nw_n;
    }
  }
  {  // starts 8..16
    V b[24], d[8];
#pragma unroll
    for (int i = 0; i < 16; ++i) b[i] = c[i];
    {
      V f[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) f[i] = m[R + i];
      sort_net<V, 8>(f);
#pragma unroll
      for (int i = 0; i < 8; ++i) b[16 + i] = f[i];
    }
    sort_net<V, 24, 16>(b);
#pragma unroll
    for (int i = 0; i < 8; ++i) d[i] = m[8 + i];
    sort_net<V, 8>(d);
    V old = m[8], nw = m[8 + R];
#pragma unroll 1
    for (int st = 8; st < RUN; ++st) {
      const V old_n = m[st + 1], nw_n = m[st + 1 + R];  // inside the 256-cell halo
      o[st] = select_split_k<KC, 24, 8, V>(b, d);
      if (st + 1 < RUN) slide_s<V, 8>(d, old, nw);
      old = old_n;
      nw = nw_n;
    }
  }
}

// window starts per thread of the 64-cell window on the split path: ceil((N + 2 G + 65) / (N / 16)) for the guard
// sizes up to 31 / 31 / 15 / 7 cells at 1024 / 2048+ / 512 / 256 points (other guards: the one-window path, KIND 2)
__host__ __device__ constexpr int gos_big_run(int M) { return M >= 11 ? 17 : M == 10 ? 18 : M == 9 ? 19 : 21; }

// BIG = the 64-cell window: its sorted window alone is 64 + 63 registers, so it is a kernel of its own -- as one path
// of a common kernel it set the register count (141 + scratch) and with it the occupancy (one 512-thread workgroup
// per CU at 8192 points) of every other window size.
// KIND: 0 = windows up to 32 cells; 1 = the 64-cell window on the split path (gos_big_run(M) starts per thread: the run
// of the usual guard sizes at each frame size); 2 = the 64-cell window, one sorted 64-cell vector slid over any number
// of starts (guards that change the run at up to 2048 points: its run-time pick of a 64-register vector goes through
// scratch memory, 640 B -- kept out of the other two).
template <int M, bool FIXED, int KIND, int FX>
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
  SideHooks hk;
  hk.init(rg);
// (side builds, -DRSP_ABLATE: mask bit 5 reads 64 resident frames instead of the batch, bit 6 drops the word stores)
  front_end<M, FIXED, V, FX>(in, hk.off(5) ? (frame & 63u) : frame, live, tau, fbase, rg, tw, log_lut,
                             reinterpret_cast<uint2*>(smem + (FIXED ? FixedRom<M, FX>::off(lay.frame_bytes) : 0)), mg, hk);

  // LDS images of the tail, all unpadded (a thread's run of window starts puts its lanes RUN = 17 words apart: odd,
  // conflict-free; the cells are handled as quads of four, like the CA tail):
  //   mag[x], x in [-256, N + 256]   magnitudes with zero / wrapped halos
  //   o1[s'], o2[s']                 order statistics of the window that starts at cell s' - (G + R)
  using V4 = typename Vec4<V>::type;
  V* mag = reinterpret_cast<V*>(fbase) + kHalo;
  V* o1 = reinterpret_cast<V*>(fbase + lay.o1_off);
  V* o2 = reinterpret_cast<V*>(fbase + lay.o2_off);
  uint32_t* det_cnt = reinterpret_cast<uint32_t*>(fbase + lay.det_off);
  uint2* det_stage = reinterpret_cast<uint2*>(fbase + lay.det_off + 8);
  const bool wrap = rg.edge != 0;
  __syncthreads();  // every thread is done reading the FFT image this overlays
  {  // magnitudes to LDS in natural bin order (as the quad tail): register (g, p) holds bin (bitrev(p) << (M - WL)) | (g T + tau)
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
  if (tau == 0) *det_cnt = 0u;
  __syncthreads();
  for (int h = tau; h < 32; h += T) {  // halos: 32 runs of 16 cells, zeros or the wrapped image
    const int x0 = h < 16 ? -kHalo + 16 * h : N + 16 * (h - 16);
    const int src = h < 16 ? x0 + N : x0 - N;
    const V4 z4 = {V(0), V(0), V(0), V(0)};
#pragma unroll
    for (int e = 0; e < 16; e += 4)
      *reinterpret_cast<V4*>(mag + x0 + e) = wrap ? *reinterpret_cast<const V4*>(mag + src + e) : z4;
    if (h == 31) mag[N + kHalo] = wrap ? mag[kHalo] : V(0);
  }
  __syncthreads();
  if (hk.off(1)) {  // (side builds) no order-statistic stage: what the rest of the kernel costs
  } else if constexpr (KIND == 1) {
    gos_stage_split<V, 64, gos_big_run(M)>(mag, o1, o2, tau, rg.G, rg.idx_lagg, rg.idx_lead);
  } else if constexpr (KIND == 2) {
    gos_stage<V, 64>(mag, o1, o2, tau, lay.run, rg.G, rg.idx_lagg, rg.idx_lead);
  } else {
    switch (rg.R) {
      case 4: gos_stage<V, 4>(mag, o1, o2, tau, lay.run, rg.G, rg.idx_lagg, rg.idx_lead); break;
      case 8: gos_stage<V, 8>(mag, o1, o2, tau, lay.run, rg.G, rg.idx_lagg, rg.idx_lead); break;
      case 16: gos_stage<V, 16>(mag, o1, o2, tau, lay.run, rg.G, rg.idx_lagg, rg.idx_lead); break;
      default:
        if (lay.run == 17 && RSP_GOS_SPLIT) {
          if (rg.idx_lagg == rg.idx_lead && rg.idx_lagg == 24) gos_stage_split2<V, 24>(mag, o1, tau, rg.G);
          else if (rg.idx_lagg == rg.idx_lead && rg.idx_lagg == 16) gos_stage_split2<V, 16>(mag, o1, tau, rg.G);
          else gos_stage_split<V, 32, 17>(mag, o1, o2, tau, rg.G, rg.idx_lagg, rg.idx_lead);
          break;
        }
        // 256- / 512-point frames: 16 / 32 threads per frame walk 18 - 20 window starts each; the same split with a
        // longer sliding part (a sorted 32-cell window slid 18 times costs 94 operations per start)
        if constexpr (M <= 9) {
          auto split_run = [&](auto run_c) {
            constexpr int RUN = decltype(run_c)::value;
            if (rg.idx_lagg == rg.idx_lead && rg.idx_lagg == 24) gos_stage_split<V, 32, RUN, 24>(mag, o1, o2, tau, rg.G, 24, 24);
            else if (rg.idx_lagg == rg.idx_lead && rg.idx_lagg == 16) gos_stage_split<V, 32, RUN, 16>(mag, o1, o2, tau, rg.G, 16, 16);
            else gos_stage_split<V, 32, RUN>(mag, o1, o2, tau, rg.G, rg.idx_lagg, rg.idx_lead);
          };
          constexpr int RUN0 = M == 9 ? 18 : 19;   // 2 G + 33 window starts beyond the frame's cells, G < 16 / G < 8
          if (RSP_GOS_SPLIT && lay.run == RUN0) { split_run(std::integral_constant<int, RUN0>{}); break; }
          if (RSP_GOS_SPLIT && lay.run == RUN0 + 1) { split_run(std::integral_constant<int, RUN0 + 1>{}); break; }
        }
        gos_stage<V, 32>(mag, o1, o2, tau, lay.run, rg.G, rg.idx_lagg, rg.idx_lead);
        break;
    }
  }
  __syncthreads();

  // Cells as quads: thread tau owns cells 4 (tau + T e) + i.  Lagging statistic of cell k = o1[k] (window start k - G - R):
  // one aligned 16-byte read per quad; leading = o2[k + D], D = 2 G + R + 1 -- odd (R is a power of two >= 4), so the
  // four values straddle two aligned quads: two reads and a (wave-uniform) choice of components.
  uint32_t word[16];
  V4 cutq[4];
  {
    const int D = 2 * rg.G + rg.R + 1;
    const int d_lo = D & ~3;          // aligned part of the offset
    const bool d3 = (D & 3) == 3;     // D mod 4 is 1 or 3
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k0 = 4 * (tau + T * e);
      const V4 a4 = *reinterpret_cast<const V4*>(o1 + k0);
      const V4 l0 = *reinterpret_cast<const V4*>(o2 + k0 + d_lo), l1 = *reinterpret_cast<const V4*>(o2 + k0 + d_lo + 4);
      const V4 b4 = d3 ? V4{l0[3], l1[0], l1[1], l1[2]} : V4{l0[1], l0[2], l0[3], l1[0]};
      const V4 cut = *reinterpret_cast<const V4*>(mag + k0);
      cutq[e] = cut;
      V nl = V(0), nr = V(0);
      if (rg.peak_grouping) {
        nl = mag[k0 - 1];
        nr = mag[k0 + 4];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const V a = a4[i], b = b4[i];
        V stat;
        if (rg.cfar_mode == 0) stat = CfarMath<V>::half_sum(a, b);
        else if (rg.cfar_mode == 1) stat = a > b ? a : b;
        else stat = a < b ? a : b;
        bool group_ok = true;
        if (rg.peak_grouping) group_ok = cut[i] > (i == 0 ? nl : cut[i - 1]) && cut[i] > (i == 3 ? nr : cut[i + 1]);
        word[4 * e + i] = CfarMath<V>::finish(stat, cut[i], group_ok, k0 + i, M, rg);
      }
    }
  }
  {
    QuadWords ww;
    QuadCuts<V4> cc;
#pragma unroll
    for (int j = 0; j < 16; ++j) ww.w[j] = word[j];
#pragma unroll
    for (int e = 0; e < 4; ++e) cc.q[e] = cutq[e];
    quad_emit<M, V, V4, SideHooks>(ww, cc, tau, frame, live, rg, hk.off(6) ? nullptr : out, fcount, fdet, det_cnt, det_stage);
  }
}

// GOS kernel LDS: magnitude with 256-cell halos + one or two order-statistic arrays + staging
template <int M>
static GosLayout gos_layout(const ChainRegs& rg) {
  constexpr int N = 1 << M, T = threads_per_frame(M);
  GosLayout l;
  l.run = (N + 2 * rg.G + rg.R + 1 + T - 1) / T;
  const int mag_bytes = 4 * (N + 2 * kHalo + 4);           // cells -256 .. N + 256, 16-byte multiple
  const int o_bytes = 4 * ((l.run * T + 8 + 3) & ~3);       // + 8: the second aligned quad of a leading read
  l.o1_off = mag_bytes;
  l.o2_off = rg.idx_lagg != rg.idx_lead ? l.o1_off + o_bytes : l.o1_off;
  l.det_off = (l.o2_off + o_bytes + 7) & ~7;
  const int total = l.det_off + 8 + 8 * kFrameDetCap;
  l.frame_bytes = ((total > FrameLds<M>::FFT_BYTES ? total : FrameLds<M>::FFT_BYTES) + 15) & ~15;
  return l;
}

}  // namespace rsp
