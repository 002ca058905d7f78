// 16..128-point frames: the run-time FFT size register (FftMagCfarChainTester.scala:82) may select
// any 2^k <= numPoints, far below what the LDS-tiled kernels of chain1d.hip are built for.  Such
// frames are tiny, so this kernel favours simplicity: ONE THREAD PER FRAME, the frame in LDS,
// lane-interleaved (word i of lane t at i * 64 + t: every lane touches the same i, so accesses are
// conflict-free and twiddle / register reads are wave-uniform).  Same arithmetic as the big
// kernels: radix-2 DIF stage by stage (FIXED16: Q2.14 twiddles, 1-bit / 15-bit trims), magnitude,
// direct-sum CFAR windows (all of CA / GO / SO / CASH / GOS, grouping, log, zero or wrap edges).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <float.h>

#include <type_traits>

#include "chain_regs.hpp"
#include "fft_lds.hpp"
#include "kernels.hpp"

namespace rsp {

namespace {

__device__ __forceinline__ int trim_s(int x, int n, int bias, int conv) {
  const int t = x + bias;
  const int r = t >> n;
  const int tie = ((t & ((1 << n) - 1)) == 0) ? conv : 0;
  return r & ~tie;
}

template <typename V>
struct SmallMath;

template <>
struct SmallMath<float> {
  static __device__ float mag(float re, float im, const ChainRegs& rg, const int16_t*) {
    const float ar = fabsf(re), ai = fabsf(im);
    const float u = fmaxf(ar, ai), v = fminf(ar, ai);
    const float jpl = fmaxf(u + v * 0.125f, u * 0.875f + v * 0.5f);
    if (rg.mag_mode == 2) return jpl;
    if (rg.mag_mode == 0) return re * re + im * im;
    return __log2f(fmaxf(jpl, FLT_MIN));
  }
  static __device__ float side(float s, const ChainRegs& rg) { return s * rg.div_f; }
  static __device__ float half_sum(float a, float b) { return 0.5f * (a + b); }
  static __device__ uint32_t finish(float stat, float cut, bool ok, int, int, const ChainRegs& rg) {
    const float thr = rg.linear ? stat * rg.scaler_f : stat + rg.scaler_f;
    return (__float_as_uint(thr) & ~1u) | (uint32_t)((cut > thr) && ok);
  }
};

template <>
struct SmallMath<int> {
  static __device__ int mag(int re, int im, const ChainRegs& rg, const int16_t* lut) {
    const int ar = re < 0 ? -re : re, ai = im < 0 ? -im : im;
    const int u = max(ar, ai), v = min(ar, ai);
    const int jpl = min(max(u + (v >> 3), ((7 * u) >> 3) + (v >> 1)), 32767);
    if (rg.mag_mode == 2) return jpl;
    if (rg.mag_mode == 0) {
      const long long s = ((long long)re * re + (long long)im * im) >> rg.bp_data;
      return (int)(s > 32767 ? 32767 : s);
    }
    const int x = jpl < 1 ? 1 : jpl;
    const int e = 31 - __clz(x);
    unsigned f = e >= rg.lut_w ? ((unsigned)x >> (e - rg.lut_w)) : ((unsigned)x << (rg.lut_w - e));
    f &= (1u << rg.lut_w) - 1u;
    return (e - rg.bp_data) * (1 << rg.bp_log) + (int)lut[f];
  }
  static __device__ int side(int s, const ChainRegs& rg) { return s >> rg.div_sum; }
  static __device__ int half_sum(int a, int b) { return (a + b) >> 1; }
  static __device__ uint32_t finish(int stat, int cut, bool ok, int k, int log2n, const ChainRegs& rg) {
    const long long lin64 = (((long long)stat * (long long)rg.scaler_raw) << rg.lin_shl) >> rg.lin_shr;
    const int lin = lin64 > (long long)rg.tmax ? rg.tmax : (lin64 < (long long)rg.tmin ? rg.tmin : (int)lin64);
    int lg = ((stat << rg.log_shl) >> rg.log_shr) + rg.log_scaler;
    lg = min(max(lg, rg.tmin), rg.tmax);
    const int thr = rg.linear ? lin : lg;
    const uint32_t peak = ((cut * (1 << rg.bp_thr)) > (thr * (1 << rg.bp_in))) && ok;
    return ((uint32_t)thr << (log2n + 1)) | ((uint32_t)k << 1) | peak;
  }
};

}  // namespace

template <bool FIXED>
__global__ void __launch_bounds__(64)
chain1d_small_kernel(const void* __restrict__ in, uint32_t* __restrict__ out, uint32_t n_frames, int log2n,
                     ChainRegs rg, const void* __restrict__ tw, const int16_t* __restrict__ log_lut,
                     uint32_t* __restrict__ fcount, uint2* __restrict__ fdet) {
  using V = typename std::conditional<FIXED, int, float>::type;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int n = 1 << log2n, t = threadIdx.x;
  const uint32_t frame = blockIdx.x * 64 + t;
  const bool live = frame < n_frames;
  V* xr = reinterpret_cast<V*>(smem);  // [n][64]
  V* xi = xr + n * 64;                 // [n][64]
  V* mg = xi + n * 64;                 // [n][64]
  V* win = mg + n * 64;                // [R][64] (GOS only)
#define AT(p, i) (p)[(i) * 64 + t]

  // ---- load
  if constexpr (FIXED) {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(in) + (size_t)(live ? frame : 0) * n;
    for (int i = 0; i < n; ++i) {
      const uint32_t b = src[i];  // {re[31:16], im[15:0]}: RspChainTesterUtils.scala:105-109
      int re = (int)(short)(b >> 16), im = (int)(short)(b & 0xffffu);
      if (rg.window) {  // Q1.15 coefficient, product rounded half-up (spec section 2.1)
        const int wq = reinterpret_cast<const int16_t*>(rg.window)[i];
        re = (int)(short)((re * wq + (1 << 14)) >> 15);
        im = (int)(short)((im * wq + (1 << 14)) >> 15);
      }
      AT(xr, i) = re;
      AT(xi, i) = im;
    }
  } else {
    const float2* src = reinterpret_cast<const float2*>(in) + (size_t)(live ? frame : 0) * n;
    for (int i = 0; i < n; ++i) {
      const float2 z = src[i];
      const float wv = rg.window ? reinterpret_cast<const float*>(rg.window)[i] : 1.0f;
      AT(xr, i) = z.x * wv;
      AT(xi, i) = z.y * wv;
    }
  }
  // ---- radix-2 DIF, stage by stage, in place (result bit-reversed)
  for (int s = 0; s < log2n; ++s) {
    const int half = n >> (s + 1);
    for (int base = 0; base < n; base += 2 * half) {
      for (int j = 0; j < half; ++j) {
        const int a = base + j, b = a + half, k = j << s;
        if constexpr (FIXED) {
          const uint32_t w = reinterpret_cast<const uint32_t*>(tw)[k];
          const int wr = (int)(short)(w >> 16), wi = (int)(short)(w & 0xffffu);
          const int ar = AT(xr, a), ai = AT(xi, a), br = AT(xr, b), bi = AT(xi, b);
          const int dr = ar - br, di = ai - bi;
          if (rg.keep_lsb_mask | rg.expand_mask) {  // per-stage options: spec section 3, as fft_lds.hpp pass_fx_opt
            const int grow = (int)((rg.expand_mask >> s) & 1u), lsb = !grow && ((rg.keep_lsb_mask >> s) & 1u);
            const int sh = (grow || lsb) ? 0 : 1, wout = 16 + __popc(rg.expand_mask & ((2u << s) - 1u));
            AT(xr, a) = wrap_bits(trim_var((long long)ar + br, sh, rg), wout);
            AT(xi, a) = wrap_bits(trim_var((long long)ai + bi, sh, rg), wout);
            AT(xr, b) = wrap_bits(trim_var((long long)dr * wr - (long long)di * wi, 14 + sh, rg), wout);
            AT(xi, b) = wrap_bits(trim_var((long long)dr * wi + (long long)di * wr, 14 + sh, rg), wout);
          } else {
            AT(xr, a) = (int)(short)trim_s(ar + br, 1, rg.trim_bias1, rg.trim_conv);
            AT(xi, a) = (int)(short)trim_s(ai + bi, 1, rg.trim_bias1, rg.trim_conv);
            AT(xr, b) = (int)(short)trim_s(dr * wr - di * wi, 15, rg.trim_bias15, rg.trim_conv);
            AT(xi, b) = (int)(short)trim_s(dr * wi + di * wr, 15, rg.trim_bias15, rg.trim_conv);
          }
        } else {
          const float2 w = reinterpret_cast<const float2*>(tw)[k];
          const float ar = AT(xr, a), ai = AT(xi, a), br = AT(xr, b), bi = AT(xi, b);
          const float dr = ar - br, di = ai - bi;
          AT(xr, a) = ar + br;
          AT(xi, a) = ai + bi;
          AT(xr, b) = dr * w.x - di * w.y;
          AT(xi, b) = dr * w.y + di * w.x;
        }
      }
    }
  }
  // ---- magnitude in natural order
  const float scale = 1.0f / (float)n;  // F32: net 1/N (FIXED16 halves per stage)
  for (int k = 0; k < n; ++k) {
    // useBitReverse = false: position k of the stream carries the in-place element k = bin bitrev(k)
    const int p = rg.rev_order ? k : (int)(__brev((unsigned)k) >> (32 - log2n));
    if constexpr (FIXED) AT(mg, k) = SmallMath<int>::mag((int)(short)(AT(xr, p) >> rg.growth), (int)(short)(AT(xi, p) >> rg.growth), rg, log_lut);
    else AT(mg, k) = SmallMath<float>::mag(AT(xr, p) * scale, AT(xi, p) * scale, rg, log_lut);
  }
  // ---- CFAR, direct window sums
  auto cell = [&](int j) -> V {
    if (rg.edge) return AT(mg, (j + n) & (n - 1));
    return (j < 0 || j >= n) ? V(0) : AT(mg, j);
  };
  const int R = rg.R, G = rg.G;
  uint32_t found = 0;
  for (int k = 0; k < n; ++k) {
    V st[2];
    for (int side = 0; side < 2; ++side) {
      const int a = side == 0 ? k - G - R : k + G + 1;
      if (rg.algorithm == 1) {  // GOS: insertion-sort the window, take the index-th smallest
        for (int d = 0; d < R; ++d) {
          const V v = cell(a + d);
          int q = d;
          while (q > 0 && AT(win, q - 1) > v) {
            AT(win, q) = AT(win, q - 1);
            --q;
          }
          AT(win, q) = v;
        }
        st[side] = AT(win, side == 0 ? rg.idx_lagg : rg.idx_lead);
      } else if (rg.cfar_mode == 3) {  // CASH: largest sub-window sum
        V best = V(0);
        bool first = true;
        for (int s0 = 0; s0 + rg.sub_window <= R; s0 += rg.sub_window) {
          V ss = V(0);
          for (int d = 0; d < rg.sub_window; ++d) ss += cell(a + s0 + d);
          best = (first || ss > best) ? ss : best;
          first = false;
        }
        st[side] = SmallMath<V>::side(best, rg);
      } else {
        V sum = V(0);
        for (int d = 0; d < R; ++d) sum += cell(a + d);
        st[side] = SmallMath<V>::side(sum, rg);
      }
    }
    V stat;
    if (rg.cfar_mode == 0) stat = SmallMath<V>::half_sum(st[0], st[1]);
    else if (rg.cfar_mode == 1) stat = st[0] > st[1] ? st[0] : st[1];
    else stat = st[0] < st[1] ? st[0] : st[1];
    const V cut = AT(mg, k);
    bool ok = true;
    if (rg.peak_grouping) ok = cut > cell(k - 1) && cut > cell(k + 1);
    const uint32_t word = SmallMath<V>::finish(stat, cut, ok, k, log2n, rg);
    if (live && out) {
      if (rg.send_cut) {  // 64-bit beat {word, cut}
        out[2 * ((size_t)frame * n + k)] = word;
        out[2 * ((size_t)frame * n + k) + 1] = __builtin_bit_cast(uint32_t, cut);
      } else {
        out[(size_t)frame * n + k] = word;
      }
    }
    if (live && fcount && (word & 1u)) {
      if (found < (uint32_t)kFrameDetCap) fdet[(size_t)frame * kFrameDetCap + found] = make_uint2((uint32_t)k, word);
      ++found;
    }
  }
  if (live && fcount) fcount[frame] = found;
#undef AT
}

hipError_t launch_chain1d_small(const Chain1dLaunch& a) {
  if (a.n_frames == 0) return hipSuccess;
  const int n = 1 << a.log2n;
  const size_t lds = (size_t)64 * 4 * (3 * (size_t)n + (a.regs.algorithm == 1 ? (size_t)a.regs.R : 0));
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  const uint32_t grid = (a.n_frames + 63) / 64;
  hipError_t e;
  static LdsGrant granted[2];
  if (a.fixed) {
    auto k = chain1d_small_kernel<true>;
    e = grant_lds(k, lds, a.device, granted[0]);
    if (e != hipSuccess) return e;
    hipExtLaunchKernelGGL(k, dim3(grid), dim3(64), lds, a.stream, a.ev_start, a.ev_stop, 0, a.in, a.out, a.n_frames, a.log2n, a.regs, a.twiddles,
                       a.log_lut, a.frame_count, a.frame_det);
  } else {
    auto k = chain1d_small_kernel<false>;
    e = grant_lds(k, lds, a.device, granted[1]);
    if (e != hipSuccess) return e;
    hipExtLaunchKernelGGL(k, dim3(grid), dim3(64), lds, a.stream, a.ev_start, a.ev_stop, 0, a.in, a.out, a.n_frames, a.log2n, a.regs, a.twiddles,
                       a.log_lut, a.frame_count, a.frame_det);
  }
  return hipGetLastError();
}

}  // namespace rsp
