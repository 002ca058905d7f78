// Quad tail of the fused 1-D chain: the default CA / GO / SO path.
// Same front end; the CFAR stage works on QUADS of 4 consecutive cells so that every LDS access of
// the tail is one conflict-free 16-byte ds_read/ds_write_b128 on UNPADDED images and every word store
// one 16-byte global store (1 KiB per wave-instruction).  Thread tau owns quads tau + T e, e = 0..3
// (cells 4 tau + 4 T e + i) for BOTH the prefix scan and the cells, so a quad's magnitudes are read
// from LDS once and stay in registers:
//   scan   in-quad prefix (3 adds), then an inclusive DPP scan of the quad totals over the wave: for a
//          fixed e the 64 lanes of a wave hold 64 consecutive quads = one 256-cell block, so the
//          block-relative prefix needs no cross-wave step and no second level;
//   cells  6 quad reads (4 prefixes, 2 magnitudes), 1 quad store.
// A window sum needs the exclusive prefix P at two positions.  With refWindow and guardWindow
// multiples of 4 the lagging positions k - G - R, k - G are quad-aligned; the leading ones
// k + G + 1, k + G + R + 1 are off by one, so that side uses P[x + 1] = P[x] + m[x] at the aligned
// x = k + G, k + G + R (two more quad reads of the magnitudes, no second prefix array).
// Halo cells of the quad tail's images, in two sizes: windows with R + G + 4 <= 48 (every reference configuration:
// R = 32, G = 4) take the SMALL one.  The halos are a fixed cost per frame, so they set the occupancy of SMALL
// frames: with 144 / 256-cell halos a 1024-point frame takes 11.4 KiB (three 4-frame workgroups per CU), with
// 48 / 64 cells 9.5 KiB (four) -- 53 -> 45.5 us per 16.7 M cells at 1024 points, 82 -> 43 us at 256 points.
#pragma once
#include "chain_front.hpp"

namespace rsp {

template <bool SMALL> struct QuadHalo {
  static constexpr int MAG = SMALL ? 48 : 144;  // magnitude cells kept right of the frame: >= R + G + 4, a multiple of 16
  static constexpr int PB = SMALL ? 64 : 256;   // prefix cells kept on either side of the frame: >= R + G + 4, a multiple of 16
};
constexpr int kQHalo = QuadHalo<false>::MAG;     // the largest window the quad tail serves: R + G + 4 <= 144

// SHORTW: the prefixes are relative to blocks of 16 cells instead of 256 (fp32, windows of at most 16 cells: a window
// sum is a difference of two prefixes, and its rounding error scales with the length of the prefix, not of the window:
// against 256-cell blocks an 8-cell window needed a 2.5x looser tolerance than every other geometry).
template <int M, bool SMALL, bool SHORTW = false>
struct QuadLds {
  static constexpr int N = 1 << M, QH = QuadHalo<SMALL>::MAG, PBH = QuadHalo<SMALL>::PB;
  static constexpr int BSH = SHORTW ? 4 : 8;          // log2 of the block length
  static constexpr int BS_HALO = SHORTW ? PBH / 16 : 1;  // block totals kept on either side of the frame
  static constexpr int MAG_SLOTS = 16 + N + QH;      // cell x in [-16, N + QH) at x + 16
  static constexpr int PB_SLOTS = N + 2 * PBH;       // cell x in [-PBH, N + PBH) at x + PBH
  static constexpr int BS_SLOTS = (N >> BSH) + 2 * BS_HALO + 1;  // blocks -BS_HALO .. (N >> BSH) + BS_HALO - 1, + one slot that holds 0
  static constexpr int MAG_OFF = 0;
  static constexpr int PB_OFF = MAG_OFF + 4 * MAG_SLOTS;
  static constexpr int BS_OFF = PB_OFF + 4 * PB_SLOTS;
  static constexpr int DET_OFF = (BS_OFF + 4 * BS_SLOTS + 7) & ~7;
  static constexpr int CFAR_BYTES = DET_OFF + 8 + 8 * kFrameDetCap;
  static constexpr int FFT_BYTES = 8 * fft_image_slots(M);
  static constexpr int BYTES = ((CFAR_BYTES > FFT_BYTES ? CFAR_BYTES : FFT_BYTES) + 15) & ~15;
  static constexpr int ROM_BYTES = fx_rom_bytes(M);
};

// the geometry the quad tail is built for (host-side dispatch, launch_m)
__host__ __device__ inline bool quad_tail_supports(int log2n, const ChainRegs& rg) {
  (void)log2n;
  // guardWindowSize a multiple of 4: every window edge of a quad's four cells is one aligned 16-byte LDS read; 2 mod 4
  // (round 3: the reference's run-time guard goes 1 .. 4): the edges are 8-byte aligned, two 8-byte reads, and a quad's
  // halves may sit in different prefix blocks (the G2 instantiations of the cell stage)
  return rg.algorithm == 0 && rg.cfar_mode <= 2 && (rg.R & 3) == 0 && (rg.G & 1) == 0 && rg.R + rg.G + 4 <= kQHalo;
}

template <int CTRL, int RMASK, bool BOUND, typename V>
__device__ __forceinline__ V dpp_v(V v) {
  static_assert(sizeof(V) == 4, "32-bit lanes");
  return __builtin_bit_cast(V, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, RMASK, 0xf, BOUND));
}
// v += (lane 15 of the previous row -> rows 1 and 3) / (lane 31 -> rows 2 and 3): ONE v_add_*_dpp whose
// disabled rows keep their value.  Written as inline asm because the compiler does not fold the masked
// broadcast into the add (it emits v_mov 0 + v_mov_dpp + v_add); the s_nop covers the 2 wait states a DPP
// read needs after a VALU write of the same register, which the compiler does not insert for asm.
template <int BCAST, typename V>
__device__ __forceinline__ V row_bcast_add(V v) {
  static_assert(BCAST == 15 || BCAST == 31, "row_bcast:15 / row_bcast:31");
  if constexpr (std::is_same<V, float>::value) {
    if constexpr (BCAST == 15) asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa" : "+v"(v));
    else asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc" : "+v"(v));
  } else {
    if constexpr (BCAST == 15) asm volatile("s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa" : "+v"(v));
    else asm volatile("s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc" : "+v"(v));
  }
  return v;
}
// inclusive scan over aligned segments of WD = 16, 32 or 64 lanes
template <int WD, typename V>
__device__ __forceinline__ V seg_scan(V v) {
  v += dpp_v<0x111, 0xf, true>(v);  // row_shr:1,2,4,8: inclusive scan of each 16-lane row
  v += dpp_v<0x112, 0xf, true>(v);
  v += dpp_v<0x114, 0xf, true>(v);
  v += dpp_v<0x118, 0xf, true>(v);
  if constexpr (WD >= 32) v = row_bcast_add<15>(v);
  if constexpr (WD >= 64) v = row_bcast_add<31>(v);
  return v;
}

// Dense words of a thread's four quads (cells 4 tau + 4 T e + i) to HBM -- one 16-byte store per quad, 1 KiB per
// wave-instruction; two with sendCut = true: 64-bit beats {word, cut} -- and the frame's detection slots: peaks are
// staged in LDS (det_cnt / det_stage), then written to the per-frame global slots without global atomics.
// Shared by the quad CFAR tail and the GOS kernel's tail.
// (the words and cuts travel BY VALUE: handed over as references to the caller's arrays, the arrays were also written to
// scratch memory -- dead stores the compiler did not remove)
struct QuadWords {
  uint32_t w[16];
  __device__ __forceinline__ uint32_t operator[](int i) const { return w[i]; }
};
template <typename V4> struct QuadCuts {
  V4 q[4];
  __device__ __forceinline__ const V4& operator[](int i) const { return q[i]; }
};
template <int M, typename V, typename V4, typename Hooks>
__device__ __forceinline__ void quad_emit(const QuadWords word, const QuadCuts<V4> cutq, int tau, uint32_t frame, bool live,
                                          const ChainRegs& rg, uint32_t* __restrict__ out, uint32_t* __restrict__ fcount,
                                          uint2* __restrict__ fdet, uint32_t* det_cnt, uint2* det_stage) {
  constexpr int N = 1 << M, T = threads_per_frame(M);
  // ---- dense words: one 16-byte store per quad (1 KiB per wave-instruction) ----
  if (!Hooks::kSerialQuads && live && out && rg.send_cut) {  // sendCut = true: 64-bit beat {word, cut}, two 16-byte stores per quad
    char* obase = reinterpret_cast<char*>(out) + ((size_t)frame * N + 4u * (size_t)tau) * 8u;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const V c0 = cutq[e][0], c1 = cutq[e][1], c2 = cutq[e][2], c3 = cutq[e][3];
      const u32x4 lo4 = {word[4 * e], bits_of(c0), word[4 * e + 1], bits_of(c1)};
      const u32x4 hi4 = {word[4 * e + 2], bits_of(c2), word[4 * e + 3], bits_of(c3)};
      stream_store(lo4, reinterpret_cast<u32x4*>(obase + (size_t)(32 * T * e)));
      stream_store(hi4, reinterpret_cast<u32x4*>(obase + (size_t)(32 * T * e) + 16));
    }
  } else if (live && out) {
    char* obase = reinterpret_cast<char*>(out);
    const uint32_t ooff = (frame * (uint32_t)N + 4u * (uint32_t)tau) * 4u;
    wave_prio(1);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const u32x4 w4 = {word[4 * e], word[4 * e + 1], word[4 * e + 2], word[4 * e + 3]};
      stream_store(w4, reinterpret_cast<u32x4*>(obase + (size_t)ooff + (size_t)(16 * T * e)));
    }
    wave_prio(0);
  }
  if (!kCountPath && fcount) {
    uint32_t any = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) any |= word[j];
    if (any & 1u) {  // rare: ~1 peak per 1000 cells; kept compact (a loop, not 16 unrolled copies)
      uint32_t hits = 0;
#pragma unroll
      for (int j = 0; j < 16; ++j) hits |= (word[j] & 1u) << j;
      uint32_t slot = atomicAdd(det_cnt, (uint32_t)__popc(hits));  // one LDS atomic per thread, not per peak (dense scenes)
      while (hits) {
        const int j = __ffs(hits) - 1;
        hits &= hits - 1;
        uint32_t w = word[0];
#pragma unroll
        for (int q = 1; q < 16; ++q) w = (j == q) ? word[q] : w;
        if (slot < (uint32_t)kFrameDetCap)
          det_stage[slot] = make_uint2((uint32_t)(4 * tau + 4 * T * (j >> 2) + (j & 3)), w);
        ++slot;
      }
    }
    // per-frame detection slots (no global atomics): count + first kFrameDetCap peaks
    Hooks::barrier();
    if (live) {
      const uint32_t cnt = *det_cnt;
      if (tau == 0) fcount[frame] = cnt;
      for (uint32_t i = tau; i < min(cnt, (uint32_t)kFrameDetCap); i += T)
        fdet[(size_t)frame * kFrameDetCap + i] = det_stage[i];
    }
  }
}

// The tail proper: mg[] (front_end's register order) -> magnitude image -> scan -> cells -> dense words to HBM
// (+ per-frame detection slots).  `fbase` = the frame's LDS (QuadLds<M, SMALL>), free to be overwritten once
// every thread has passed the first barrier below.  Barriers are the hook policy's (Hooks::barrier(): __syncthreads()
// in the plain kernels; the pipelined kernel of chain1d_pipe.hip keeps LDS-DMA loads in flight across the tail and
// brings a barrier without a vector-memory wait).
// the four scans of a thread (its quads e = 0..3) step by step SIDE BY SIDE: a DPP read needs two wait states behind
// the VALU write of its source, and four independent chains cover them for each other -- one chain after the other
// (seg_scan per quad) had an s_nop behind nearly every one of its 24 DPP adds
template <int WD, typename V>
__device__ __forceinline__ void seg_scan_x4(V (&v)[4]) {
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] += dpp_v<0x111, 0xf, true>(v[e]);
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] += dpp_v<0x112, 0xf, true>(v[e]);
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] += dpp_v<0x114, 0xf, true>(v[e]);
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] += dpp_v<0x118, 0xf, true>(v[e]);
  if constexpr (std::is_same<V, float>::value) {
    if constexpr (WD >= 32)
      asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa\n\tv_add_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa\n\t"
                   "v_add_f32_dpp %2, %2, %2 row_bcast:15 row_mask:0xa\n\tv_add_f32_dpp %3, %3, %3 row_bcast:15 row_mask:0xa"
                   : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
    if constexpr (WD >= 64)
      asm volatile("s_nop 0\n\tv_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc\n\tv_add_f32_dpp %1, %1, %1 row_bcast:31 row_mask:0xc\n\t"
                   "v_add_f32_dpp %2, %2, %2 row_bcast:31 row_mask:0xc\n\tv_add_f32_dpp %3, %3, %3 row_bcast:31 row_mask:0xc"
                   : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
  } else {
    if constexpr (WD >= 32)
      asm volatile("s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa\n\tv_add_u32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa\n\t"
                   "v_add_u32_dpp %2, %2, %2 row_bcast:15 row_mask:0xa\n\tv_add_u32_dpp %3, %3, %3 row_bcast:15 row_mask:0xa"
                   : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
    if constexpr (WD >= 64)
      asm volatile("s_nop 0\n\tv_add_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc\n\tv_add_u32_dpp %1, %1, %1 row_bcast:31 row_mask:0xc\n\t"
                   "v_add_u32_dpp %2, %2, %2 row_bcast:31 row_mask:0xc\n\tv_add_u32_dpp %3, %3, %3 row_bcast:31 row_mask:0xc"
                   : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
  }
}

// inclusive scan over aligned groups of 4 lanes (16 cells): row shifts by 1 and 2, kept out of the lanes whose source
// sits in the neighbouring group
__device__ __forceinline__ float seg4_scan(float v, unsigned tid) {
  const float a = dpp_v<0x111, 0xf, true>(v);
  v += (tid & 3u) ? a : 0.0f;
  const float b = dpp_v<0x112, 0xf, true>(v);
  v += (tid & 2u) ? b : 0.0f;
  return v;
}

template <int M, bool FIXED, bool SMALL, bool SHORTW = false, typename V, typename Hooks>
__device__ __forceinline__ void quad_tail(unsigned char* fbase, const V (&mg)[16], int tau, uint32_t frame, bool live,
                                          const ChainRegs& rg, uint32_t* __restrict__ out,
                                          uint32_t* __restrict__ fcount, uint2* __restrict__ fdet, Hooks& hk) {
  constexpr int N = 1 << M, T = threads_per_frame(M);
  constexpr int WD = T < 64 ? T : 64;  // lanes of a wave that belong to one frame
  constexpr int SPB = 64 / WD;         // lane segments (values of e) per 256-cell block
  using L = QuadLds<M, SMALL, SHORTW>;
  constexpr int QH = L::QH, PBH = L::PBH, BSH = L::BSH, BH = L::BS_HALO;
  constexpr int NB = N >> BSH;          // blocks per frame
  constexpr int ZS = NB + BH;           // the slot that holds 0: "no block total to add"
  using V4 = typename Vec4<V>::type;
  static_assert(!SHORTW || (!FIXED && SMALL), "16-cell blocks: fp32, windows of at most 16 cells");

  V* mag = reinterpret_cast<V*>(fbase + L::MAG_OFF) + 16;   // mag[x], x in [-16, N + QH)
  V* pb = reinterpret_cast<V*>(fbase + L::PB_OFF) + PBH;    // pb[x], x in [-PBH, N + PBH)
  V* bs = reinterpret_cast<V*>(fbase + L::BS_OFF) + BH;     // bs[-BH] .. bs[NB + BH - 1], bs[ZS] = 0
  uint32_t* det_cnt = reinterpret_cast<uint32_t*>(fbase + L::DET_OFF);
  uint2* det_stage = reinterpret_cast<uint2*>(fbase + L::DET_OFF + 8);
  const bool wrap = rg.edge != 0;
  Hooks::barrier();  // every thread is done reading the FFT image this overlays
  {  // magnitudes to LDS in natural bin order: register (g, p) holds bin (bitrev(p) << (M - WL)) | (g T + tau)
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
  if (tau == 0) {
    *det_cnt = 0u;
    bs[ZS] = V(0);  // the "no block total" slot of the window fix-ups
  }
  Hooks::barrier();
  hk.stamp(8);

  // ---- scan: block-relative exclusive prefix sums of the thread's 4 quads ----
  V4 mq[4];
  {
    const V4 zero4 = {V(0), V(0), V(0), V(0)};
    V inc[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) mq[e] = *reinterpret_cast<const V4*>(mag + 4 * (tau + T * e));
    V p1[4], p2[4], p3[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      p1[e] = mq[e][0];
      p2[e] = p1[e] + mq[e][1];
      p3[e] = p2[e] + mq[e][2];
      inc[e] = p3[e] + mq[e][3];
    }
    if constexpr (SHORTW) {
#pragma unroll
      for (int e = 0; e < 4; ++e) inc[e] = seg4_scan(inc[e], threadIdx.x);
    } else {
      seg_scan_x4<WD, V>(inc);
    }
    V tot[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int q = tau + T * e;  // quad index; block q >> 6, position q & 63
      tot[e] = inc[e];            // inclusive through this quad, within the lane segment
      V exc = inc[e] - (p3[e] + mq[e][3]);
      if constexpr (SPB > 1 && !SHORTW) {  // a block spans SPB values of e (frames of 256 / 512 points): carry the earlier ones
        V carry = V(0);
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2) {
          if (e2 < e && e2 >= e - e % SPB)
            carry += __builtin_bit_cast(V, __shfl(__builtin_bit_cast(int, inc[e2]), (threadIdx.x & 63 & ~(WD - 1)) | (WD - 1)));
        }
        exc += carry;
        tot[e] += carry;
      }
      const V4 pq = V4{V(0), p1[e], p2[e], p3[e]} + exc;
      *reinterpret_cast<V4*>(pb + 4 * q) = pq;
      // halos: zeros, or the wrapped image of the first / last block (prefixes), of the first QH
      // cells (magnitudes right of the frame) and of the last cell (left neighbour of cell 0)
      if (q < PBH / 4) {
        *reinterpret_cast<V4*>(pb + 4 * q + N) = wrap ? pq : zero4;
        if (q < QH / 4) *reinterpret_cast<V4*>(mag + 4 * q + N) = wrap ? mq[e] : zero4;
      }
      if (q >= N / 4 - PBH / 4) *reinterpret_cast<V4*>(pb + 4 * q - N) = wrap ? pq : zero4;
      if (q == N / 4 - 1) mag[-1] = wrap ? mq[e][3] : V(0);
    }
    if constexpr (SHORTW) {
      // block totals of the 16-cell blocks: the lane that holds a block's fourth quad, + the halo images
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int q = tau + T * e;
        if ((q & 3) == 3) {
          const int blk = q >> 2;
          bs[blk] = tot[e];
          if (blk < BH) bs[blk + NB] = wrap ? tot[e] : V(0);
          if (blk >= NB - BH) bs[blk - NB] = wrap ? tot[e] : V(0);
        }
      }
    } else {
    // block totals: the lane that holds a block's last quad (for T >= 64 the same lane for every e)
    auto block_total = [&](int e) {
      const int blk = (tau + T * e) >> 6;
      bs[blk] = tot[e];
      if (blk == N / 256 - 1) bs[-1] = wrap ? tot[e] : V(0);
      if (blk == 0) bs[N / 256] = wrap ? tot[e] : V(0);
    };
    if constexpr (T % 64 == 0) {
      if ((tau & 63) == 63) {
#pragma unroll
        for (int e = 0; e < 4; ++e) block_total(e);
      }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (((tau + T * e) & 63) == 63) block_total(e);
    }
    }
  }
  Hooks::barrier();

  hk.stamp(9);
  // ---- CFAR on quads: cells k0 + i, k0 = 4 tau + 4 T e, i = 0..3 ----
  uint32_t word[16];
  {
    const int R = rg.R, G = rg.G;
    const int k00 = 4 * tau;
    const V* pa = pb + (k00 - G);       // P[k - G]        lagging end
    const V* pbq = pb + (k00 - G - R);  // P[k - G - R]    lagging start
    const V* pe = pb + (k00 + G + R);   // P[k + G + R]    + m = P[k + G + R + 1] leading end
    const V* ps = pb + (k00 + G);       // P[k + G]        + m = P[k + G + 1]     leading start
    const V* me = mag + (k00 + G + R);
    const V* ms = mag + (k00 + G);
    const V* mc = mag + k00;
    constexpr int ES = 4 * T;  // cells between a thread's consecutive quads
    const float kA = rg.linear ? rg.div_f * rg.scaler_f : rg.div_f, kB = rg.linear ? 0.0f : rg.scaler_f;
    // block (256 cells; 16 with SHORTW) of the two window starts, and whether the window ends in the next block: then
    // the start block's total is added.  A quad never straddles a block, so this is per quad; the
    // "no" case reads the slot that holds 0, which keeps the read unconditional (no divergent branch).
    auto cells = [&](auto mode_c, auto group_c, auto g2_c) {
      constexpr int MODE = decltype(mode_c)::value;
      constexpr bool GROUP = decltype(group_c)::value;
      // G2: guardWindowSize = 2 mod 4.  The window edges of a quad's cells start two cells off quad alignment: 8-byte
      // reads, and cells 0,1 / 2,3 of the quad (NH = 2 halves) have their own prefix blocks and fix-ups.
      constexpr bool G2 = decltype(g2_c)::value;
      constexpr int NH = G2 ? 2 : 1;
      typedef V V2 __attribute__((ext_vector_type(2)));
      auto ld4 = [](const V* p) -> V4 {
        if constexpr (G2) {
          const V2 lo = *reinterpret_cast<const V2*>(p), hi = *reinterpret_cast<const V2*>(p + 2);
          return V4{lo[0], lo[1], hi[0], hi[1]};
        } else {
          return *reinterpret_cast<const V4*>(p);
        }
      };
      int i0[4][NH], i1[4][NH];
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        const int kh = k00 + 2 * h;  // first cell of the half
        if constexpr ((4 * T) % (1 << BSH) == 0) {  // a thread's quads sit whole blocks apart: same case for all four
          const int bu0 = (kh - G - R) >> BSH, bu1 = (kh + G) >> BSH;
          const bool z0 = ((kh - G) >> BSH) == bu0, z1 = ((kh + G + R) >> BSH) == bu1;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            i0[e][h] = z0 ? ZS : bu0 + ((4 * T) >> BSH) * e;
            i1[e][h] = z1 ? ZS : bu1 + ((4 * T) >> BSH) * e;
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int k0 = kh + 4 * T * e;
            const int bu0 = (k0 - G - R) >> BSH, bu1 = (k0 + G) >> BSH;
            i0[e][h] = ((k0 - G) >> BSH) != bu0 ? bu0 : ZS;
            i1[e][h] = ((k0 + G + R) >> BSH) != bu1 ? bu1 : ZS;
          }
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int k0 = k00 + 4 * T * e;
        const V4 Pa = ld4(pa + ES * e), Pb = ld4(pbq + ES * e);
        const V4 Pe = ld4(pe + ES * e), Ps = ld4(ps + ES * e);
        const V4 Me = ld4(me + ES * e), Ms = ld4(ms + ES * e);
        // (kSerialQuads: the quad's own magnitudes are read again instead of being carried from the scan in registers)
        const V4 cut = Hooks::kSerialQuads ? *reinterpret_cast<const V4*>(mc + ES * e) : mq[e];
        V f0h[2], f1h[2];  // block fix-ups of the quad's two halves (equal unless G2)
        f0h[0] = bs[i0[e][0]];
        f1h[0] = bs[i1[e][0]];
        f0h[1] = G2 ? bs[i0[e][NH - 1]] : f0h[0];
        f1h[1] = G2 ? bs[i1[e][NH - 1]] : f1h[0];
        V nl = V(0), nr = V(0);
        if constexpr (GROUP) {
          nl = mc[ES * e - 1];
          nr = mc[ES * e + 4];
        }
        if constexpr (!FIXED) {
          // two cells per packed op: sums, combination and threshold of a quad in 14-16 v_pk ops
          const f32x2 kAA = {MODE == 0 ? kA * 0.5f : kA, MODE == 0 ? kA * 0.5f : kA}, kBB = {kB, kB};
          auto half = [&](auto hc) {
            constexpr int h = decltype(hc)::value;
            const float f0 = f0h[h], f1 = f1h[h];
            const f32x2 f00 = {f0, f0}, f11 = {f1, f1}, f01 = {f0 + f1, f0 + f1};
            const f32x2 a = __builtin_shufflevector(Pa, Pa, 2 * h, 2 * h + 1), b = __builtin_shufflevector(Pb, Pb, 2 * h, 2 * h + 1);
            const f32x2 pe2 = __builtin_shufflevector(Pe, Pe, 2 * h, 2 * h + 1), ps2 = __builtin_shufflevector(Ps, Ps, 2 * h, 2 * h + 1);
            const f32x2 me2 = __builtin_shufflevector(Me, Me, 2 * h, 2 * h + 1), ms2 = __builtin_shufflevector(Ms, Ms, 2 * h, 2 * h + 1);
            const f32x2 c2 = __builtin_shufflevector(cut, cut, 2 * h, 2 * h + 1);
            const f32x2 lag = a - b;
            const f32x2 lead = (pe2 - ps2) + (me2 - ms2);
            f32x2 thr2;
            if constexpr (MODE == 0) {
              thr2 = __builtin_elementwise_fma((lag + lead) + f01, kAA, kBB);
            } else {
              const f32x2 lg = lag + f00, ld = lead + f11;
              const f32x2 comb = {MODE == 1 ? fmaxf(lg.x, ld.x) : fminf(lg.x, ld.x),
                                  MODE == 1 ? fmaxf(lg.y, ld.y) : fminf(lg.y, ld.y)};
              thr2 = __builtin_elementwise_fma(comb, kAA, kBB);
            }
            if constexpr (!GROUP) {
              // cut > thr  <=>  thr - cut < 0: the sign bit of the (correctly rounded, never flushed: both
              // operands are normal and differ by >= 1 ulp) difference IS the peak flag
              const f32x2 d = thr2 - c2;
              word[4 * e + 2 * h] = (__float_as_uint(thr2.x) & ~1u) | (__float_as_uint(d.x) >> 31);
              word[4 * e + 2 * h + 1] = (__float_as_uint(thr2.y) & ~1u) | (__float_as_uint(d.y) >> 31);
            } else {
#pragma unroll
              for (int u = 0; u < 2; ++u) {
                const int i = 2 * h + u;
                const bool group_ok = cut[i] > (i == 0 ? nl : cut[i - 1]) && cut[i] > (i == 3 ? nr : cut[i + 1]);
                const float thr = thr2[u];
                const uint32_t peak = (cut[i] > thr) && group_ok;
                word[4 * e + i] = (__float_as_uint(thr) & ~1u) | peak;
              }
            }
          };
          half(std::integral_constant<int, 0>{});
          half(std::integral_constant<int, 1>{});
        } else {
          const V4 lag = (Pa - Pb) + V4{f0h[0], f0h[0], f0h[1], f0h[1]};
          const V4 lead = ((Pe - Ps) + (Me - Ms)) + V4{f1h[0], f1h[0], f1h[1], f1h[1]};
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            bool group_ok = true;
            if constexpr (GROUP) group_ok = cut[i] > (i == 0 ? nl : cut[i - 1]) && cut[i] > (i == 3 ? nr : cut[i + 1]);
            const V sl = CfarMath<V>::side(lag[i], rg), sd = CfarMath<V>::side(lead[i], rg);
            V stat;
            if constexpr (MODE == 0) stat = CfarMath<V>::half_sum(sl, sd);
            else if constexpr (MODE == 1) stat = sl > sd ? sl : sd;
            else stat = sl < sd ? sl : sd;
            word[4 * e + i] = CfarMath<V>::finish(stat, cut[i], group_ok, k0 + i, M, rg);
#ifdef RSP_DBG_OUT  // debugging side builds only: expose the tail's inputs instead of the words
            word[4 * e + i] = (uint32_t)(RSP_DBG_OUT == 1 ? cut[i] : RSP_DBG_OUT == 2 ? Pa[i] : RSP_DBG_OUT == 3 ? Me[i] : Pe[i]);
#endif
          }
          // one quad's loads at a time: hoisting all four quads' 24 LDS reads above the 64-bit threshold arithmetic
          // took the FIXED16 kernel to 142 VGPRs (three workgroups per CU at 4096 points instead of four)
          __builtin_amdgcn_sched_barrier(0);
        }
        // the pipelined kernel holds its prefetched samples (32 VGPRs) through the tail: quad by quad there too
        if constexpr (!FIXED && Hooks::kSerialQuads) __builtin_amdgcn_sched_barrier(0);
      }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    auto by_mode = [&](auto group_c, auto g2_c) {
      if (rg.cfar_mode == 0) cells(I0{}, group_c, g2_c);
      else if (rg.cfar_mode == 1) cells(I1{}, group_c, g2_c);
      else cells(I2{}, group_c, g2_c);
    };
    const bool g2 = (G & 3) == 2;
    if constexpr (kCountPath) {  // static instruction counts of ONE path (tools/count_insts.sh): CA, no grouping
      cells(I0{}, std::false_type{}, std::false_type{});
    } else if constexpr (Hooks::kSerialQuads) {  // (the pipelined experiment: aligned guards, no peak grouping)
      by_mode(std::false_type{}, std::false_type{});
    } else if (rg.peak_grouping) {
      if (g2) by_mode(std::true_type{}, std::true_type{});
      else by_mode(std::true_type{}, std::false_type{});
    } else {
      if (g2) by_mode(std::false_type{}, std::true_type{});
      else by_mode(std::false_type{}, std::false_type{});
    }
  }

  hk.stamp(10);
  hk.before_stores();
  {
    QuadWords ww;
    QuadCuts<V4> cc;
#pragma unroll
    for (int j = 0; j < 16; ++j) ww.w[j] = word[j];
#pragma unroll
    for (int e = 0; e < 4; ++e) cc.q[e] = mq[e];
    quad_emit<M, V, V4, Hooks>(ww, cc, tau, frame, live, rg, out, fcount, fdet, det_cnt, det_stage);
  }
  hk.stamp(11);
  hk.report();
}

template <int M, bool FIXED, int FX, bool SMALL, bool SHORTW = false>
__global__ void __launch_bounds__(wg_size(M))
chain1d_quad_kernel(const void* __restrict__ in, uint32_t* __restrict__ out, uint32_t n_frames,
                    ChainRegs rg, const void* __restrict__ tw, const int16_t* __restrict__ log_lut,
                    uint32_t* __restrict__ fcount, uint2* __restrict__ fdet) {
  constexpr int T = threads_per_frame(M), FPW = frames_per_wg(M);
  using L = QuadLds<M, SMALL, SHORTW>;
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
  hk.stamp(0);
// (side builds, -DRSP_ABLATE: mask bit 5 reads 64 L2-resident frames instead of the batch, bit 6 drops the word stores)
  front_end<M, FIXED, V, FX>(in, hk.off(5) ? (frame & 63u) : frame, live, tau, fbase, rg, tw, log_lut,
                             reinterpret_cast<uint2*>(smem + (FIXED ? FixedRom<M, FX>::off(L::BYTES) : 0)), mg, hk);
  hk.stamp(7);
  quad_tail<M, FIXED, SMALL, SHORTW>(fbase, mg, tau, frame, live, rg, hk.off(6) ? nullptr : out, fcount, fdet, hk);
}

}  // namespace rsp
