// Data-type independent part of the 1-D chain: the launch splitter and the detection compaction kernels.
#include <hip/hip_runtime.h>

#include "chain_regs.hpp"
#include "kernels.hpp"

namespace rsp {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

hipError_t launch_chain1d_part_f32(const Chain1dLaunch& a);
extern "C" int rsp_experiment_chain1d(const Chain1dLaunch* a, hipError_t* err) __attribute__((weak));
hipError_t launch_chain1d_part_fx0(const Chain1dLaunch& a);
hipError_t launch_chain1d_part_fx1(const Chain1dLaunch& a);
hipError_t launch_chain1d_part_fx2(const Chain1dLaunch& a);
static hipError_t launch_chain1d_part_fx(const Chain1dLaunch& a) {
  if (a.regs.keep_lsb_mask | a.regs.expand_mask) return launch_chain1d_part_fx2(a);
  return a.regs.trim_conv ? launch_chain1d_part_fx0(a) : launch_chain1d_part_fx1(a);
}

hipError_t launch_chain1d(const Chain1dLaunch& a0) {
  if (a0.n_frames == 0) return hipSuccess;
  if (a0.log2n < kMinLog2N) return launch_chain1d_small(a0);
  // kernels address a launch's input with 32-bit byte offsets: split into < 4 GiB pieces
  const uint64_t beat = a0.fixed ? 4 : 8;
  uint32_t max_frames = (uint32_t)((0xFFFFFFFFull / (beat << a0.log2n)) & ~63ull);
  if (a0.max_frames_per_launch) {  // whole workgroups only, never zero
    const uint32_t want = (a0.max_frames_per_launch + 63u) & ~63u;
    if (want < max_frames) max_frames = want;
  }
  Chain1dLaunch a = a0;
  for (uint32_t done = 0; done < a0.n_frames; done += max_frames) {
    a.n_frames = a0.n_frames - done < max_frames ? a0.n_frames - done : max_frames;
    a.in = static_cast<const char*>(a0.in) + ((uint64_t)done << a0.log2n) * beat;
    a.out = a0.out ? a0.out + (((uint64_t)done << a0.log2n) << (a0.regs.send_cut ? 1 : 0)) : nullptr;
    a.frame_count = a0.frame_count ? a0.frame_count + done : nullptr;
    a.frame_det = a0.frame_det ? a0.frame_det + (uint64_t)done * kFrameDetCap : nullptr;
    a.ev_start = done == 0 ? a0.ev_start : nullptr;                          // first piece starts the clock,
    a.ev_stop = done + max_frames >= a0.n_frames ? a0.ev_stop : nullptr;     // the last one stops it
    hipError_t e = hipSuccess;
    // side libraries of tools/experiments/ (kept buildable so that recorded negative results can be re-measured)
    // define this symbol; the product library does not, and the branch is never taken
    if (!(rsp_experiment_chain1d && a.experiment && rsp_experiment_chain1d(&a, &e)))
      e = a.fixed ? launch_chain1d_part_fx(a) : launch_chain1d_part_f32(a);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

// ---------------------------------------------------------------- detection compaction

// Last-workgroup epilogue shared by both compaction kernels: counters = {found, cursor, ticket}, all
// zero on entry.  Thread 0 of every workgroup has added its share with RETURNING device-scope atomics
// (so they have been performed when it goes on) before it takes a ticket; the workgroup that draws the
// last ticket publishes {found, stored} and re-zeroes the counters for the next launch on this stream.
// No __threadfence: only the counters travel between workgroups, and they are only ever touched by
// device-scope atomics -- an agent-scope fence per workgroup writes back / invalidates L2 on this
// multi-XCD part and cost 180 ns per workgroup (750 us for the 4096 workgroups of a 67 M-cell map).
__device__ __forceinline__ void publish_counts(uint32_t* counters, uint32_t cap, uint32_t* d_count) {
  if (threadIdx.x == 0) {
    const uint32_t ticket = atomicAdd(&counters[2], 1u);
    if (ticket == gridDim.x - 1) {
      const uint32_t found = atomicExch(&counters[0], 0u);
      const uint32_t cursor = atomicExch(&counters[1], 0u);
      atomicExch(&counters[2], 0u);
      d_count[0] = found;
      d_count[1] = cursor < cap ? cursor : cap;
    }
  }
}

// found / cursor shares of a workgroup (thread 0 only): returning atomics, see publish_counts
__device__ __forceinline__ uint32_t reserve_block(uint32_t* counters, uint32_t found, uint32_t entries) {
  uint32_t base = 0u;
  if (found) {
    const uint32_t r = atomicAdd(&counters[0], found);
    asm volatile("" ::"v"(r));  // keep the return value live: the wave waits for the atomic
  }
  if (entries) base = atomicAdd(&counters[1], entries);
  return base;
}

// Per-frame slots written by the chain kernels -> one compact list, frames in ascending order.  One thread per frame,
// 256 frames per workgroup.  PREFIX = true (up to kPrefixFrames frames): every workgroup sums the counts of ALL
// frames before its own (a few KiB of L2 reads, all in flight together) instead of reserving its block of the list with
// device-scope atomics -- no counters, no ticket, a deterministic list, and three atomic round trips (~2.5 us of an
// 8 us launch) off the critical path; the last workgroup publishes {found, stored}.  PREFIX = false: a block of the
// list is reserved with one returning atomic per workgroup (blocks land in completion order).
constexpr uint32_t kPrefixFrames = 16384;

// Frames whose peaks did not fit their slots (> kFrameDetCap: thresholds near the noise floor, clutter) are re-read from
// the dense words.  Left to the slot stage's 16 workgroups -- one frame after the other, 256 frames each -- a scene in which
// every frame overflows took 2.2 ms per 4096 x 4096 batch (48x the chain kernel).  The launch therefore carries helper
// workgroups, 4 frames each: a helper reads its frames' counts, leaves at once when none overflows (the usual case: one
// memory round trip next to the slot stage's own), and otherwise takes each overflowing frame's base from the same
// prefix over the counts the slot stage uses -- no queue, no inter-workgroup signal, the list's order unchanged.
constexpr uint32_t kOvfFramesPerWg = 4;
__device__ __forceinline__ void overflow_helper(const uint32_t* __restrict__ fcount, uint32_t n_frames,
                                                const uint32_t* __restrict__ words, int log2n, int word_shift,
                                                rsp_detection* __restrict__ list, uint32_t cap, uint32_t h) {
  __shared__ uint32_t cnt_sh[kOvfFramesPerWg], part[4], cursor;
  const uint32_t f0 = h * kOvfFramesPerWg;
  if (f0 >= n_frames || !words) return;
  if (threadIdx.x < kOvfFramesPerWg) cnt_sh[threadIdx.x] = f0 + threadIdx.x < n_frames ? fcount[f0 + threadIdx.x] : 0u;
  __syncthreads();
  bool any = false;
#pragma unroll
  for (uint32_t j = 0; j < kOvfFramesPerWg; ++j) any |= cnt_sh[j] > (uint32_t)kFrameDetCap;
  if (!any) return;  // uniform
  // peaks of all the frames before f0 (a multiple of 4): with dense words every frame contributes its full count
  uint32_t pre = 0u;
  const u32x4* fc4 = reinterpret_cast<const u32x4*>(fcount);
  for (uint32_t i = threadIdx.x; i < f0 / 4; i += 256) {
    const u32x4 c = fc4[i];
    pre += (c.x + c.y) + (c.z + c.w);
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) pre += __shfl_xor(pre, d);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = pre;
  __syncthreads();
  uint32_t base = part[0] + part[1] + part[2] + part[3];
  for (uint32_t j = 0; j < kOvfFramesPerWg; ++j) {
    const uint32_t c = cnt_sh[j];
    if (c > (uint32_t)kFrameDetCap) {  // uniform
      if (threadIdx.x == 0) cursor = 0u;
      __syncthreads();
      const uint32_t of = f0 + j;
      const uint32_t* row = words + (((size_t)of << log2n) << word_shift);  // word_shift = 1: 64-bit beats {word, cut}
      for (uint32_t x = threadIdx.x; x < (1u << log2n); x += 256) {
        const uint32_t w = row[(size_t)x << word_shift];
        if (w & 1u) {
          const uint32_t slot = base + atomicAdd(&cursor, 1u);
          if (slot < cap) list[slot] = rsp_detection{of, x, 0u, w};
        }
      }
      __syncthreads();
    }
    base += c;
  }
}

template <bool PREFIX>
__global__ void __launch_bounds__(256)
compact_frames_kernel(const uint32_t* __restrict__ fcount, const uint2* __restrict__ fdet,
                      uint32_t n_frames, const uint32_t* __restrict__ words, int log2n, int word_shift,
                      rsp_detection* __restrict__ list, uint32_t cap,
                      uint32_t* __restrict__ counters, uint32_t* __restrict__ d_count) {
  __shared__ uint32_t wave_tot[4], wave_found[4], wave_pre[4], wave_pref[4];
  __shared__ uint32_t base_sh, ovf_n, ovf_cursor;
  __shared__ uint32_t ovf_frame[256], ovf_base[256];
  const uint32_t n_main = (n_frames + 255) / 256;  // workgroups of the slot stage; any further ones are overflow helpers
  if constexpr (PREFIX) {
    if (blockIdx.x >= n_main) {
      overflow_helper(fcount, n_frames, words, log2n, word_shift, list, cap, blockIdx.x - n_main);
      return;
    }
  }
  const uint32_t f = blockIdx.x * 256 + threadIdx.x;
  const uint32_t found = f < n_frames ? fcount[f] : 0u;
  u32x4 early[8];  // slots 0..15 of the frame, requested before the count is known (stale slots are never used)
  {
    const u32x4* src = reinterpret_cast<const u32x4*>(fdet + (size_t)(f < n_frames ? f : 0) * kFrameDetCap);
#pragma unroll
    for (int j = 0; j < 8; ++j) early[j] = src[j];
  }
  // a frame whose peaks did not fit its slots is re-read from the dense words when there are any
  const bool ovf = found > (uint32_t)kFrameDetCap;
  const uint32_t mine = (ovf && !words) ? (uint32_t)kFrameDetCap : found;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) ovf_n = 0u;
  uint32_t pre = 0, pre_found = 0;  // entries / peaks of the frames before this workgroup's (partial sums per thread)
  if constexpr (PREFIX) {
    const uint32_t before = blockIdx.x * 256u;  // a multiple of 4, all of them < n_frames
    const u32x4* fc4 = reinterpret_cast<const u32x4*>(fcount);
    for (uint32_t i = threadIdx.x; i < before / 4; i += 256) {
      const u32x4 c = fc4[i];
      pre_found += (c.x + c.y) + (c.z + c.w);
      if (words) pre += (c.x + c.y) + (c.z + c.w);
      else pre += min(c.x, (uint32_t)kFrameDetCap) + min(c.y, (uint32_t)kFrameDetCap) + min(c.z, (uint32_t)kFrameDetCap) + min(c.w, (uint32_t)kFrameDetCap);
    }
  }
  uint32_t inc = mine, tot_found = found;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t t = __shfl_up(inc, d);
    if (lane >= d) inc += t;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    tot_found += __shfl_xor(tot_found, d);
    if constexpr (PREFIX) {
      pre += __shfl_xor(pre, d);
      pre_found += __shfl_xor(pre_found, d);
    }
  }
  if (lane == 63) wave_tot[wave] = inc;
  if (lane == 0) {
    wave_found[wave] = tot_found;
    wave_pre[wave] = pre;
    wave_pref[wave] = pre_found;
  }
  __syncthreads();
  uint32_t off = inc - mine;
  for (int w = 0; w < wave; ++w) off += wave_tot[w];
  const uint32_t tot = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
  const uint32_t fnd = wave_found[0] + wave_found[1] + wave_found[2] + wave_found[3];
  uint32_t base;
  if constexpr (PREFIX) {
    const uint32_t wg_base = wave_pre[0] + wave_pre[1] + wave_pre[2] + wave_pre[3];
    base = wg_base + off;
    if (blockIdx.x == n_main - 1 && threadIdx.x == 0) {
      const uint32_t cursor = wg_base + tot;
      d_count[0] = wave_pref[0] + wave_pref[1] + wave_pref[2] + wave_pref[3] + fnd;
      d_count[1] = cursor < cap ? cursor : cap;
    }
  } else {
    if (threadIdx.x == 0) base_sh = reserve_block(counters, fnd, tot);
    __syncthreads();
    base = base_sh + off;
  }
  if (ovf && words) {
    if constexpr (!PREFIX) {  // (PREFIX: the overflow helpers of this launch re-read such frames, 4 frames per workgroup)
      const uint32_t s = atomicAdd(&ovf_n, 1u);
      ovf_frame[s] = f;
      ovf_base[s] = base;
    }
  } else {
    // the first 16 slots were requested together with the count (one memory round trip for all but ~1e-4 of the
    // frames at 5 peaks per frame); later slots four per round (two 16-byte loads)
    auto put = [&](uint32_t i, const u32x4& e01, const u32x4& e23) {
      const uint32_t bins[4] = {e01.x, e01.z, e23.x, e23.z}, wds[4] = {e01.y, e01.w, e23.y, e23.w};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (i + q < mine && base + i + q < cap) {
          rsp_detection d;
          d.frame = f;
          d.bin = bins[q];
          d.doppler = 0;
          d.word = wds[q];
          list[base + i + q] = d;
        }
      }
    };
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (4u * r < mine) put(4u * r, early[2 * r], early[2 * r + 1]);
    const u32x4* src = reinterpret_cast<const u32x4*>(fdet + (size_t)f * kFrameDetCap);
    for (uint32_t i = 16; i < mine; i += 4) put(i, src[i / 2], (i + 2 < mine) ? src[i / 2 + 1] : u32x4{0u, 0u, 0u, 0u});
  }
  __syncthreads();
  // overflow frames (rare: > kFrameDetCap peaks in one frame): the whole workgroup re-reads the frame
  const uint32_t n_ovf = ovf_n;
  for (uint32_t q = 0; q < n_ovf; ++q) {
    if (threadIdx.x == 0) ovf_cursor = 0u;
    __syncthreads();
    const uint32_t of = ovf_frame[q], ob = ovf_base[q];
    const uint32_t* row = words + (((size_t)of << log2n) << word_shift);  // word_shift = 1: 64-bit beats {word, cut}
    for (uint32_t x = threadIdx.x; x < (1u << log2n); x += 256) {
      const uint32_t w = row[(size_t)x << word_shift];
      if (w & 1u) {
        const uint32_t slot = ob + atomicAdd(&ovf_cursor, 1u);
        if (slot < cap) {
          rsp_detection d;
          d.frame = of;
          d.bin = x;
          d.doppler = 0;
          d.word = w;
          list[slot] = d;
        }
      }
    }
    __syncthreads();
  }
  if constexpr (!PREFIX) publish_counts(counters, cap, d_count);
}

hipError_t launch_compact_frames(const uint32_t* fcount, const uint2* fdet, uint32_t n_frames,
                                 const uint32_t* words, int log2n, int word_shift, rsp_detection* list,
                                 uint32_t cap, uint32_t* counters, uint32_t* d_count, hipStream_t stream) {
  if (n_frames == 0) return hipMemsetAsync(d_count, 0, 2 * sizeof(uint32_t), stream);
  if (n_frames <= kPrefixFrames)
    hipLaunchKernelGGL(compact_frames_kernel<true>,
                       dim3((n_frames + 255) / 256 + (words ? (n_frames + kOvfFramesPerWg - 1) / kOvfFramesPerWg : 0u)), dim3(256), 0, stream,
                       fcount, fdet, n_frames, words, log2n, word_shift, list, cap, counters, d_count);
  else
    hipLaunchKernelGGL(compact_frames_kernel<false>, dim3((n_frames + 255) / 256), dim3(256), 0, stream,
                       fcount, fdet, n_frames, words, log2n, word_shift, list, cap, counters, d_count);
  return hipGetLastError();
}

// Dense words -> compact list of the peak cells (word bit 0, Tester:165).  One pass: a workgroup
// owns 16 384 consecutive cells, every thread loads its 64 words with four 16-byte loads issued
// together (16 KiB in flight per workgroup: the kernel runs at streaming rate, 4 B per cell), counts
// its peaks, the workgroup reserves a block of the list with ONE global atomic and the (rare) peaks
// are written from registers.
constexpr int kCompactCellsPerWg = 16384;

__global__ void __launch_bounds__(256)
compact_kernel(const uint32_t* __restrict__ words, uint64_t n_cells, uint32_t log2_row,
               uint32_t log2_rows_per_frame, uint32_t word_shift, rsp_detection* __restrict__ list, uint32_t cap,
               uint32_t* __restrict__ counters, uint32_t* __restrict__ d_count) {
  // word_shift = 1: 64-bit beats {word, cut} (sendCut): n_cells counts 32-bit WORDS, every other one is a cut
  const uint32_t odd = word_shift ? 0u : 1u;
  __shared__ uint32_t wave_cnt[4];
  __shared__ uint32_t base_sh;
  const uint64_t lo = (uint64_t)blockIdx.x * kCompactCellsPerWg;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // thread t, load j covers cells lo + 1024 j + 4 t .. + 3 (n_cells is a multiple of 4: whole rows of >= 16 cells)
  u32x4 w[16];
  uint32_t mine = 0;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const uint64_t c = lo + 1024u * j + 4u * threadIdx.x;
    w[j] = c < n_cells ? *reinterpret_cast<const u32x4*>(words + c) : u32x4{0u, 0u, 0u, 0u};
  }
#pragma unroll
  for (int j = 0; j < 16; ++j) mine += (w[j].x & 1u) + (w[j].y & odd) + (w[j].z & 1u) + (w[j].w & odd);
  uint32_t inc = mine;  // inclusive scan over the wave
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t t = __shfl_up(inc, d);
    if (lane >= d) inc += t;
  }
  if (lane == 63) wave_cnt[wave] = inc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t t = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    base_sh = reserve_block(counters, t, t);
  }
  __syncthreads();
  if (mine) {  // rare: the thread's words are read again (L2 hits) by a compact loop instead of keeping 64 registers live
    uint32_t slot = base_sh + inc - mine;
    for (int w0 = 0; w0 < wave; ++w0) slot += wave_cnt[w0];
#pragma unroll 1
    for (int j = 0; j < 16; ++j) {
      const uint64_t c = lo + 1024u * j + 4u * threadIdx.x;
      if (c >= n_cells) break;
      const u32x4 v4 = *reinterpret_cast<const u32x4*>(words + c);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint32_t v = v4[q];
        if ((v & 1u) && ((q & 1) == 0 || odd)) {
          if (slot < cap) {
            const uint64_t i = (c + q) >> word_shift;
            rsp_detection d;
            d.bin = (uint32_t)(i & ((1ull << log2_row) - 1ull));
            const uint64_t row = i >> log2_row;
            d.doppler = (uint32_t)(row & ((1ull << log2_rows_per_frame) - 1ull));
            d.frame = (uint32_t)(row >> log2_rows_per_frame);
            d.word = v;
            list[slot] = d;
          }
          ++slot;
        }
      }
    }
  }
}

// {found, cursor} -> d_count, counters re-zeroed: one thread, its own launch.  The dense compaction runs
// thousands of workgroups; a ticket per workgroup on ONE address (publish_counts) serialises at the
// memory side (~13 ns each) and cost more than this ~2 us launch.
__global__ void compact_finalize_kernel(uint32_t* __restrict__ counters, uint32_t cap, uint32_t* __restrict__ d_count,
                                        bool use_found) {
  const uint32_t found = atomicExch(&counters[0], 0u);
  uint32_t cursor = atomicExch(&counters[1], 0u);
  if (use_found) cursor = found;  // lists appended peak by peak (2-D CFAR kernels): every peak found was offered a slot
  d_count[0] = found;
  d_count[1] = cursor < cap ? cursor : cap;
}

hipError_t launch_compact_finalize(uint32_t* counters, uint32_t cap, uint32_t* d_count, bool stored_is_found,
                                   hipStream_t stream) {
  hipLaunchKernelGGL(compact_finalize_kernel, dim3(1), dim3(1), 0, stream, counters, cap, d_count, stored_is_found);
  return hipGetLastError();
}

hipError_t launch_compact(const uint32_t* words, uint64_t n_cells, uint32_t log2_row,
                          uint32_t log2_rows_per_frame, uint32_t word_shift, rsp_detection* list, uint32_t cap,
                          uint32_t* counters, uint32_t* d_count, hipStream_t stream) {
  if (n_cells == 0) return hipMemsetAsync(d_count, 0, 2 * sizeof(uint32_t), stream);
  const uint64_t n_words = n_cells << word_shift;
  const uint64_t blocks = (n_words + kCompactCellsPerWg - 1) / kCompactCellsPerWg;
  hipLaunchKernelGGL(compact_kernel, dim3((uint32_t)blocks), dim3(256), 0, stream, words, n_words,
                     log2_row, log2_rows_per_frame, word_shift, list, cap, counters, d_count);
  hipLaunchKernelGGL(compact_finalize_kernel, dim3(1), dim3(1), 0, stream, counters, cap, d_count, false);
  return hipGetLastError();
}

}  // namespace rsp
