// Host-visible launchers of the gfx950 kernels (chain1d.hip, rd2d.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rspchain.h"
#include "chain_regs.hpp"

namespace rsp {

constexpr int kFrameDetCap = 64;  // peaks listed per frame by the fused path

struct Chain1dLaunch {
  const void* in;       // device: n_frames x 2^log2n beats (4 B FIXED16 / 8 B F32)
  uint32_t* out;        // device: n_frames x 2^log2n words
  uint32_t n_frames;
  int log2n;
  bool fixed;
  ChainRegs regs;
  const void* twiddles;    // device: W_N^k, k < N/2 (f32x2, or packed Q2.14 pairs)
  const int16_t* log_lut;  // device: log2 fraction table (FIXED16, mag mode 1)
  // optional fused detection output: per-frame peak count + first kFrameDetCap {bin, word}
  uint32_t* frame_count;   // device: n_frames, or NULL
  uint2* frame_det;        // device: n_frames x kFrameDetCap
  uint32_t* zero_a;        // device words zeroed by block 0 (the compaction launch's counters), or NULL
  uint32_t* zero_b;
  hipStream_t stream;
};

hipError_t launch_chain1d(const Chain1dLaunch& a);
hipError_t launch_chain1d_small(const Chain1dLaunch& a);
// F32 4096-point CA/GO/SO frames, one wave per frame (chain1d_wave.hip)
bool chain1d_wave_supports(const Chain1dLaunch& a);
hipError_t launch_chain1d_wave(const Chain1dLaunch& a);  // 16..128-point frames (small.hip)
size_t chain1d_lds_bytes(int log2n);

// 2-D range-Doppler chain (rd2d.hip): in [n_ch][nd][nr] complex64 -> out [n_ch][nd][nr] words
struct Rd2dLaunch {
  const void* in;
  uint32_t* out;
  uint32_t n_ch;
  int log2nr, log2nd;
  ChainRegs regs;       // mag_mode, scaler, linear, edge; R / G = range training / guard half-widths
  int ref_d, guard_d;   // Doppler training / guard half-widths
  const void* tw_range;
  const void* tw_doppler;
  void* scratch_complex;  // device: n_ch * nd * nr * 8 B
  float* scratch_mag;     // device: n_ch * nd * nr * 4 B
  hipStream_t stream;
};
hipError_t launch_rd2d(const Rd2dLaunch& a);

hipError_t launch_compact_frames(const uint32_t* fcount, const uint2* fdet, uint32_t n_frames,
                                 rsp_detection* list, uint32_t cap, uint32_t* counters,
                                 uint32_t* d_count, hipStream_t stream);

hipError_t launch_compact(const uint32_t* words, uint64_t n_cells, uint32_t log2_row,
                          uint32_t log2_rows_per_frame, rsp_detection* list, uint32_t cap,
                          uint32_t* count, hipStream_t stream);

}  // namespace rsp
