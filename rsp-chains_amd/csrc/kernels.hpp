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
  const void* twiddles;    // device: F32 per-pass base-twiddle tables (fft_lds.hpp load_tw); FIXED16 W_N^k, k < N/2, packed Q2.14 pairs
  const int16_t* log_lut;  // device: log2 fraction table (FIXED16, mag mode 1)
  // optional fused detection output: per-frame peak count + first kFrameDetCap {bin, word}
  uint32_t* frame_count;   // device: n_frames, or NULL
  uint2* frame_det;        // device: n_frames x kFrameDetCap
  hipStream_t stream;
  int device;                    // HIP ordinal the launch runs on (per-device one-time kernel attributes)
  uint32_t max_frames_per_launch;  // 0 = as many as 32-bit byte offsets allow; tests force the split path
  bool force_generic_tail;         // tests / A-B: per-cell CFAR tail even where the quad tail applies
  bool experiment;                 // RSP_OPT_EXPERIMENT: hand the launch to a side library's kernel (tools/experiments/)
  // optional: events bound to the kernel DISPATCH (hipExtLaunchKernelGGL): its own begin / end timestamps, what
  // rocprofv3's kernel trace reports -- an event pair recorded around the launch also times the dispatch (~2 us)
  hipEvent_t ev_start, ev_stop;
};

hipError_t launch_chain1d(const Chain1dLaunch& a);
hipError_t launch_chain1d_small(const Chain1dLaunch& a);  // 16..128-point frames (small.hip)

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
  const void* win_doppler;  // slow-time window coefficients (float / Q1.15), or NULL; the fast-time one is regs.window
  void* scratch_complex;  // device: n_ch * nd * nr * 8 B
  float* scratch_mag;     // device: n_ch * nd * nr * 4 B
  hipStream_t stream;
  int device;
  bool force_tiled_cfar;  // tests: take the run-time-window kernel even for the compile-time windows
  bool fixed;             // FIXED16 data path: beats in, ROM / Q1.15 windows, int32 magnitudes, integer CFAR
  const int16_t* log_lut; // FIXED16: log2 look-up table of the magnitude block
  size_t chunk_bytes;     // intermediates (12 B/cell) per chunk of channels; 0 = the whole batch at once
  // optional fused detection list: the CFAR kernel appends its peak cells itself (one device-scope
  // atomic per PEAK, none otherwise), so no second pass over the dense words is needed
  rsp_detection* det_list;  // device, or NULL
  uint32_t det_cap;
  uint32_t* det_count;      // device uint32[2]: {found, stored}: zeroed by the range pass, advanced by the CFAR kernels
};
hipError_t launch_rd2d(const Rd2dLaunch& a);
// channels per chunk (>= 1) and the scratch the API has to provide for it
inline uint32_t rd2d_chunk_channels(int log2nr, int log2nd, uint32_t n_ch, size_t chunk_bytes) {
  if (chunk_bytes == 0) return n_ch;
  const size_t per_ch = (size_t)12 << (log2nr + log2nd);
  size_t k = chunk_bytes / per_ch;
  if (k < 1) k = 1;
  return k > n_ch ? n_ch : (uint32_t)k;
}

// Detection compaction.  `counters` = 3 device words owned by the chain handle {found, cursor, ticket},
// zero between launches: the last workgroup to finish publishes d_count[0] = peaks found,
// d_count[1] = entries stored in list (<= cap) and zeroes them again -- no memset node, and the chain
// kernel does not have to touch them.
constexpr int kCompactCounters = 3;
// per-frame slots of the fused path -> list.  A frame with more than kFrameDetCap peaks is re-read from
// its dense words when `words` is given (complete list); without dense words it contributes its first
// kFrameDetCap peaks and d_count[1] < d_count[0] tells.
hipError_t launch_compact_frames(const uint32_t* fcount, const uint2* fdet, uint32_t n_frames,
                                 const uint32_t* words, int log2n, int word_shift, rsp_detection* list,
                                 uint32_t cap, uint32_t* counters, uint32_t* d_count, hipStream_t stream);

hipError_t launch_compact_finalize(uint32_t* counters, uint32_t cap, uint32_t* d_count, bool stored_is_found,
                                   hipStream_t stream);
hipError_t launch_compact(const uint32_t* words, uint64_t n_cells, uint32_t log2_row,
                          uint32_t log2_rows_per_frame, uint32_t word_shift, rsp_detection* list, uint32_t cap,
                          uint32_t* counters, uint32_t* d_count, hipStream_t stream);

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel instantiation, device), not per launch:
// `granted` is a function-local static of the launcher that owns the instantiation.
constexpr int kMaxDevices = 16;
struct LdsGrant {
  uint32_t bytes[kMaxDevices] = {};
};
template <typename K>
inline hipError_t grant_lds(K kernel, size_t lds, int device, LdsGrant& granted) {
  if (lds > 160 * 1024) return hipErrorOutOfMemory;  // more than a CU has: the caller reports the configuration as unsupported
  if (lds <= 48 * 1024) return hipSuccess;  // within the default limit
  if (device < 0 || device >= kMaxDevices) device = kMaxDevices - 1;  // shared slot: still correct, set again
  // benign race between host threads of different handles: both would set the same attribute
  if (granted.bytes[device] >= lds && device != kMaxDevices - 1) return hipSuccess;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e == hipSuccess) granted.bytes[device] = (uint32_t)lds;
  return e;
}

}  // namespace rsp
