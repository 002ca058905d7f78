// Kernel-visible snapshot of the chain's register file + elaboration constants.
// Host fills it from rsp_chain_params and the CSR writes (include/rspchain.h);
// passed to kernels by value (lives in SGPRs / kernarg).
#pragma once
#include <stdint.h>

namespace rsp {

struct ChainRegs {
  // FFT
  int32_t trim_bias1, trim_bias15;  // 0 (floor) or 2^(n-1) (half-up, convergent)
  int32_t trim_conv;                // 1 = convergent tie fix-up
  // magnitude (MAG CSR 0 + MAGParams)
  int32_t mag_mode, bp_data, bp_log, lut_w;
  // CFAR (CSR 0x04..0x2C + CFARParams protos)
  int32_t bp_in, bp_thr, w_thr, bp_scaler;
  uint32_t scaler_raw;
  float scaler_f;   // F32 path: scaler_raw / 2^bp_scaler
  float div_f;      // F32 path: 2^-divSum
  int32_t linear, div_sum, peak_grouping, algorithm, cfar_mode;
  int32_t R, G, idx_lagg, idx_lead, sub_window, edge;
  // FIXED16 threshold arithmetic, pre-split on the host so the kernel needs no sign tests:
  // linear: thr = ((stat * scaler) << lin_shl) >> lin_shr   (one of the two is 0)
  // log:    thr = ((stat << log_shl) >> log_shr) + log_scaler
  int32_t lin_shl, lin_shr, log_shl, log_shr, log_scaler, tmax, tmin;
  // 1: every statistic x scaler product of this configuration fits 31 bits (and both factors 24): the linear threshold
  // is one v_mul_i32_i24 instead of a 64-bit multiply -- every reference configuration; 0: the general 64-bit form
  int32_t fast32;
  // elaboration options every reference configuration leaves at their defaults (all 0 here)
  uint32_t keep_lsb_mask;  // bit s: FFTParams keepMSBorLSB(s) = false -- stage s drops its MSB instead of its LSB
  uint32_t expand_mask;    // bit s: FFTParams expandLogic(s) = 1 -- stage s keeps its (w+1)-bit results
  int32_t growth;          // popcount of expand_mask over the active stages: bits shed at the FFT output
  int32_t rev_order;       // 1: FFTParams useBitReverse = false -- stream position p carries bin bitrev(p)
  int32_t send_cut;        // 1: CFARParams sendCut = true -- 64-bit output beat {word, cut}
  const void* window;      // pre-FFT window coefficients (float / Q1.15 int16 per sample), or NULL
};

constexpr int kMinLog2N = 8;       // LDS-tiled kernels: scan rows are 16 lanes x 16 cells = 256 cells
constexpr int kMinLog2NSmall = 4;  // one-thread-per-frame kernel (small.hip): 16..128 points
constexpr int kMaxLog2N = 14;    // 16384 points: one 1024-thread workgroup per CU, 137 KiB of LDS (8192: 68 KiB per frame)
constexpr int kMaxLog2N2d = 13;  // 2-D chain: range FFT up to 8192 points (tiled intermediate maps)
constexpr int kMaxRef = 128;   // refWindow + guardWindow <= 256 (LDS prefix halo)

}  // namespace rsp
