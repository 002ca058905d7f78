/*
 * rsp_oracle.h -- CPU restatement of the sdf-fft -> logMagMux -> CFAR hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under rsp-chains_amd/ may include, link or
 * call this file; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker / the timed CPU baseline.
 *
 * PARITY UNPINNED.  The arithmetic of this path lives in three un-vendored git
 * submodules of the reference (generators/sdf-fft, generators/logMagMux,
 * generators/cfar -- /root/reference/.gitmodules:1-18; directories are empty,
 * commit SHAs unknown), the reference's own tests assert nothing numeric
 * (src/test/scala/FftMagCfarChainTester.scala:244-249) and use an unseeded RNG
 * (src/test/scala/RspChainTesterUtils.scala:59), and no JVM/sbt/verilator exists
 * in this pipeline.  This file therefore restates (a) what IS visible at the
 * reference's call sites -- parameter meaning, wire formats, register map,
 * 1/N FFT scaling, the JPL magnitude formula, run-time defaults -- each cited
 * below, and (b) build-defined choices for everything else (rounding, edge
 * policy, GO/SO combination), each marked BUILD-DEFINED.  It is pinned only by
 * known-answer tests derivable from the reference's files (tests/test_oracle_kat.py).
 */
#ifndef RSP_ORACLE_H
#define RSP_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* trimType (BUILD-DEFINED; upstream default believed Convergent) */
enum { ORC_TRIM_FLOOR = 0, ORC_TRIM_HALF_UP = 1, ORC_TRIM_CONVERGENT = 2 };
/* logMagMux mode register; value 2 = JPL is the only one the reference pins
 * (FftMagCfarChainTester.scala:84); 0/1 are BUILD-DEFINED. */
enum { ORC_MAG_SQR = 0, ORC_MAG_LOG2 = 1, ORC_MAG_JPL = 2 };
/* cfarMode register values: FftMagCfarChainTester.scala:86-92 */
enum { ORC_CFAR_CA = 0, ORC_CFAR_GO = 1, ORC_CFAR_SO = 2, ORC_CFAR_CASH = 3 };
/* frame-edge policy (BUILD-DEFINED): cells outside the frame read as zero
 * (a shift-register window flushed between frames), or the frame is cyclic. */
enum { ORC_EDGE_ZERO = 0, ORC_EDGE_WRAP = 1 };

/* Run-time register state of the chain: one field per CSR of SURVEY App. A.3
 * (FftMagCfarChainTester.scala:82-132) + the elaboration-time fixed-point
 * formats of FftMagCfarChain.scala:78-112 that change the arithmetic. */
typedef struct orc_cfg {
  int32_t log2n;          /* FFT CSR 0: number of active stages (Tester:82)        */
  int32_t trim;           /* FFTParams trimType (BUILD-DEFINED default convergent)  */
  int32_t mag_mode;       /* MAG CSR 0 (Tester:84)                                  */
  int32_t bp_data;        /* FFTParams/MAGParams binPoint (FftMagCfarChain:89,93)   */
  int32_t bp_log;         /* MAGParams binPointLog (FftMagCfarChain:95)             */
  int32_t log2_lut_width; /* MAGParams log2LookUpWidth (FftMagCfarChain:96)         */
  int32_t bp_in;          /* CFARParams protoIn BP (FftMagCfarChain:102)            */
  int32_t bp_thr;         /* CFARParams protoThreshold BP (:103)                    */
  int32_t w_thr;          /* CFARParams protoThreshold width (:103)                 */
  int32_t bp_scaler;      /* CFARParams protoScaler BP (:104)                       */
  uint32_t scaler;        /* CFAR CSR 0x04 raw (Tester:101)                         */
  int32_t linear;         /* CFAR CSR 0x08 logOrLinearMode, 1 = linear (Tester:104) */
  int32_t div_sum;        /* CFAR CSR 0x0C (Tester:107)                             */
  int32_t peak_grouping;  /* CFAR CSR 0x10 (Tester:109)                             */
  int32_t algorithm;      /* CFAR CSR 0x14: 0 = CA family, 1 = GOS (Tester:110-118) */
  int32_t cfar_mode;      /* CFAR CSR 0x18 (Tester:119)                             */
  int32_t ref_window;     /* CFAR CSR 0x1C refWindowSize (Tester:120)               */
  int32_t guard_window;   /* CFAR CSR 0x20 guardWindowSize (Tester:121)             */
  int32_t index_lagg;     /* CFAR CSR 0x24 (Tester:125)                             */
  int32_t index_lead;     /* CFAR CSR 0x28 (Tester:126)                             */
  int32_t sub_window;     /* CFAR CSR 0x2C (Tester:131); CASH only                  */
  int32_t edge;           /* BUILD-DEFINED frame-edge policy                        */
  /* elaboration options every reference configuration leaves at the value 0 here   */
  uint32_t keep_lsb_mask; /* bit s: FFTParams keepMSBorLSB(s) = false (FftMagCfarChain:87) */
  uint32_t expand_mask;   /* bit s: FFTParams expandLogic(s) = 1 (FftMagCfarChain:86)      */
  int32_t no_bit_reverse; /* 1: FFTParams useBitReverse = false (FftMagCfarChain:82)       */
  int32_t send_cut;       /* 1: CFARParams sendCut = true (FftMagCfarChain:107)            */
  int32_t window;         /* pre-FFT window (no reference item): ORC_WIN_*   */
} orc_cfg;

/* pre-FFT window functions (build extension); coefficients Q1.15 / double */
enum { ORC_WIN_NONE = 0, ORC_WIN_HANN = 1, ORC_WIN_HAMMING = 2, ORC_WIN_BLACKMAN = 3 };
double orc_window_coeff(int window, int i, int n);

/* ---- wire formats --------------------------------------------------------- */
/* RspChainTesterUtils.scala:105-109: data[31:16] = re, data[15:0] = im, int16. */
uint32_t orc_pack_iq(int32_t re, int32_t im);
void orc_unpack_iq(uint32_t beat, int16_t* re, int16_t* im);
/* FftMagCfarChainTester.scala:163-167: thr = w >> (log2N+1), peak = w & 1, the
 * log2N bits between are the bin index. */
uint32_t orc_pack_out(uint32_t thr, uint32_t bin, uint32_t peak, int log2n);

/* ---- fixed-point path (16-bit FixedPoint, FftMagCfarChain.scala:78-112) ---- */
/* Twiddle ROM: W_N^k, k < N/2, Q2.14 (twiddleWidth 16, BP = width-2). */
void orc_twiddles_q14(int log2n, int16_t* wr, int16_t* wi);
/* N-point forward FFT, radix-2 DIF butterfly graph of an SDF pipeline, one
 * 1-bit trim per stage (expandLogic = 0, keepMSBorLSB = true,
 * FftMagCfarChain.scala:86-87) => net 1/N (Tester:77), natural-order output
 * (useBitReverse = true, FftMagCfarChain.scala:82). */
void orc_fft_fixed(const int16_t* re_in, const int16_t* im_in, int log2n, int trim,
                   int16_t* re_out, int16_t* im_out);
/* The same with the per-stage options of FFTParams.fixed (FftMagCfarChain.scala:82,86-87), all
 * BUILD-DEFINED (docs/FIXED_POINT_SPEC.md section 3):
 *   expandLogic(s) = 1   stage s keeps its (w+1)-bit results: the word grows by one bit, nothing is
 *                        rounded away at that stage;
 *   keepMSBorLSB(s) = false (expandLogic(s) = 0)  the (w+1)-bit result is cut back to w bits by dropping
 *                        its MSB: no halving at that stage, overflow wraps;
 *   the 2 x 16-bit stream to the magnitude block (beatBytes = 4) carries the 16 most significant bits
 *   of a grown word (floor shift by the number of expanding stages);
 *   useBitReverse = false: the frame leaves in bit-reversed order (position p holds bin bitrev(p)). */
void orc_fft_fixed_ex(const int16_t* re_in, const int16_t* im_in, int log2n, int trim,
                      uint32_t keep_lsb_mask, uint32_t expand_mask, int no_bit_reverse,
                      int16_t* re_out, int16_t* im_out);
int32_t orc_mag_fixed(int16_t re, int16_t im, const orc_cfg* c);
/* One frame of magnitudes -> N output words. */
void orc_cfar_fixed(const int32_t* mag, const orc_cfg* c, uint32_t* out_words,
                    int32_t* thr_out /* may be NULL */);
/* Whole chain, n_frames frames of 2^log2n beats each.  With c->send_cut the output beat is 64 bits,
 * two words per cell: {word as above, cut} (BUILD-DEFINED packing of the wider CFAR beat). */
void orc_chain_fixed(const uint32_t* in_beats, size_t n_frames, const orc_cfg* c,
                     uint32_t* out_words);

/* ---- floating-point path (BASELINE.json configs 2-5; float64 arithmetic) ---- */
typedef struct orc_fcfg {
  int32_t log2n;
  int32_t mag_mode;
  double scaler;          /* thresholdScaler as a real number (Tester:101 / 2^BP) */
  int32_t linear;
  int32_t div_sum;
  int32_t peak_grouping;
  int32_t algorithm;
  int32_t cfar_mode;
  int32_t ref_window;
  int32_t guard_window;
  int32_t index_lagg;
  int32_t index_lead;
  int32_t edge;
  int32_t sub_window; /* CASH only */
  int32_t no_bit_reverse;
  int32_t window;
} orc_fcfg;

/* Forward FFT with the same net 1/N scaling, in float64. interleaved re,im. */
void orc_fft_f64(const double* in, int log2n, double* out);
double orc_mag_f64(double re, double im, int mode);
/* thr[] and peak[] per cell; margin[] = |cut - thr| so that a test can skip
 * cells whose decision is within rounding of the fp32 device path. */
void orc_cfar_f64(const double* mag, const orc_fcfg* c, double* thr, uint8_t* peak,
                  double* margin);
/* fp32 complex in (interleaved), float64 arithmetic.  n_threads > 1 uses OpenMP
 * over frames (used only by bench.py's cpu_baseline leg). */
void orc_chain_f32in(const float* in, size_t n_frames, const orc_fcfg* c, double* thr,
                     uint8_t* peak, double* margin, double* mag_out /* may be NULL */,
                     int n_threads);

/* 2-D range-Doppler (BASELINE.json configs 3/5; no reference counterpart,
 * SURVEY F5): in[ch][doppler d][range r] complex fp32; range FFT over r
 * (log2nr), Doppler FFT over d (log2nd), each scaled 1/N; magnitude; 2-D
 * CA-CFAR with training band (ref_r, ref_d) beyond guard band (guard_r,
 * guard_d), cyclic in Doppler, edge policy `edge` in range.  Output maps are
 * [ch][d][r].  GO / SO (BUILD-DEFINED, spec section 6): the training region is split into a LAGGING half (cells at
 * smaller range than the CUT, + the cells of the CUT's own column at smaller Doppler) and the mirror-image LEADING
 * half; statistic per half = its sum / (count / 2); CA = their mean (= sum / count), GO = max, SO = min. */
typedef struct orc_rdcfg {
  int32_t log2nr, log2nd;
  int32_t mag_mode;
  double scaler;
  int32_t ref_r, ref_d, guard_r, guard_d;
  int32_t edge;
  int32_t window_r, window_d; /* ORC_WIN_*: windows over fast / slow time (build extension) */
  int32_t cfar_mode;          /* ORC_CFAR_CA / _GO / _SO over the lagging / leading halves of the training region */
} orc_rdcfg;
void orc_rd_f32in(const float* in, size_t n_ch, const orc_rdcfg* c, double* thr, uint8_t* peak,
                  double* margin, double* mag_out /* may be NULL */, int n_threads);

/* The 2-D chain on the FIXED16 data path (see rsp_oracle.c): c carries the 1-D register file (log2n = log2nr,
 * ref_window / guard_window = range half-widths, window = range window); out_words n_ch x nd x nr;
 * mag_out (n_ch x nd x nr int32) may be NULL. */
void orc_rd_fixed(const uint32_t* in_beats, size_t n_ch, const orc_cfg* c, int log2nd, int ref_d, int guard_d,
                  int window_d, uint32_t* out_words, int32_t* mag_out, int n_threads);

/* ---- PLFG -> NCO stimulus of the full chain (RspChain.scala:41-42,57-58) -------------------
 * BUILD-DEFINED model (generators/plfg and generators/nco are empty submodules), anchored on
 * FixedPLFGParams / FixedNCOParams (RspChain.scala:84-106), the tester's register program
 * (RspChainVanillaTester.scala:80-94), its statement "peak is expected on frequency bin
 * startingPoint * numOfPoints / (4 * tableSize)" (:85) and the float model calcExpectedNcoOut
 * (RspChainTesterUtils.scala:174-181: amplitude 2^14, sample index 1..N, re = cos, im = sin).
 *   PLFG: chirp i of the frame program = chirp type ordinal[i], repeated repeated[i] times; a chirp
 *   type o is segment_nums[o] segments, RAM row o * max_segments + s; a segment word is
 *   {length[31:24], slope[23:8], -, slope sign[1], reset-to-start[0]}; the value starts at
 *   start_value at every chirp start, each sample outputs it and then adds +-slope.  The stream
 *   repeats its program for as long as it is enabled (enable = 0: output 0).
 *   NCO: phase accumulator of phase_width bits, incremented by the PLFG value BEFORE each output
 *   (sample index starts at 1); quarter-wave table of table_size entries,
 *   table[k] = floor(sin(2 pi k / 4 table_size) * 2^(table_width-2) + 1/2) (RoundHalfUp). */
typedef struct orc_stim_cfg {
  int32_t enable, start_value, num_chirps;
  int32_t max_segments;
  int32_t segment_nums[8], repeated[8], ordinal[8];
  uint32_t ram[64];
  int32_t table_size, table_width, phase_width;
} orc_stim_cfg;
void orc_plfg(const orc_stim_cfg* c, size_t n, int32_t* values);
void orc_plfg_nco(const orc_stim_cfg* c, size_t n, uint32_t* beats);

#ifdef __cplusplus
}
#endif
#endif
