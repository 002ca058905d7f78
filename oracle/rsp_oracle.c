/*
 * rsp_oracle.c -- CPU restatement of sdf-fft -> logMagMux -> CFAR.  See rsp_oracle.h:
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (reference arithmetic is in empty,
 * un-vendored submodules; its tests pin no numbers).
 *
 * Citations are relative to /root/reference/.  "Tester" =
 * src/test/scala/FftMagCfarChainTester.scala, "Utils" =
 * src/test/scala/RspChainTesterUtils.scala, "Chain" =
 * src/main/scala/FftMagCfarChain.scala, "RT" = src/test/scala/RspChainVanillaTester.scala.
 */
#include "rsp_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ------------------------------------------------------------------ wire formats */

/* Utils:105-109 (formAXI4StreamComplexData): two 16-digit two's-complement
 * strings concatenated, real first. */
uint32_t orc_pack_iq(int32_t re, int32_t im) {
  return ((uint32_t)(re & 0xFFFF) << 16) | (uint32_t)(im & 0xFFFF);
}

void orc_unpack_iq(uint32_t beat, int16_t* re, int16_t* im) {
  *re = (int16_t)(beat >> 16);
  *im = (int16_t)(beat & 0xFFFF);
}

/* Tester:153,163-167: threshold above bit log2N+1, bin in the middle, peak = bit 0. */
uint32_t orc_pack_out(uint32_t thr, uint32_t bin, uint32_t peak, int log2n) {
  return (thr << (log2n + 1)) | ((bin & ((1u << log2n) - 1u)) << 1) | (peak & 1u);
}

/* ------------------------------------------------------------------ fixed-point FFT */

/* Arithmetic right shift by n with the FFT's trim type.  BUILD-DEFINED: the
 * reference passes no trimType at Chain:78-90, so upstream's default applies
 * (believed Convergent, unverifiable). */
static int64_t trim_shift(int64_t x, int n, int trim) {
  if (n <= 0) return x << (-n);
  switch (trim) {
    case ORC_TRIM_FLOOR:
      return x >> n;
    case ORC_TRIM_HALF_UP:
      return (x + ((int64_t)1 << (n - 1))) >> n;
    default: { /* convergent: ties to even */
      int64_t t = x + ((int64_t)1 << (n - 1));
      int64_t r = t >> n;
      if ((t & (((int64_t)1 << n) - 1)) == 0) r &= ~(int64_t)1;
      return r;
    }
  }
}


/* twiddleWidth = 16 (Chain:80), BP = 14: W_N^k = exp(-2 pi i k / N), rounded to
 * nearest (BUILD-DEFINED rounding of the ROM contents). */
void orc_twiddles_q14(int log2n, int16_t* wr, int16_t* wi) {
  int n = 1 << log2n;
  for (int k = 0; k < n / 2; k++) {
    double a = -2.0 * M_PI * (double)k / (double)n;
    wr[k] = (int16_t)lround(cos(a) * 16384.0);
    wi[k] = (int16_t)lround(sin(a) * 16384.0);
  }
}

static unsigned bitrev(unsigned x, int bits) {
  unsigned r = 0;
  for (int i = 0; i < bits; i++) {
    r = (r << 1) | (x & 1u);
    x >>= 1;
  }
  return r;
}

/*
 * The SDF pipeline computes, stage by stage, the radix-2 decimation-in-frequency
 * butterfly graph: stage s pairs samples half = N >> (s+1) apart (the depth of
 * that stage's delay-feedback line), forwards a+b, and feeds (a-b) * W to the
 * next stage.  With expandLogic = 0 and keepMSBorLSB = true (Chain:86-87) each
 * stage's (w+1)-bit result is cut back to w bits by dropping its LSB, so every
 * stage halves and the net gain is 1/N (Tester:77 divides the float model by
 * fftSize).  BUILD-DEFINED: one rounding per output -- (a+b) is trimmed by 1
 * bit, (a-b)*W by 15 bits (14 twiddle fraction bits + the dropped LSB).
 * The bit-reversed result is reordered (useBitReverse = true, Chain:82).
 */
static int64_t wrap_bits(int64_t x, int bits) { /* two's-complement wrap to `bits` bits */
  uint64_t m = ((uint64_t)1 << bits) - 1u, v = (uint64_t)x & m;
  return (v >> (bits - 1)) ? (int64_t)(v | ~m) : (int64_t)v;
}

void orc_fft_fixed_ex(const int16_t* re_in, const int16_t* im_in, int log2n, int trim,
                      uint32_t keep_lsb_mask, uint32_t expand_mask, int no_bit_reverse,
                      int16_t* re_out, int16_t* im_out) {
  int n = 1 << log2n;
  int64_t* xr = (int64_t*)malloc(sizeof(int64_t) * (size_t)n);
  int64_t* xi = (int64_t*)malloc(sizeof(int64_t) * (size_t)n);
  int16_t* wr = (int16_t*)malloc(sizeof(int16_t) * (size_t)(n / 2 + 1));
  int16_t* wi = (int16_t*)malloc(sizeof(int16_t) * (size_t)(n / 2 + 1));
  for (int i = 0; i < n; i++) {
    xr[i] = re_in[i];
    xi[i] = im_in[i];
  }
  orc_twiddles_q14(log2n, wr, wi);
  int w = 16, growth = 0; /* dataWidth = 16 (Chain:79) + one bit per expanding stage so far */
  for (int s = 0; s < log2n; s++) {
    int half = n >> (s + 1);
    int grow = (int)((expand_mask >> s) & 1u);
    int lsb = !grow && ((keep_lsb_mask >> s) & 1u);
    /* result formats: grow -> w+1 bits, nothing trimmed; keep MSB -> trim 1 bit, w bits;
     * keep LSB -> no trim, wrapped to w bits.  The product carries 14 twiddle fraction bits more. */
    int sh_sum = (grow || lsb) ? 0 : 1, sh_prod = 14 + sh_sum, wout = w + grow;
    for (int base = 0; base < n; base += 2 * half) {
      for (int j = 0; j < half; j++) {
        int a = base + j, b = a + half;
        int k = j << s; /* W_{2*half}^j == W_N^(j * 2^s) */
        int64_t sr = xr[a] + xr[b], si = xi[a] + xi[b];
        int64_t dr = xr[a] - xr[b], di = xi[a] - xi[b];
        int64_t pr = dr * wr[k] - di * wi[k];
        int64_t pi = dr * wi[k] + di * wr[k];
        xr[a] = wrap_bits(trim_shift(sr, sh_sum, trim), wout);
        xi[a] = wrap_bits(trim_shift(si, sh_sum, trim), wout);
        xr[b] = wrap_bits(trim_shift(pr, sh_prod, trim), wout);
        xi[b] = wrap_bits(trim_shift(pi, sh_prod, trim), wout);
      }
    }
    w = wout;
    growth += grow;
  }
  for (int k = 0; k < n; k++) {
    unsigned p = no_bit_reverse ? (unsigned)k : bitrev((unsigned)k, log2n);
    /* position k of the output stream holds element p of the in-place result = bin bitrev(p) */
    re_out[k] = (int16_t)(xr[p] >> growth);
    im_out[k] = (int16_t)(xi[p] >> growth);
  }
  free(xr);
  free(xi);
  free(wr);
  free(wi);
}

void orc_fft_fixed(const int16_t* re_in, const int16_t* im_in, int log2n, int trim,
                   int16_t* re_out, int16_t* im_out) {
  orc_fft_fixed_ex(re_in, im_in, log2n, trim, 0u, 0u, 0, re_out, im_out);
}

/* pre-FFT window (no reference item): w[i], i < n, symmetric ("periodic = false") */
double orc_window_coeff(int window, int i, int n) {
  double x = 2.0 * M_PI * (double)i / (double)(n - 1);
  switch (window) {
    case ORC_WIN_HANN: return 0.5 - 0.5 * cos(x);
    case ORC_WIN_HAMMING: return 0.54 - 0.46 * cos(x);
    case ORC_WIN_BLACKMAN: return 0.42 - 0.5 * cos(x) + 0.08 * cos(2.0 * x);
    default: return 1.0;
  }
}

/* ------------------------------------------------------------------ fixed-point magnitude */

static int32_t jpl_fixed(int16_t re, int16_t im) {
  /* Utils:120-127: u = max(|re|,|im|), v = min(...), max(u + v/8, 7u/8 + v/2).
   * BUILD-DEFINED: each division is a floor shift; result saturates at the
   * signed 16-bit maximum of MAGParams dataWidth (Chain:92). */
  int32_t ar = re < 0 ? -(int32_t)re : re, ai = im < 0 ? -(int32_t)im : im;
  int32_t u = ar > ai ? ar : ai, v = ar > ai ? ai : ar;
  int32_t t1 = u + (v >> 3);
  int32_t t2 = ((7 * u) >> 3) + (v >> 1);
  int32_t m = t1 > t2 ? t1 : t2;
  return m > 32767 ? 32767 : m;
}

int32_t orc_mag_fixed(int16_t re, int16_t im, const orc_cfg* c) {
  switch (c->mag_mode) {
    case ORC_MAG_SQR: {
      /* Utils:205-208 "sqrMag": re^2 + im^2, kept at binPoint (BUILD-DEFINED:
       * floor shift by binPoint, saturate to 16-bit signed). */
      int64_t s = (int64_t)re * re + (int64_t)im * im;
      s >>= c->bp_data;
      return (int32_t)(s > 32767 ? 32767 : s);
    }
    case ORC_MAG_LOG2: {
      /* Utils:209-212 "log2Mag" = log2(jplMag).  BUILD-DEFINED realisation:
       * leading-one position + 2^log2LookUpWidth-entry fraction table, output
       * Q(dataWidthLog - binPointLog).binPointLog (Chain:94-96). */
      int32_t x = jpl_fixed(re, im);
      if (x < 1) x = 1;
      int e = 31 - __builtin_clz((unsigned)x);
      int lw = c->log2_lut_width;
      uint32_t f = e >= lw ? ((uint32_t)x >> (e - lw)) : ((uint32_t)x << (lw - e));
      f &= (1u << lw) - 1u;
      int32_t lut = (int32_t)llround(log2(1.0 + (double)f / (double)(1u << lw)) *
                                     (double)(1 << c->bp_log));
      return ((e - c->bp_data) * (1 << c->bp_log)) + lut;
    }
    default:
      return jpl_fixed(re, im);
  }
}

/* ------------------------------------------------------------------ CFAR helpers */

static int cmp_i32(const void* a, const void* b) {
  int32_t x = *(const int32_t*)a, y = *(const int32_t*)b;
  return (x > y) - (x < y);
}
static int cmp_f64(const void* a, const void* b) {
  double x = *(const double*)a, y = *(const double*)b;
  return (x > y) - (x < y);
}

/* index of cell (k + off) under the edge policy; -1 = outside (reads as 0) */
static int cell_index(int k, int off, int n, int edge) {
  int j = k + off;
  if (edge == ORC_EDGE_WRAP) return ((j % n) + n) % n;
  return (j < 0 || j >= n) ? -1 : j;
}

/* ------------------------------------------------------------------ fixed-point CFAR */

/*
 * 1-D sliding-window CFAR over one frame.  Window geometry (Chain:105-106,
 * Tester:120-121): [R lagging cells][G guard][CUT][G guard][R leading cells].
 * Registers: SURVEY App. A.3 / Tester:100-132.
 * BUILD-DEFINED (upstream cfar generator unavailable):
 *   - per-side statistic = window sum >> divSum (CA family) or the
 *     index-th smallest cell of the window (GOS, Tester:123-127);
 *   - cfarMode 0 (Cell Averaging) = (lagg + lead) >> 1, 1 = max, 2 = min,
 *     3 (CASH) = min over sides of the largest sub-window sum, >> divSum;
 *   - linear: threshold = (statistic * scaler) >> (BP_in + BP_scaler - BP_thr);
 *     log: threshold = statistic + scaler (both aligned to BP_thr);
 *     saturated to the signed protoThreshold width (Chain:103);
 *   - peak = CUT > threshold (binary points aligned), with peakGrouping also
 *     CUT > both neighbours;
 *   - cells outside the frame read 0 (edge 0) or wrap (edge 1).
 * Output word: Tester:163-167.
 */
/* statistic -> threshold (spec section 5: alignment by floor shifts, saturation to protoThreshold's range) */
static int64_t threshold_fixed(int64_t stat, const orc_cfg* c) {
  int64_t tmax = ((int64_t)1 << (c->w_thr - 1)) - 1, tmin = -((int64_t)1 << (c->w_thr - 1));
  int64_t thr;
  if (c->linear) {
    thr = trim_shift(stat * (int64_t)c->scaler, c->bp_in + c->bp_scaler - c->bp_thr, ORC_TRIM_FLOOR);
  } else {
    thr = trim_shift(stat, c->bp_in - c->bp_thr, ORC_TRIM_FLOOR) +
          trim_shift((int64_t)c->scaler, c->bp_scaler - c->bp_thr, ORC_TRIM_FLOOR);
  }
  if (thr > tmax) thr = tmax;
  if (thr < tmin) thr = tmin;
  return thr;
}

void orc_cfar_fixed(const int32_t* mag, const orc_cfg* c, uint32_t* out_words, int32_t* thr_out) {
  int n = 1 << c->log2n, R = c->ref_window, G = c->guard_window;
  int32_t* win = (int32_t*)malloc(sizeof(int32_t) * (size_t)(R > 0 ? R : 1));
  for (int k = 0; k < n; k++) {
    int64_t stat_side[2];
    for (int side = 0; side < 2; side++) { /* 0 = lagging (k-d), 1 = leading (k+d) */
      int64_t sum = 0;
      for (int d = 0; d < R; d++) {
        /* window position d = 0 is the oldest lagging cell / nearest leading cell;
         * ordering only matters for CASH sub-windows */
        int off = side == 0 ? -(G + R) + d : (G + 1) + d;
        int j = cell_index(k, off, n, c->edge);
        win[d] = j < 0 ? 0 : mag[j];
        sum += win[d];
      }
      if (c->algorithm == 1) { /* GOS */
        int idx = side == 0 ? c->index_lagg : c->index_lead;
        qsort(win, (size_t)R, sizeof(int32_t), cmp_i32);
        stat_side[side] = win[idx];
      } else if (c->cfar_mode == ORC_CFAR_CASH) {
        int sw = c->sub_window > 0 ? c->sub_window : R;
        int64_t best = INT64_MIN;
        for (int s0 = 0; s0 + sw <= R; s0 += sw) {
          int64_t ss = 0;
          for (int d = 0; d < sw; d++) ss += win[s0 + d];
          if (ss > best) best = ss;
        }
        stat_side[side] = best >> c->div_sum;
      } else {
        stat_side[side] = sum >> c->div_sum;
      }
    }
    int64_t a = stat_side[0], b = stat_side[1], stat;
    switch (c->cfar_mode) {
      case ORC_CFAR_CA:
        stat = (a + b) >> 1;
        break;
      case ORC_CFAR_GO:
        stat = a > b ? a : b;
        break;
      default: /* SO and CASH */
        stat = a < b ? a : b;
        break;
    }
    int64_t thr = threshold_fixed(stat, c);
    int64_t cut = mag[k];
    int peak = cut * ((int64_t)1 << c->bp_thr) > thr * ((int64_t)1 << c->bp_in);
    if (c->peak_grouping) {
      int jl = cell_index(k, -1, n, c->edge), jr = cell_index(k, 1, n, c->edge);
      int64_t l = jl < 0 ? 0 : mag[jl], r = jr < 0 ? 0 : mag[jr];
      peak = peak && cut > l && cut > r;
    }
    if (thr_out) thr_out[k] = (int32_t)thr;
    out_words[k] = orc_pack_out((uint32_t)(int32_t)thr, (uint32_t)k, (uint32_t)peak, c->log2n);
  }
  free(win);
}

/* Tester:137 streams fftSize beats with TLAST on the final one; Tester:145-151
 * collects exactly fftSize output words per frame. */
void orc_chain_fixed(const uint32_t* in_beats, size_t n_frames, const orc_cfg* c,
                     uint32_t* out_words) {
  int n = 1 << c->log2n;
  int16_t* re = (int16_t*)malloc(sizeof(int16_t) * (size_t)n * 4);
  int16_t *im = re + n, *fr = re + 2 * n, *fi = re + 3 * n;
  int32_t* mag = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
  uint32_t* words = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)n);
  for (size_t f = 0; f < n_frames; f++) {
    for (int i = 0; i < n; i++) {
      orc_unpack_iq(in_beats[f * (size_t)n + (size_t)i], &re[i], &im[i]);
      if (c->window != ORC_WIN_NONE) {
        /* BUILD-DEFINED: coefficient rounded to Q1.15 (32767 = 1 - 2^-15), product rounded half-up to 16 bits */
        int32_t wq = (int32_t)lround(orc_window_coeff(c->window, i, n) * 32767.0);
        re[i] = (int16_t)(((int32_t)re[i] * wq + (1 << 14)) >> 15);
        im[i] = (int16_t)(((int32_t)im[i] * wq + (1 << 14)) >> 15);
      }
    }
    orc_fft_fixed_ex(re, im, c->log2n, c->trim, c->keep_lsb_mask, c->expand_mask, c->no_bit_reverse, fr, fi);
    for (int i = 0; i < n; i++) mag[i] = orc_mag_fixed(fr[i], fi[i], c);
    if (!c->send_cut) {
      orc_cfar_fixed(mag, c, out_words + f * (size_t)n, NULL);
    } else { /* 64-bit beat: low word = the sendCut = false word, high word = the cell under test */
      orc_cfar_fixed(mag, c, words, NULL);
      for (int i = 0; i < n; i++) {
        out_words[2 * (f * (size_t)n + (size_t)i)] = words[i];
        out_words[2 * (f * (size_t)n + (size_t)i) + 1] = (uint32_t)mag[i];
      }
    }
  }
  free(re);
  free(mag);
  free(words);
}

/* ------------------------------------------------------------------ float64 path */

/* Iterative radix-2 DIT in float64, scaled by 1/N (Tester:77). */
static void fft_f64_strided(const double* in, size_t istride, int log2n, const double* tw,
                            double* out /* interleaved, contiguous */) {
  int n = 1 << log2n;
  for (int i = 0; i < n; i++) {
    unsigned p = bitrev((unsigned)i, log2n);
    out[2 * p] = in[2 * istride * (size_t)i];
    out[2 * p + 1] = in[2 * istride * (size_t)i + 1];
  }
  for (int s = 1; s <= log2n; s++) {
    int m = 1 << s, half = m >> 1, step = n >> s;
    for (int base = 0; base < n; base += m) {
      for (int j = 0; j < half; j++) {
        double wr = tw[2 * (j * step)], wi = tw[2 * (j * step) + 1];
        double* a = out + 2 * (base + j);
        double* b = out + 2 * (base + j + half);
        double tr = b[0] * wr - b[1] * wi, ti = b[0] * wi + b[1] * wr;
        b[0] = a[0] - tr;
        b[1] = a[1] - ti;
        a[0] += tr;
        a[1] += ti;
      }
    }
  }
  double sc = 1.0 / (double)n;
  for (int i = 0; i < 2 * n; i++) out[i] *= sc;
}

static double* make_tw_f64(int log2n) {
  int n = 1 << log2n;
  double* tw = (double*)malloc(sizeof(double) * 2 * (size_t)(n / 2 + 1));
  for (int k = 0; k < n / 2 + 1; k++) {
    double a = -2.0 * M_PI * (double)k / (double)n;
    tw[2 * k] = cos(a);
    tw[2 * k + 1] = sin(a);
  }
  return tw;
}

void orc_fft_f64(const double* in, int log2n, double* out) {
  double* tw = make_tw_f64(log2n);
  fft_f64_strided(in, 1, log2n, tw, out);
  free(tw);
}

double orc_mag_f64(double re, double im, int mode) {
  double ar = fabs(re), ai = fabs(im);
  double u = ar > ai ? ar : ai, v = ar > ai ? ai : ar;
  double t1 = u + v / 8.0, t2 = 7.0 * u / 8.0 + v / 2.0; /* Utils:123-125 */
  double jpl = t1 > t2 ? t1 : t2;
  switch (mode) {
    case ORC_MAG_SQR:
      return re * re + im * im; /* Utils:206 */
    case ORC_MAG_LOG2:
      return log2(jpl > (double)FLT_MIN ? jpl : (double)FLT_MIN); /* Utils:210 */
    default:
      return jpl;
  }
}

static double dmin(double a, double b) { return a < b ? a : b; }

void orc_cfar_f64(const double* mag, const orc_fcfg* c, double* thr, uint8_t* peak,
                  double* margin) {
  int n = 1 << c->log2n, R = c->ref_window, G = c->guard_window;
  double* win = (double*)malloc(sizeof(double) * (size_t)(R > 0 ? R : 1));
  double div = ldexp(1.0, -c->div_sum);
  for (int k = 0; k < n; k++) {
    double side_stat[2];
    for (int side = 0; side < 2; side++) {
      double sum = 0.0;
      for (int d = 0; d < R; d++) {
        int off = side == 0 ? -(G + R) + d : (G + 1) + d;
        int j = cell_index(k, off, n, c->edge);
        win[d] = j < 0 ? 0.0 : mag[j];
        sum += win[d];
      }
      if (c->algorithm == 1) {
        int idx = side == 0 ? c->index_lagg : c->index_lead;
        qsort(win, (size_t)R, sizeof(double), cmp_f64);
        side_stat[side] = win[idx];
      } else if (c->cfar_mode == ORC_CFAR_CASH) { /* same rule as the fixed-point path */
        int sw = c->sub_window > 0 ? c->sub_window : R;
        double best = 0.0;
        int first = 1;
        for (int s0 = 0; s0 + sw <= R; s0 += sw) {
          double ss = 0.0;
          for (int d = 0; d < sw; d++) ss += win[s0 + d];
          if (first || ss > best) best = ss;
          first = 0;
        }
        side_stat[side] = best * div;
      } else {
        side_stat[side] = sum * div;
      }
    }
    double a = side_stat[0], b = side_stat[1], stat;
    switch (c->cfar_mode) {
      case ORC_CFAR_CA:
        stat = 0.5 * (a + b);
        break;
      case ORC_CFAR_GO:
        stat = a > b ? a : b;
        break;
      default:
        stat = a < b ? a : b;
        break;
    }
    double t = c->linear ? stat * c->scaler : stat + c->scaler;
    double cut = mag[k];
    int p = cut > t;
    /* margin = how far the inputs of the decision may move before the flag can flip.  The flag
     * is an AND of comparisons: a set flag flips when its weakest comparison flips; a clear flag
     * stays clear as long as ANY failing comparison keeps failing, so its margin is the largest
     * margin among the failing ones (a neighbour cannot turn a cell with cut <= thr into a peak). */
    double mg = fabs(cut - t);
    if (c->peak_grouping) {
      int jl = cell_index(k, -1, n, c->edge), jr = cell_index(k, 1, n, c->edge);
      double l = jl < 0 ? 0.0 : mag[jl], r = jr < 0 ? 0.0 : mag[jr];
      double d[3] = {cut - t, cut - l, cut - r};
      p = d[0] > 0.0 && d[1] > 0.0 && d[2] > 0.0;
      if (p) {
        mg = dmin(d[0], dmin(d[1], d[2]));
      } else {
        mg = 0.0;
        for (int q = 0; q < 3; q++)
          if (!(d[q] > 0.0) && -d[q] > mg) mg = -d[q];
      }
    }
    thr[k] = t;
    peak[k] = (uint8_t)p;
    if (margin) margin[k] = mg;
  }
  free(win);
}

void orc_chain_f32in(const float* in, size_t n_frames, const orc_fcfg* c, double* thr,
                     uint8_t* peak, double* margin, double* mag_out, int n_threads) {
  int n = 1 << c->log2n;
  double* tw = make_tw_f64(c->log2n);
#ifdef _OPENMP
#pragma omp parallel num_threads(n_threads > 0 ? n_threads : 1)
#endif
  {
    double* x = (double*)malloc(sizeof(double) * 2 * (size_t)n);
    double* y = (double*)malloc(sizeof(double) * 2 * (size_t)n);
    double* m = (double*)malloc(sizeof(double) * (size_t)n);
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
    for (long f = 0; f < (long)n_frames; f++) {
      const float* src = in + 2 * (size_t)f * (size_t)n;
      for (int i = 0; i < 2 * n; i++) x[i] = (double)src[i];
      if (c->window != ORC_WIN_NONE)
        for (int i = 0; i < n; i++) {
          double wv = (double)(float)orc_window_coeff(c->window, i, n); /* fp32 coefficient table, as the device holds */
          x[2 * i] *= wv;
          x[2 * i + 1] *= wv;
        }
      fft_f64_strided(x, 1, c->log2n, tw, y);
      for (int i = 0; i < n; i++) {
        int p = c->no_bit_reverse ? (int)bitrev((unsigned)i, c->log2n) : i; /* position i holds bin p */
        m[i] = orc_mag_f64(y[2 * p], y[2 * p + 1], c->mag_mode);
      }
      if (mag_out) memcpy(mag_out + (size_t)f * (size_t)n, m, sizeof(double) * (size_t)n);
      orc_cfar_f64(m, c, thr + (size_t)f * (size_t)n, peak + (size_t)f * (size_t)n,
                   margin ? margin + (size_t)f * (size_t)n : NULL);
    }
    free(x);
    free(y);
    free(m);
  }
  free(tw);
}

/* ------------------------------------------------------------------ 2-D range-Doppler */

/*
 * No reference counterpart: every chain in src/main/scala holds one
 * 1-D FFT.  Defined by BASELINE.json configs 3/5.  Training region = the
 * (2(ref_r+guard_r)+1) x (2(ref_d+guard_d)+1) box around the CUT minus the
 * (2 guard_r+1) x (2 guard_d+1) guard box; statistic = training sum / training
 * cell count (out-of-map cells count as zeros, as in the 1-D zero-edge rule);
 * Doppler axis cyclic, range axis per c->edge.
 */
void orc_rd_f32in(const float* in, size_t n_ch, const orc_rdcfg* c, double* thr, uint8_t* peak,
                  double* margin, double* mag_out, int n_threads) {
  int nr = 1 << c->log2nr, nd = 1 << c->log2nd;
  size_t map = (size_t)nr * (size_t)nd;
  double* twr = make_tw_f64(c->log2nr);
  double* twd = make_tw_f64(c->log2nd);
  int hr = c->ref_r + c->guard_r, hd = c->ref_d + c->guard_d;
  int er = nr + 2 * hr, ed = nd + 2 * hd; /* halo-extended map */
  double count = (double)(2 * hr + 1) * (2 * hd + 1) - (double)(2 * c->guard_r + 1) * (2 * c->guard_d + 1);
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(n_threads > 0 ? n_threads : 1)
#endif
  for (long ch = 0; ch < (long)n_ch; ch++) {
    double* a = (double*)malloc(sizeof(double) * 2 * map);
    double* b = (double*)malloc(sizeof(double) * 2 * map);
    double* m = (double*)malloc(sizeof(double) * map);
    double* col = (double*)malloc(sizeof(double) * 2 * (size_t)nd);
    double* sat = (double*)calloc((size_t)(er + 1) * (size_t)(ed + 1), sizeof(double));
    const float* src = in + 2 * map * (size_t)ch;
    for (size_t i = 0; i < 2 * map; i++) a[i] = (double)src[i];
    if (c->window_r != ORC_WIN_NONE || c->window_d != ORC_WIN_NONE)
      for (int d = 0; d < nd; d++) {
        double wd = (double)(float)orc_window_coeff(c->window_d, d, nd);
        for (int r = 0; r < nr; r++) {
          double wv = wd * (double)(float)orc_window_coeff(c->window_r, r, nr);
          a[2 * ((size_t)d * (size_t)nr + (size_t)r)] *= wv;
          a[2 * ((size_t)d * (size_t)nr + (size_t)r) + 1] *= wv;
        }
      }
    for (int d = 0; d < nd; d++) /* range FFT along r (contiguous) */
      fft_f64_strided(a + 2 * (size_t)d * (size_t)nr, 1, c->log2nr, twr, b + 2 * (size_t)d * (size_t)nr);
    for (int r = 0; r < nr; r++) { /* Doppler FFT along d (stride nr) */
      fft_f64_strided(b + 2 * (size_t)r, (size_t)nr, c->log2nd, twd, col);
      for (int d = 0; d < nd; d++)
        m[(size_t)d * (size_t)nr + (size_t)r] = orc_mag_f64(col[2 * d], col[2 * d + 1], c->mag_mode);
    }
    if (mag_out) memcpy(mag_out + map * (size_t)ch, m, sizeof(double) * map);
    /* summed-area table over the halo-extended map: sat[(d+1)*(er+1) + (r+1)] */
    for (int d = 0; d < ed; d++) {
      double rowsum = 0.0;
      int sd = (((d - hd) % nd) + nd) % nd; /* Doppler always cyclic */
      for (int r = 0; r < er; r++) {
        int sr = cell_index(r - hr, 0, nr, c->edge);
        rowsum += sr < 0 ? 0.0 : m[(size_t)sd * (size_t)nr + (size_t)sr];
        sat[(size_t)(d + 1) * (size_t)(er + 1) + (size_t)(r + 1)] =
            sat[(size_t)d * (size_t)(er + 1) + (size_t)(r + 1)] + rowsum;
      }
    }
#define BOX(d0, d1, r0, r1) /* inclusive ext coords */                               \
  (sat[(size_t)((d1) + 1) * (size_t)(er + 1) + (size_t)((r1) + 1)] -                 \
   sat[(size_t)(d0) * (size_t)(er + 1) + (size_t)((r1) + 1)] -                       \
   sat[(size_t)((d1) + 1) * (size_t)(er + 1) + (size_t)(r0)] +                       \
   sat[(size_t)(d0) * (size_t)(er + 1) + (size_t)(r0)])
    for (int d = 0; d < nd; d++) {
      for (int r = 0; r < nr; r++) {
        int cd = d + hd, cr = r + hr;
        double t;
        if (c->cfar_mode == ORC_CFAR_CA) {
          double outer = BOX(cd - hd, cd + hd, cr - hr, cr + hr);
          double inner = BOX(cd - c->guard_d, cd + c->guard_d, cr - c->guard_r, cr + c->guard_r);
          t = c->scaler * (outer - inner) / count;
        } else {
          int gd = c->guard_d, gr = c->guard_r;
          double lag = BOX(cd - hd, cd + hd, cr - hr, cr - 1) - (gr > 0 ? BOX(cd - gd, cd + gd, cr - gr, cr - 1) : 0.0) +
                       BOX(cd - hd, cd - gd - 1, cr, cr);
          double lead = BOX(cd - hd, cd + hd, cr + 1, cr + hr) - (gr > 0 ? BOX(cd - gd, cd + gd, cr + 1, cr + gr) : 0.0) +
                        BOX(cd + gd + 1, cd + hd, cr, cr);
          double half = 0.5 * count;
          double st = c->cfar_mode == ORC_CFAR_GO ? (lag > lead ? lag : lead) : (lag < lead ? lag : lead);
          t = c->scaler * st / half;
        }
        size_t o = map * (size_t)ch + (size_t)d * (size_t)nr + (size_t)r;
        double cut = m[(size_t)d * (size_t)nr + (size_t)r];
        thr[o] = t;
        peak[o] = (uint8_t)(cut > t);
        if (margin) margin[o] = fabs(cut - t);
      }
    }
#undef BOX
    free(a);
    free(b);
    free(m);
    free(col);
    free(sat);
  }
  free(twr);
  free(twd);
}

/* 2-D chain on the FIXED16 data path (BUILD-DEFINED like the float one: no reference counterpart).
 * Range FFT per row and Doppler FFT per column with the arithmetic of orc_fft_fixed (1-bit trim per stage, Q2.14
 * twiddles, c->trim; stage options not supported), windows in Q1.15 as in orc_chain_fixed (the Doppler window is
 * applied to the range spectrum), magnitude = orc_mag_fixed, 2-D CFAR over the same training region as orc_rd_f32in
 * in integer arithmetic: CA: statistic = training sum >> div_sum; GO / SO: each half's sum >> (div_sum - 1) (a half
 * holds half the cells; div_sum = 0 shifts by 0), the greater / smaller; threshold, saturation and peak test as
 * orc_cfar_fixed; word = orc_pack_out(threshold, range bin, peak, log2nr).  c->log2n = log2nr, c->ref_window /
 * guard_window = the range half-widths, c->window = the range window. */
void orc_rd_fixed(const uint32_t* in_beats, size_t n_ch, const orc_cfg* c, int log2nd, int ref_d, int guard_d,
                  int window_d, uint32_t* out_words, int32_t* mag_out, int n_threads) {
  int log2nr = c->log2n, nr = 1 << log2nr, nd = 1 << log2nd;
  size_t map = (size_t)nr * (size_t)nd;
  int ref_r = c->ref_window, guard_r = c->guard_window;
  int hr = ref_r + guard_r, hd = ref_d + guard_d;
  int er = nr + 2 * hr, ed = nd + 2 * hd;
  (void)n_threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(n_threads > 0 ? n_threads : 1)
#endif
  for (long ch = 0; ch < (long)n_ch; ch++) {
    int16_t* re = (int16_t*)malloc(sizeof(int16_t) * 2 * map);
    int16_t* im = re + map;
    int big = nr > nd ? nr : nd;
    int16_t* t = (int16_t*)malloc(sizeof(int16_t) * 4 * (size_t)big);
    int16_t *ti = t + big, *fr = t + 2 * big, *fi = t + 3 * big;
    int32_t* m = (int32_t*)malloc(sizeof(int32_t) * map);
    int64_t* sat = (int64_t*)calloc((size_t)(er + 1) * (size_t)(ed + 1), sizeof(int64_t));
    const uint32_t* src = in_beats + map * (size_t)ch;
    for (int d = 0; d < nd; d++) { /* range FFT along r */
      for (int r = 0; r < nr; r++) {
        orc_unpack_iq(src[(size_t)d * (size_t)nr + (size_t)r], &t[r], &ti[r]);
        if (c->window != ORC_WIN_NONE) {
          int32_t wq = (int32_t)lround(orc_window_coeff(c->window, r, nr) * 32767.0);
          t[r] = (int16_t)(((int32_t)t[r] * wq + (1 << 14)) >> 15);
          ti[r] = (int16_t)(((int32_t)ti[r] * wq + (1 << 14)) >> 15);
        }
      }
      orc_fft_fixed_ex(t, ti, log2nr, c->trim, 0, 0, 0, re + (size_t)d * (size_t)nr, im + (size_t)d * (size_t)nr);
    }
    for (int r = 0; r < nr; r++) { /* Doppler FFT along d */
      for (int d = 0; d < nd; d++) {
        t[d] = re[(size_t)d * (size_t)nr + (size_t)r];
        ti[d] = im[(size_t)d * (size_t)nr + (size_t)r];
        if (window_d != ORC_WIN_NONE) {
          int32_t wq = (int32_t)lround(orc_window_coeff(window_d, d, nd) * 32767.0);
          t[d] = (int16_t)(((int32_t)t[d] * wq + (1 << 14)) >> 15);
          ti[d] = (int16_t)(((int32_t)ti[d] * wq + (1 << 14)) >> 15);
        }
      }
      orc_fft_fixed_ex(t, ti, log2nd, c->trim, 0, 0, 0, fr, fi);
      for (int d = 0; d < nd; d++) m[(size_t)d * (size_t)nr + (size_t)r] = orc_mag_fixed(fr[d], fi[d], c);
    }
    if (mag_out) memcpy(mag_out + map * (size_t)ch, m, sizeof(int32_t) * map);
    for (int d = 0; d < ed; d++) { /* summed-area table over the halo-extended map */
      int64_t rowsum = 0;
      int sd = (((d - hd) % nd) + nd) % nd; /* Doppler always cyclic */
      for (int r = 0; r < er; r++) {
        int sr = cell_index(r - hr, 0, nr, c->edge);
        rowsum += sr < 0 ? 0 : (int64_t)m[(size_t)sd * (size_t)nr + (size_t)sr];
        sat[(size_t)(d + 1) * (size_t)(er + 1) + (size_t)(r + 1)] = sat[(size_t)d * (size_t)(er + 1) + (size_t)(r + 1)] + rowsum;
      }
    }
#define BOX(d0, d1, r0, r1) /* inclusive ext coords */                               \
  (sat[(size_t)((d1) + 1) * (size_t)(er + 1) + (size_t)((r1) + 1)] -                 \
   sat[(size_t)(d0) * (size_t)(er + 1) + (size_t)((r1) + 1)] -                       \
   sat[(size_t)((d1) + 1) * (size_t)(er + 1) + (size_t)(r0)] +                       \
   sat[(size_t)(d0) * (size_t)(er + 1) + (size_t)(r0)])
    for (int d = 0; d < nd; d++) {
      for (int r = 0; r < nr; r++) {
        int cd = d + hd, cr = r + hr, gd = guard_d, gr = guard_r;
        int64_t stat;
        if (c->cfar_mode == ORC_CFAR_CA) {
          stat = (BOX(cd - hd, cd + hd, cr - hr, cr + hr) - BOX(cd - gd, cd + gd, cr - gr, cr + gr)) >> c->div_sum;
        } else {
          int64_t lag = BOX(cd - hd, cd + hd, cr - hr, cr - 1) - (gr > 0 ? BOX(cd - gd, cd + gd, cr - gr, cr - 1) : 0) +
                        BOX(cd - hd, cd - gd - 1, cr, cr);
          int64_t lead = BOX(cd - hd, cd + hd, cr + 1, cr + hr) - (gr > 0 ? BOX(cd - gd, cd + gd, cr + 1, cr + gr) : 0) +
                         BOX(cd + gd + 1, cd + hd, cr, cr);
          int sh = c->div_sum > 0 ? c->div_sum - 1 : 0;
          lag >>= sh;
          lead >>= sh;
          stat = c->cfar_mode == ORC_CFAR_GO ? (lag > lead ? lag : lead) : (lag < lead ? lag : lead);
        }
        int64_t thr = threshold_fixed(stat, c);
        int64_t cut = m[(size_t)d * (size_t)nr + (size_t)r];
        int peak = cut * ((int64_t)1 << c->bp_thr) > thr * ((int64_t)1 << c->bp_in);
        out_words[map * (size_t)ch + (size_t)d * (size_t)nr + (size_t)r] =
            orc_pack_out((uint32_t)(int32_t)thr, (uint32_t)r, (uint32_t)peak, log2nr);
      }
    }
#undef BOX
    free(re);
    free(t);
    free(m);
    free(sat);
  }
}

/* ------------------------------------------------------------------ PLFG -> NCO stimulus */

void orc_plfg(const orc_stim_cfg* c, size_t n, int32_t* values) {
  size_t i = 0;
  if (!c->enable) {
    for (; i < n; i++) values[i] = 0;
    return;
  }
  while (i < n) {
    size_t before = i;
    for (int ch = 0; ch < c->num_chirps && i < n; ch++) {
      int o = c->ordinal[ch];
      for (int rep = 0; rep < c->repeated[ch] && i < n; rep++) {
        int32_t v = c->start_value;
        for (int sg = 0; sg < c->segment_nums[o] && i < n; sg++) {
          uint32_t w = c->ram[o * c->max_segments + sg];
          int len = (int)(w >> 24);
          int32_t slope = (int32_t)((w >> 8) & 0xFFFF);
          if (w & 2u) slope = -slope;
          if (w & 1u) v = c->start_value;
          for (int k = 0; k < len && i < n; k++) {
            values[i++] = v;
            v += slope;
          }
        }
      }
    }
    if (i == before) { /* empty program: hold the start value */
      for (; i < n; i++) values[i] = c->start_value;
    }
  }
}

static int32_t nco_quarter(const orc_stim_cfg* c, int k) { /* k in 0..table_size */
  double v = sin(2.0 * M_PI * (double)k / (4.0 * (double)c->table_size)) * ldexp(1.0, c->table_width - 2);
  return (int32_t)floor(v + 0.5);
}

static int32_t nco_sin(const orc_stim_cfg* c, uint32_t phase) {
  uint32_t ts = (uint32_t)c->table_size, q = phase / ts, r = phase % ts;
  switch (q & 3u) {
    case 0: return nco_quarter(c, (int)r);
    case 1: return nco_quarter(c, (int)(ts - r));
    case 2: return -nco_quarter(c, (int)r);
    default: return -nco_quarter(c, (int)(ts - r));
  }
}

void orc_plfg_nco(const orc_stim_cfg* c, size_t n, uint32_t* beats) {
  int32_t* v = (int32_t*)malloc(sizeof(int32_t) * (n ? n : 1));
  orc_plfg(c, n, v);
  uint32_t mask = (1u << c->phase_width) - 1u, phase = 0;
  for (size_t i = 0; i < n; i++) {
    phase = (phase + (uint32_t)v[i]) & mask;
    int32_t sn = nco_sin(c, phase), cs = nco_sin(c, (phase + (uint32_t)c->table_size) & mask);
    int32_t lim = (1 << (c->table_width - 1)) - 1; /* 2^14 fits 16 bits; clamp for narrower tables */
    if (sn > lim) sn = lim;
    if (cs > lim) cs = lim;
    beats[i] = orc_pack_iq(cs, sn);
  }
  free(v);
}
