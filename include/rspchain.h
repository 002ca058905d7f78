/*
 * rspchain.h -- C ABI of the MI355X-native sdf-fft -> logMagMux -> CFAR path.
 *
 * This is the drop-in boundary: every entry point replaces one interface of the
 * reference (milovanovic/rsp-chains, a Chisel generator; citations relative to
 * /root/reference/).  The reference drives its chain through exactly three
 * things -- an elaboration-time parameter object, memory-mapped register writes,
 * and a 32-bit AXI4-Stream in / out -- and so does this library:
 *
 *   reference                                              this ABI
 *   -----------------------------------------------------  -------------------------
 *   FftMagCfarVanillaParameters(fftParams, magParams,      rsp_chain_params
 *     cfarParams, *Address, beatBytes)                       (+ rsp_chain_default_params)
 *     src/main/scala/FftMagCfarChain.scala:21-29,77-116
 *   LazyModule(new FftMagCfarChainVanilla(params))         rsp_chain_create / _destroy
 *     FftMagCfarChain.scala:31-49,119
 *   AXI4MasterModel.memWriteWord(addr, value)              rsp_chain_write_reg / _read_reg
 *     src/test/scala/FftMagCfarChainTester.scala:82-132
 *   AXI4StreamModel master.addTransactions(beats, last)    rsp_chain_process (host buffers)
 *     + peek(out.bits.data) loop, fftSize words            rsp_chain_process_device (HBM)
 *     FftMagCfarChainTester.scala:137,145-151
 *
 * Plain C types only.  Return value 0 = RSP_OK, negative = error; the message
 * is available from rsp_last_error() (thread-local).  Nothing throws across the
 * ABI.  The caller owns every buffer it passes; the library owns the device
 * buffers, twiddle ROMs and stream inside a handle.  A handle is not
 * thread-safe: one handle per host thread; distinct handles may run
 * concurrently.  There is no CPU fallback: without a HIP device
 * rsp_chain_create fails with RSP_ERR_DEVICE.
 */
#ifndef RSPCHAIN_H
#define RSPCHAIN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RSP_ABI_VERSION 3
#define RSP_MAX_STAGES 16

enum {
  RSP_OK = 0,
  RSP_ERR_INVALID = -1,     /* a Scala `require` of the reference would have failed   */
  RSP_ERR_UNSUPPORTED = -2, /* legal in the reference, not implemented on the GPU path */
  RSP_ERR_DEVICE = -3,      /* no HIP device / HIP runtime error                       */
  RSP_ERR_ADDRESS = -4,     /* register address decodes to no block (AXI DECERR)       */
  RSP_ERR_NOMEM = -5
};

/* FixedPoint(width.W, binaryPoint.BP), e.g. FftMagCfarChain.scala:102-104 */
typedef struct rsp_fixed_proto {
  int32_t width;
  int32_t binaryPoint;
} rsp_fixed_proto;

/* AddressSet(base, mask), FftMagCfarChain.scala:113-115 */
typedef struct rsp_address_set {
  uint32_t base;
  uint32_t mask;
} rsp_address_set;

/* FFTParams.fixed(...), FftMagCfarChain.scala:78-90 (same field names). */
typedef struct rsp_fft_params {
  int32_t dataWidth;
  int32_t twiddleWidth;
  int32_t numPoints;
  int32_t useBitReverse;
  int32_t runTime;
  int32_t numAddPipes; /* latency only: no numerical meaning on the GPU */
  int32_t numMulPipes; /* latency only */
  int32_t expandLogic[RSP_MAX_STAGES];
  int32_t keepMSBorLSB[RSP_MAX_STAGES];
  int32_t minSRAMdepth; /* area only */
  int32_t binPoint;
  int32_t trimType; /* build extension: RSP_TRIM_*; reference uses upstream default */
} rsp_fft_params;

/* MAGParams.fixed(...), FftMagCfarChain.scala:91-100 */
typedef struct rsp_mag_params {
  int32_t dataWidth;
  int32_t binPoint;
  int32_t dataWidthLog;
  int32_t binPointLog;
  int32_t log2LookUpWidth;
  int32_t useLast;
  int32_t numAddPipes;
  int32_t numMulPipes;
} rsp_mag_params;

/* CFARParams(...), FftMagCfarChain.scala:101-112 */
typedef struct rsp_cfar_params {
  rsp_fixed_proto protoIn;
  rsp_fixed_proto protoThreshold;
  rsp_fixed_proto protoScaler;
  int32_t leadLaggWindowSize;
  int32_t guardWindowSize;
  int32_t sendCut;
  int32_t fftSize;
  int32_t minSubWindowSize; /* -1 = None */
  int32_t includeCASH;
  int32_t CFARAlgorithm; /* RSP_ALG_CA / _GOS / _GOSCA: FftMagCfarChainTester.scala:105,110,123 */
  int32_t numMulPipes;
  int32_t edgeMode; /* build extension: RSP_EDGE_* */
} rsp_cfar_params;

enum { RSP_TRIM_FLOOR = 0, RSP_TRIM_HALF_UP = 1, RSP_TRIM_CONVERGENT = 2 };
enum { RSP_ALG_CA = 0, RSP_ALG_GOS = 1, RSP_ALG_GOSCA = 2 };
enum { RSP_EDGE_ZERO = 0, RSP_EDGE_WRAP = 1 };
enum { RSP_MAG_SQR = 0, RSP_MAG_LOG2 = 1, RSP_MAG_JPL = 2 }; /* 2: FftMagCfarChainTester.scala:84 */
enum { RSP_WINDOW_NONE = 0, RSP_WINDOW_HANN = 1, RSP_WINDOW_HAMMING = 2, RSP_WINDOW_BLACKMAN = 3 };
enum { RSP_MODE_CA = 0, RSP_MODE_GO = 1, RSP_MODE_SO = 2, RSP_MODE_CASH = 3 }; /* Tester:86-92 */

/* Sample type streamed through the chain.  FIXED16 is the reference's
 * FixedPoint(16.W) data path with its 32-bit {re[31:16], im[15:0]} beat
 * (src/test/scala/RspChainTesterUtils.scala:105-109).  F32 is the fp32 data
 * path BASELINE.json's GPU configs ask for: a beat is one interleaved
 * complex64 {re, im} and an output word is the fp32 threshold with its
 * mantissa LSB replaced by the peak flag (bin index = position). */
enum { RSP_DTYPE_FIXED16 = 0, RSP_DTYPE_F32 = 1 };

/* FftMagCfarVanillaParameters, FftMagCfarChain.scala:21-29, + GPU extensions. */
typedef struct rsp_chain_params {
  rsp_fft_params fftParams;
  rsp_mag_params magParams;
  rsp_cfar_params cfarParams;
  rsp_address_set fftAddress;
  rsp_address_set magAddress;
  rsp_address_set cfarAddress;
  int32_t beatBytes;
  /* ---- extensions with no reference counterpart ---- */
  int32_t dtype;     /* RSP_DTYPE_* */
  int32_t device;    /* HIP device ordinal */
  int32_t dopplerPoints; /* 0 = 1-D chain; else slow-time FFT size of the 2-D range-Doppler chain (either dtype) */
  int32_t refDoppler;    /* 2-D CFAR training / guard half-widths along Doppler */
  int32_t guardDoppler;
  int32_t window;        /* RSP_WINDOW_*: pre-FFT window over fast time (range); SURVEY 8f-n4, no reference item */
  int32_t windowDoppler; /* RSP_WINDOW_*: 2-D chain, window over slow time */
  int32_t reserved[6];
} rsp_chain_params;

typedef struct rsp_chain rsp_chain;

/* One detection of the compact list (rsp_chain_detections_device). */
typedef struct rsp_detection {
  uint32_t frame; /* frame (1-D) or channel (2-D) index within the call        */
  uint32_t bin;   /* range bin                                                   */
  uint32_t doppler; /* Doppler bin (0 for the 1-D chain)                         */
  uint32_t word;  /* the dense output word of that cell                          */
} rsp_detection;

/* --- construction --------------------------------------------------------- */
uint32_t rsp_abi_version(void);
const char* rsp_last_error(void);
/* Parameters of FftMagCfarChainVanillaApp, FftMagCfarChain.scala:77-116. */
void rsp_chain_default_params(rsp_chain_params* p);
/* Checks what elaboration would check; usable without a device. */
int rsp_chain_validate_params(const rsp_chain_params* p);
int rsp_chain_create(const rsp_chain_params* p, rsp_chain** out);
void rsp_chain_destroy(rsp_chain* c);

/* --- control plane: AXI4 register file ------------------- */
/* memWriteWord(addr, value), FftMagCfarChainTester.scala:82-132.  Register state
 * after create = RunTimeRspChainParams() defaults
 * (src/test/scala/RspChainVanillaTester.scala:35-48) with fftSize = numPoints. */
int rsp_chain_write_reg(rsp_chain* c, uint32_t addr, uint32_t value);
int rsp_chain_read_reg(rsp_chain* c, uint32_t addr, uint32_t* value);
/* Cross-register `require`s of RunTimeRspChainParams (RspChainVanillaTester.scala:50-61)
 * + consistency FFT stages <-> CFAR fftSize; called by process, exposed for hosts. */
int rsp_chain_check_regs(rsp_chain* c);

/* --- data plane ------------------------------------------------------------ */
/* Stream n_frames frames of fftSize beats in (TLAST is implied on the final
 * beat of every frame, Tester:137), collect fftSize output words per frame
 * (Tester:145-151).  Host buffers; synchronous.  Beat = 4 bytes (FIXED16) or
 * 8 bytes (F32).  For the 2-D chain a "frame" is one channel's
 * dopplerPoints x fftSize map, [doppler][range] row-major.
 * CFARParams.sendCut = true (FftMagCfarChain.scala:107) widens the output beat to 64 bits: TWO
 * words per cell, {the word described above, the cell under test} (FIXED16: the magnitude as a
 * sign-extended integer; F32: its fp32 bits) -- out_words then holds 2 x fftSize words per frame.
 * FFTParams.useBitReverse = false: the FFT block streams its result in bit-reversed order and the
 * blocks behind it work on that order; position p of a frame then refers to bin bitrev(p).
 * The batch moves as a pipeline of chunks -- H2D(k + 1) || kernels(k) || D2H(k - 1) on three streams.  Buffers
 * from rsp_host_alloc, or pinned with rsp_host_register, are DMA'd in place at the link's rate; pageable
 * memory (a plain malloc / JVM direct buffer / numpy array) is staged through a ring of pinned chunks by a pool
 * of copy threads.  Results are identical either way. */
int rsp_chain_process(rsp_chain* c, const void* in_beats, size_t n_frames, uint32_t* out_words);
/* Same with buffers already resident in HBM; asynchronous on the chain's stream. */
int rsp_chain_process_device(rsp_chain* c, const void* d_in_beats, size_t n_frames,
                             uint32_t* d_out_words);
/* Fused form: one pass produces the dense words AND the compact detection list.
 * d_count points at TWO device uint32: d_count[0] = peaks found, d_count[1] = entries
 * stored in d_list (<= cap); stored < found means the list is truncated.  With dense
 * words the list is complete up to cap (the reference emits every peak, Tester:145-167).
 * d_out_words may be NULL: detection-list-only output (8 B read per cell, no dense
 * write); then a frame contributes at most RSP_FRAME_DET_CAP peaks (the kernel's
 * per-frame staging) and d_count[1] < d_count[0] tells when that limit hit.
 * 1-D chain, list order: up to 16384 frames per call the frames appear in ascending order (a
 * frame's peaks contiguous, in no particular order); above that only a frame's peaks stay contiguous.
 * 2-D chain: the CFAR kernel appends its peak cells to the list itself (dense words are
 * always written: d_out_words must not be NULL); d_count is also the list's cursor while the call
 * runs (zeroed by its first kernel); list order is unspecified. */
#define RSP_FRAME_DET_CAP 64
int rsp_chain_process_detect_device(rsp_chain* c, const void* d_in_beats, size_t n_frames,
                                    uint32_t* d_out_words, rsp_detection* d_list, uint32_t cap,
                                    uint32_t* d_count);
/* Compact the peak cells of a dense device result into list[0..cap) (device
 * memory); d_count[0] (device uint32[2]) receives the number found (may exceed cap),
 * d_count[1] the number stored.  Order within the list is unspecified. */
int rsp_chain_detections_device(rsp_chain* c, const uint32_t* d_out_words, size_t n_frames,
                                rsp_detection* d_list, uint32_t cap, uint32_t* d_count);
/* Host-buffer convenience: dense process + compaction (complete: no per-frame limit);
 * list sorted by (frame, doppler, bin); *n_found may exceed cap, min(*n_found, cap) entries are returned. */
int rsp_chain_process_detections(rsp_chain* c, const void* in_beats, size_t n_frames,
                                 rsp_detection* list, size_t cap, size_t* n_found);

/* Tuning / test knobs (no reference counterpart; replaces environment variables of round 1). */
enum {
  RSP_OPT_MAX_FRAMES_PER_LAUNCH = 1, /* split a call into launches of at most this many frames (0 = automatic) */
  RSP_OPT_FORCE_TILED_CFAR2D = 2,    /* 2-D chain: run-time-window CFAR kernel even for the compile-time windows */
  RSP_OPT_FORCE_GENERIC_TAIL = 3,    /* 1-D chain: per-cell CFAR tail even where the 16-byte "quad" tail applies */
  RSP_OPT_EXPERIMENT = 5,            /* A/B runs: 1 = hand 1-D launches to the experimental kernel of a side library built from tools/experiments/ (no effect on the product library) */
  RSP_OPT_HOST_CHUNK_BYTES = 6,      /* host-buffer entries: input bytes per chunk of the H2D || kernel || D2H pipeline (0 = automatic, ~16 MiB) */
  RSP_OPT_RD_CHUNK_BYTES = 4         /* 2-D chain: bytes of intermediates (12 B/cell) per chunk of channels; 0 = whole batch (default: chunks of 48-192 MiB measured 0-60 % slower) */
};
int rsp_chain_set_option(rsp_chain* c, int option, int64_t value);

/* --- stream / memory / timing plumbing -------------------------------------- */
/* Run on a caller-owned hipStream_t (e.g. the stream a host framework is already using); NULL restores
 * the chain's own stream. */
int rsp_chain_set_stream(rsp_chain* c, void* hip_stream);
int rsp_chain_synchronize(rsp_chain* c);
/* hipEvent pair on the chain's stream around whatever is enqueued between the calls. */
int rsp_chain_timer_start(rsp_chain* c);
int rsp_chain_timer_stop(rsp_chain* c, float* elapsed_ms); /* synchronises */
/* Per-launch timing of the dominant (chain) kernel alone: while enabled, every data-plane call of the 1-D chain binds
 * a HIP event pair to the kernel's dispatch (hipExtLaunchKernelGGL: the kernel's own begin / end timestamps -- what
 * rocprofv3's kernel trace reports; an event pair recorded around the launch would add the ~2 us of the dispatch);
 * the 2-D chain brackets its three kernels with one recorded pair.  _read synchronises, returns the summed durations
 * and the number of launches, and resets. */
int rsp_chain_profile_enable(rsp_chain* c, int on);
int rsp_chain_profile_read(rsp_chain* c, float* total_ms, uint32_t* launches);
int rsp_device_count(int* n);
int rsp_device_malloc(int device, void** ptr, size_t bytes);
int rsp_device_free(int device, void* ptr);
/* Pinned (page-locked) host memory for the host-buffer entry points: allocate the stream buffers here, or pin
 * caller-owned memory for as long as it is used with the chain (unregister it BEFORE freeing it). */
int rsp_host_alloc(int device, void** ptr, size_t bytes);
int rsp_host_free(void* ptr);
int rsp_host_register(int device, void* ptr, size_t bytes);
int rsp_host_unregister(void* ptr);
int rsp_memcpy_h2d(int device, void* dst, const void* src, size_t bytes);
int rsp_memcpy_d2h(int device, void* dst, const void* src, size_t bytes);

/* --- PLFG -> NCO stimulus of the full chain (RspChainVanilla) ----------------------------------
 * Replaces PLFGDspBlockMem + AXI4NCOLazyModuleBlock (src/main/scala/RspChain.scala:41-42) and the
 * wiring nco.freq := plfg.streamNode, fft.streamNode := nco.streamNode (:57-58): generates the
 * chain's input beats on the device instead of reading them from HBM.  Generator sources are
 * empty submodules, so the model is build-defined (spelled out in DESIGN.md), anchored on
 * the parameter sets (RspChain.scala:84-106) and the tester's register program
 * (src/test/scala/RspChainVanillaTester.scala:80-94). */
typedef struct rsp_plfg_params { /* FixedPLFGParams, RspChain.scala:84-93 */
  int32_t maxNumOfSegments, maxNumOfDifferentChirps, maxNumOfRepeatedChirps, maxChirpOrdinalNum,
      maxNumOfFrames, maxNumOfSamplesWidth, outputWidthInt, outputWidthFrac;
} rsp_plfg_params;
typedef struct rsp_nco_params { /* FixedNCOParams, RspChain.scala:94-106 */
  int32_t tableSize, tableWidth, phaseWidth, rasterizedMode, nInterpolationTerms, ditherEnable,
      syncROMEnable, phaseAccEnable, roundingMode /* 0 = RoundHalfUp */, pincType /* 0 = Streaming */,
      poffType /* 0 = Fixed */;
} rsp_nco_params;
typedef struct rsp_stimulus_params {
  rsp_plfg_params plfgParams;
  rsp_nco_params ncoParams;
  rsp_address_set plfgAddress; /* RspChain.scala:141 */
  rsp_address_set plfgRAM;     /* :142 */
  rsp_address_set ncoAddress;  /* :143 */
  int32_t beatBytes;
  int32_t device;
} rsp_stimulus_params;
typedef struct rsp_stimulus rsp_stimulus;
void rsp_stimulus_default_params(rsp_stimulus_params* p); /* RspChainVanillaApp, RspChain.scala:83-106,141-143 */
int rsp_stimulus_create(const rsp_stimulus_params* p, rsp_stimulus** out);
void rsp_stimulus_destroy(rsp_stimulus* s);
/* PLFG registers / RAM words: RspChainVanillaTester.scala:80-94 (offsets in beats: 0 enable,
 * 1 reset, 2 frames, 4 chirps, 5 start value, 6.. segments per chirp type, +4.. repeats, +8.. ordinals) */
int rsp_stimulus_write_reg(rsp_stimulus* s, uint32_t addr, uint32_t value);
int rsp_stimulus_read_reg(rsp_stimulus* s, uint32_t addr, uint32_t* value);
/* n_samples beats {cos[31:16], sin[15:0]} from reset, into device / host memory */
int rsp_stimulus_generate_device(rsp_stimulus* s, uint32_t* d_beats, size_t n_samples, void* hip_stream);
int rsp_stimulus_generate(rsp_stimulus* s, uint32_t* beats, size_t n_samples);
const char* rsp_stimulus_last_error(void);

/* --- wire-format helpers ---------------------------------------------------- */
/* formAXI4StreamComplexData, RspChainTesterUtils.scala:105-109 */
uint32_t rsp_pack_iq(int32_t re, int32_t im);
/* Tester:163-167: threshold = word >> (log2(fftSize)+1) (arithmetic), peak = word & 1 */
void rsp_unpack_word(uint32_t word, int32_t log2_fft_size, int32_t* threshold, uint32_t* bin,
                     uint32_t* peak);
/* F32 output word: threshold bits with LSB cleared, peak in the LSB. */
void rsp_unpack_word_f32(uint32_t word, float* threshold, uint32_t* peak);

#ifdef __cplusplus
}
#endif
#endif
