"""ctypes front-end of the CPU oracle (oracle/rsp_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py, never by the product package.  PARITY UNPINNED: see
the header of rsp_oracle.h.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "librsp_oracle.so")

TRIM_FLOOR, TRIM_HALF_UP, TRIM_CONVERGENT = 0, 1, 2
WIN_NONE, WIN_HANN, WIN_HAMMING, WIN_BLACKMAN = 0, 1, 2, 3
MAG_SQR, MAG_LOG2, MAG_JPL = 0, 1, 2
CFAR_CA, CFAR_GO, CFAR_SO, CFAR_CASH = 0, 1, 2, 3
EDGE_ZERO, EDGE_WRAP = 0, 1


class OrcCfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "log2n", "trim", "mag_mode", "bp_data", "bp_log", "log2_lut_width", "bp_in",
        "bp_thr", "w_thr", "bp_scaler")] + [("scaler", C.c_uint32)] + [(n, C.c_int32) for n in (
            "linear", "div_sum", "peak_grouping", "algorithm", "cfar_mode", "ref_window",
            "guard_window", "index_lagg", "index_lead", "sub_window", "edge")] + [
                ("keep_lsb_mask", C.c_uint32), ("expand_mask", C.c_uint32), ("no_bit_reverse", C.c_int32),
                ("send_cut", C.c_int32), ("window", C.c_int32)]


class OrcFCfg(C.Structure):
    _fields_ = [("log2n", C.c_int32), ("mag_mode", C.c_int32), ("scaler", C.c_double)] + [
        (n, C.c_int32) for n in ("linear", "div_sum", "peak_grouping", "algorithm", "cfar_mode",
                                 "ref_window", "guard_window", "index_lagg", "index_lead", "edge",
                                 "sub_window", "no_bit_reverse", "window")]


class OrcRdCfg(C.Structure):
    _fields_ = [("log2nr", C.c_int32), ("log2nd", C.c_int32), ("mag_mode", C.c_int32),
                ("scaler", C.c_double)] + [(n, C.c_int32) for n in (
                    "ref_r", "ref_d", "guard_r", "guard_d", "edge", "window_r", "window_d", "cfar_mode")]


class OrcStimCfg(C.Structure):
    _fields_ = [("enable", C.c_int32), ("start_value", C.c_int32), ("num_chirps", C.c_int32),
                ("max_segments", C.c_int32), ("segment_nums", C.c_int32 * 8), ("repeated", C.c_int32 * 8),
                ("ordinal", C.c_int32 * 8), ("ram", C.c_uint32 * 64), ("table_size", C.c_int32),
                ("table_width", C.c_int32), ("phase_width", C.c_int32)]


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (Makefile beside this file)."""
    src = os.path.join(_HERE, "rsp_oracle.c")
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB_PATH)
        for f in ("rsp_oracle.c", "rsp_oracle.h") if os.path.exists(os.path.join(_HERE, f)))
    if force or (stale and os.path.exists(src)):
        subprocess.run(["make", "-C", _HERE, "-B", "librsp_oracle.so"], check=True,
                       stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        try:
            build()
        except Exception:  # no compiler on this box: use the prebuilt library
            if not os.path.exists(_LIB_PATH):
                raise
        L = C.CDLL(_LIB_PATH)
        P = C.POINTER
        L.orc_pack_iq.restype = C.c_uint32
        L.orc_pack_iq.argtypes = [C.c_int32, C.c_int32]
        L.orc_pack_out.restype = C.c_uint32
        L.orc_pack_out.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_int]
        L.orc_twiddles_q14.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
        L.orc_fft_fixed.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_fft_fixed_ex.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_int,
                                       C.c_void_p, C.c_void_p]
        L.orc_window_coeff.restype = C.c_double
        L.orc_window_coeff.argtypes = [C.c_int, C.c_int, C.c_int]
        L.orc_mag_fixed.restype = C.c_int32
        L.orc_mag_fixed.argtypes = [C.c_int16, C.c_int16, P(OrcCfg)]
        L.orc_cfar_fixed.argtypes = [C.c_void_p, P(OrcCfg), C.c_void_p, C.c_void_p]
        L.orc_chain_fixed.argtypes = [C.c_void_p, C.c_size_t, P(OrcCfg), C.c_void_p]
        L.orc_fft_f64.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.orc_mag_f64.restype = C.c_double
        L.orc_mag_f64.argtypes = [C.c_double, C.c_double, C.c_int]
        L.orc_cfar_f64.argtypes = [C.c_void_p, P(OrcFCfg), C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_chain_f32in.argtypes = [C.c_void_p, C.c_size_t, P(OrcFCfg), C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_int]
        L.orc_rd_f32in.argtypes = [C.c_void_p, C.c_size_t, P(OrcRdCfg), C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_int]
        L.orc_rd_fixed.argtypes = [C.c_void_p, C.c_size_t, P(OrcCfg), C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                   C.c_void_p, C.c_int]
        L.orc_plfg.argtypes = [P(OrcStimCfg), C.c_size_t, C.c_void_p]
        L.orc_plfg_nco.argtypes = [P(OrcStimCfg), C.c_size_t, C.c_void_p]
        _lib = L
    return _lib


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def default_cfg(**kw) -> OrcCfg:
    """Register/elaboration state of FftMagCfarChainVanillaSpec + RunTimeRspChainParams()
    (FftMagCfarChainTester.scala:198-241, RspChainVanillaTester.scala:35-48)."""
    c = OrcCfg(log2n=10, trim=TRIM_CONVERGENT, mag_mode=MAG_JPL, bp_data=12, bp_log=9,
               log2_lut_width=9, bp_in=12, bp_thr=12, w_thr=16, bp_scaler=12,
               scaler=int(3.5 * 4096), linear=1, div_sum=5, peak_grouping=0, algorithm=0,
               cfar_mode=CFAR_GO, ref_window=32, guard_window=4, index_lagg=0, index_lead=0,
               sub_window=0, edge=EDGE_ZERO, keep_lsb_mask=0, expand_mask=0, no_bit_reverse=0, send_cut=0, window=0)
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def default_fcfg(**kw) -> OrcFCfg:
    c = OrcFCfg(log2n=12, mag_mode=MAG_JPL, scaler=3.5, linear=1, div_sum=5, peak_grouping=0,
                algorithm=0, cfar_mode=CFAR_CA, ref_window=32, guard_window=4, index_lagg=0,
                index_lead=0, edge=EDGE_ZERO, sub_window=0, no_bit_reverse=0, window=0)
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def pack_iq(re, im) -> np.ndarray:
    re = np.asarray(re).astype(np.int64)
    im = np.asarray(im).astype(np.int64)
    return (((re & 0xFFFF) << 16) | (im & 0xFFFF)).astype(np.uint32)


def twiddles_q14(log2n: int):
    n = 1 << log2n
    wr = np.zeros(n // 2, np.int16)
    wi = np.zeros(n // 2, np.int16)
    lib().orc_twiddles_q14(log2n, _p(wr), _p(wi))
    return wr, wi


def fft_fixed(re, im, trim=TRIM_CONVERGENT, keep_lsb_mask=0, expand_mask=0, no_bit_reverse=0):
    re = np.ascontiguousarray(re, np.int16)
    im = np.ascontiguousarray(im, np.int16)
    n = re.size
    log2n = n.bit_length() - 1
    assert 1 << log2n == n
    orr = np.zeros(n, np.int16)
    oi = np.zeros(n, np.int16)
    lib().orc_fft_fixed_ex(_p(re), _p(im), log2n, trim, keep_lsb_mask, expand_mask, no_bit_reverse, _p(orr), _p(oi))
    return orr, oi


def mag_fixed(re, im, cfg: OrcCfg) -> np.ndarray:
    re = np.asarray(re, np.int16).ravel()
    im = np.asarray(im, np.int16).ravel()
    L = lib()
    return np.array([L.orc_mag_fixed(int(a), int(b), C.byref(cfg)) for a, b in zip(re, im)], np.int32)


def cfar_fixed(mag, cfg: OrcCfg):
    mag = np.ascontiguousarray(mag, np.int32)
    n = 1 << cfg.log2n
    assert mag.size == n
    out = np.zeros(n, np.uint32)
    thr = np.zeros(n, np.int32)
    lib().orc_cfar_fixed(_p(mag), C.byref(cfg), _p(out), _p(thr))
    return out, thr


def chain_fixed(in_beats, cfg: OrcCfg) -> np.ndarray:
    beats = np.ascontiguousarray(in_beats, np.uint32).ravel()
    n = 1 << cfg.log2n
    assert beats.size % n == 0
    out = np.zeros(beats.size * (2 if cfg.send_cut else 1), np.uint32)   # sendCut: {word, cut} per cell
    lib().orc_chain_fixed(_p(beats), beats.size // n, C.byref(cfg), _p(out))
    return out


def fft_f64(x) -> np.ndarray:
    x = np.ascontiguousarray(x, np.complex128)
    n = x.size
    log2n = n.bit_length() - 1
    out = np.zeros(n, np.complex128)
    lib().orc_fft_f64(_p(x), log2n, _p(out))
    return out


def mag_f64(z, mode=MAG_JPL) -> np.ndarray:
    z = np.asarray(z, np.complex128).ravel()
    L = lib()
    return np.array([L.orc_mag_f64(float(v.real), float(v.imag), mode) for v in z])


def cfar_f64(mag, cfg: OrcFCfg):
    mag = np.ascontiguousarray(mag, np.float64)
    n = 1 << cfg.log2n
    assert mag.size == n
    thr = np.zeros(n)
    peak = np.zeros(n, np.uint8)
    margin = np.zeros(n)
    lib().orc_cfar_f64(_p(mag), C.byref(cfg), _p(thr), _p(peak), _p(margin))
    return thr, peak, margin


def chain_f32(x, cfg: OrcFCfg, n_threads: int = 1, want_mag: bool = False):
    """x: complex64 [n_frames, N].  Returns thr, peak, margin (, mag) as [n_frames, N]."""
    x = np.ascontiguousarray(x, np.complex64)
    n = 1 << cfg.log2n
    x = x.reshape(-1, n)
    f = x.shape[0]
    thr = np.zeros((f, n))
    peak = np.zeros((f, n), np.uint8)
    margin = np.zeros((f, n))
    mag = np.zeros((f, n)) if want_mag else None
    lib().orc_chain_f32in(_p(x), f, C.byref(cfg), _p(thr), _p(peak), _p(margin),
                          _p(mag) if want_mag else None, n_threads)
    return (thr, peak, margin, mag) if want_mag else (thr, peak, margin)


def rd_f32(x, cfg: OrcRdCfg, n_threads: int = 1, want_mag: bool = False):
    """x: complex64 [n_ch, n_doppler, n_range]."""
    x = np.ascontiguousarray(x, np.complex64)
    nr, nd = 1 << cfg.log2nr, 1 << cfg.log2nd
    x = x.reshape(-1, nd, nr)
    ch = x.shape[0]
    thr = np.zeros((ch, nd, nr))
    peak = np.zeros((ch, nd, nr), np.uint8)
    margin = np.zeros((ch, nd, nr))
    mag = np.zeros((ch, nd, nr)) if want_mag else None
    lib().orc_rd_f32in(_p(x), ch, C.byref(cfg), _p(thr), _p(peak), _p(margin),
                       _p(mag) if want_mag else None, n_threads)
    return (thr, peak, margin, mag) if want_mag else (thr, peak, margin)


def rd_fixed(in_beats, cfg: OrcCfg, log2nd: int, ref_d: int, guard_d: int, window_d: int = 0, n_threads: int = 1,
             want_mag: bool = False):
    """2-D chain, FIXED16: in_beats uint32 [n_ch, n_doppler, n_range] packed IQ; cfg = the 1-D register file with
    log2n = log2(n_range), ref_window / guard_window = range half-widths.  Returns words [n_ch, nd, nr] (, magnitudes)."""
    nr, nd = 1 << cfg.log2n, 1 << log2nd
    beats = np.ascontiguousarray(in_beats, np.uint32).reshape(-1, nd, nr)
    ch = beats.shape[0]
    out = np.zeros((ch, nd, nr), np.uint32)
    mag = np.zeros((ch, nd, nr), np.int32) if want_mag else None
    lib().orc_rd_fixed(_p(beats), ch, C.byref(cfg), log2nd, ref_d, guard_d, window_d, _p(out),
                       _p(mag) if want_mag else None, n_threads)
    return (out, mag) if want_mag else out


def tester_stim_cfg(start_value=16, ram0=0x24000000, **kw) -> OrcStimCfg:
    """The PLFG program RspChainVanillaTester.scala:86-94 writes + FixedNCOParams of RspChain.scala:94-106."""
    c = OrcStimCfg(enable=1, start_value=start_value, num_chirps=1, max_segments=4, table_size=128, table_width=16,
                   phase_width=9)
    c.segment_nums[0], c.repeated[0], c.ordinal[0], c.ram[0] = 1, 1, 0, ram0
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def plfg(cfg: OrcStimCfg, n: int) -> np.ndarray:
    v = np.zeros(n, np.int32)
    lib().orc_plfg(C.byref(cfg), n, _p(v))
    return v


def plfg_nco(cfg: OrcStimCfg, n: int) -> np.ndarray:
    b = np.zeros(n, np.uint32)
    lib().orc_plfg_nco(C.byref(cfg), n, _p(b))
    return b
