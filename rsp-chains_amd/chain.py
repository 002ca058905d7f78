"""Host-side mirror of the reference's chain interface (package `rspChain`).

The reference is a Scala/Chisel generator; no JVM exists in this pipeline, so the
host layer above the C ABI is written in Python (tests, bench) and C++
(host/RspChain.hpp), keeping the reference's names, argument meaning and error
behaviour so that a test here reads like
/root/reference/src/test/scala/FftMagCfarChainTester.scala:

    params = FftMagCfarVanillaParameters(fftParams=FFTParams.fixed(...), ...)   # Chain:77-116
    dut = FftMagCfarChainVanilla(params)                                        # Chain:119
    dut.memWriteWord(params.fftAddress.base, log2Up(fftSize))                   # Tester:82
    ...
    out = dut.stream(formAXI4StreamComplexData(inData, 16))                     # Tester:137-151

Scala `require` failures become ValueError; features the GPU path lacks raise
NotImplementedError; device errors raise RuntimeError.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

from . import _native as N

# CFARAlgorithm tags: FftMagCfarChainTester.scala:105,110,123
CACFARType, GOSCFARType, GOSCACFARType = "CACFARType", "GOSCFARType", "GOSCACFARType"
_ALG = {CACFARType: 0, GOSCFARType: 1, GOSCACFARType: 2}
_TRIM = {"RoundDown": 0, "Floor": 0, "RoundHalfUp": 1, "Convergent": 2}
_EDGE = {"zero": 0, "wrap": 1}
_WINDOW = {None: 0, "none": 0, "hann": 1, "hamming": 2, "blackman": 3}   # RSP_WINDOW_*
FIXED16, F32 = 0, 1


def log2Up(x: int) -> int:
    """chisel3.util.log2Up"""
    return max(1, (int(x) - 1).bit_length())


def isPow2(x: int) -> bool:
    return x > 0 and (x & (x - 1)) == 0


class RspError(RuntimeError):
    pass


def _check(rc: int):
    if rc == N.RSP_OK:
        return
    msg = N.lib().rsp_last_error().decode()
    if rc == N.RSP_ERR_INVALID:
        raise ValueError(f"requirement failed: {msg}")
    if rc == N.RSP_ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    if rc == N.RSP_ERR_ADDRESS:
        raise IndexError(msg)
    raise RspError(msg)


@dataclass(frozen=True)
class FixedPoint:
    """FixedPoint(width.W, binaryPoint.BP)"""
    width: int
    binaryPoint: int


@dataclass(frozen=True)
class AddressSet:
    base: int
    mask: int


@dataclass
class FFTParams:
    """fft.FFTParams as built by FFTParams.fixed(...) at FftMagCfarChain.scala:78-90."""
    dataWidth: int = 16
    twiddleWidth: int = 16
    numPoints: int = 1024
    useBitReverse: bool = True
    runTime: bool = True
    numAddPipes: int = 1
    numMulPipes: int = 1
    expandLogic: Sequence[int] = ()
    keepMSBorLSB: Sequence[bool] = ()
    minSRAMdepth: int = 1024
    binPoint: int = 12
    trimType: str = "Convergent"  # build extension (upstream default, not set by the reference)

    @staticmethod
    def fixed(**kw) -> "FFTParams":
        p = FFTParams(**kw)
        stages = log2Up(p.numPoints)
        if not p.expandLogic:
            p.expandLogic = [0] * stages
        if not p.keepMSBorLSB:
            p.keepMSBorLSB = [True] * stages
        if len(p.expandLogic) != stages or len(p.keepMSBorLSB) != stages:
            raise ValueError("requirement failed: expandLogic/keepMSBorLSB need one entry per stage")
        return p

    @property
    def protoIQ(self) -> FixedPoint:  # FftMagCfarChainTester.scala:47
        return FixedPoint(self.dataWidth, self.binPoint)


@dataclass
class MAGParams:
    """magnitude.MAGParams.fixed(...), FftMagCfarChain.scala:91-100."""
    dataWidth: int = 16
    binPoint: int = 12
    dataWidthLog: int = 16
    binPointLog: int = 9
    log2LookUpWidth: int = 9
    useLast: bool = True
    numAddPipes: int = 1
    numMulPipes: int = 1

    @staticmethod
    def fixed(**kw) -> "MAGParams":
        return MAGParams(**kw)


@dataclass
class CFARParams:
    """cfar.CFARParams(...), FftMagCfarChain.scala:101-112."""
    protoIn: FixedPoint = FixedPoint(16, 12)
    protoThreshold: FixedPoint = FixedPoint(16, 12)
    protoScaler: FixedPoint = FixedPoint(16, 12)
    leadLaggWindowSize: int = 64
    guardWindowSize: int = 4
    sendCut: bool = False
    fftSize: int = 1024
    minSubWindowSize: Optional[int] = None
    includeCASH: bool = False
    CFARAlgorithm: str = CACFARType
    numMulPipes: int = 1
    edgeMode: str = "zero"  # build extension


@dataclass
class FftMagCfarVanillaParameters:
    """rspChain.FftMagCfarVanillaParameters, FftMagCfarChain.scala:21-29 (+ GPU extensions)."""
    fftParams: FFTParams
    magParams: MAGParams
    cfarParams: CFARParams
    fftAddress: AddressSet = AddressSet(0x30000100, 0xFF)
    magAddress: AddressSet = AddressSet(0x30000200, 0xFF)
    cfarAddress: AddressSet = AddressSet(0x30002000, 0xFFF)
    beatBytes: int = 4
    dtype: int = FIXED16
    device: int = 0
    # 2-D range-Doppler chain (BASELINE.json configs 3/5; no reference counterpart): slow-time FFT
    # size (0 = the reference's 1-D chain) and the Doppler half-widths of the 2-D CA-CFAR
    dopplerPoints: int = 0
    refDoppler: int = 0
    guardDoppler: int = 0
    # pre-FFT window functions (the reference has none): over fast time / over slow time (2-D chain)
    window: Optional[str] = None
    windowDoppler: Optional[str] = None

    def to_c(self) -> N.ChainParamsC:
        p = N.ChainParamsC()
        f, m, c = self.fftParams, self.magParams, self.cfarParams
        pf = p.fftParams
        pf.dataWidth, pf.twiddleWidth, pf.numPoints = f.dataWidth, f.twiddleWidth, f.numPoints
        pf.useBitReverse, pf.runTime = int(f.useBitReverse), int(f.runTime)
        pf.numAddPipes, pf.numMulPipes = f.numAddPipes, f.numMulPipes
        for s in range(N.RSP_MAX_STAGES):
            pf.expandLogic[s] = int(f.expandLogic[s]) if s < len(f.expandLogic) else 0
            pf.keepMSBorLSB[s] = int(f.keepMSBorLSB[s]) if s < len(f.keepMSBorLSB) else 1
        pf.minSRAMdepth, pf.binPoint = f.minSRAMdepth, f.binPoint
        if f.trimType not in _TRIM:
            raise ValueError(f"requirement failed: trimType {f.trimType!r}")
        pf.trimType = _TRIM[f.trimType]
        pm = p.magParams
        pm.dataWidth, pm.binPoint, pm.dataWidthLog = m.dataWidth, m.binPoint, m.dataWidthLog
        pm.binPointLog, pm.log2LookUpWidth, pm.useLast = m.binPointLog, m.log2LookUpWidth, int(m.useLast)
        pm.numAddPipes, pm.numMulPipes = m.numAddPipes, m.numMulPipes
        pc = p.cfarParams
        for name in ("protoIn", "protoThreshold", "protoScaler"):
            fp = getattr(c, name)
            getattr(pc, name).width, getattr(pc, name).binaryPoint = fp.width, fp.binaryPoint
        pc.leadLaggWindowSize, pc.guardWindowSize = c.leadLaggWindowSize, c.guardWindowSize
        pc.sendCut, pc.fftSize = int(c.sendCut), c.fftSize
        pc.minSubWindowSize = -1 if c.minSubWindowSize is None else c.minSubWindowSize
        pc.includeCASH = int(c.includeCASH)
        if c.CFARAlgorithm not in _ALG:
            raise ValueError(f"requirement failed: CFARAlgorithm {c.CFARAlgorithm!r}")
        pc.CFARAlgorithm, pc.numMulPipes = _ALG[c.CFARAlgorithm], c.numMulPipes
        pc.edgeMode = _EDGE[c.edgeMode]
        for name in ("fftAddress", "magAddress", "cfarAddress"):
            a = getattr(self, name)
            getattr(p, name).base, getattr(p, name).mask = a.base, a.mask
        p.beatBytes, p.dtype, p.device = self.beatBytes, self.dtype, self.device
        p.dopplerPoints, p.refDoppler, p.guardDoppler = self.dopplerPoints, self.refDoppler, self.guardDoppler
        if self.window not in _WINDOW or self.windowDoppler not in _WINDOW:
            raise ValueError(f"requirement failed: window {self.window!r} / {self.windowDoppler!r}")
        p.window, p.windowDoppler = _WINDOW[self.window], _WINDOW[self.windowDoppler]
        return p


@dataclass
class RunTimeRspChainParams:
    """rspChain.RunTimeRspChainParams, RspChainVanillaTester.scala:35-62 (same defaults and requires)."""
    CFARAlgorithm: Optional[str] = "CA"
    CFARMode: str = "Greatest Of"
    refWindowSize: int = 32
    guardWindowSize: int = 4
    subWindowSize: Optional[int] = None
    fftSize: int = 1024
    thresholdScaler: float = 3.5
    divSum: Optional[int] = 5
    peakGrouping: int = 0
    indexLagg: Optional[int] = None
    indexLead: Optional[int] = None
    magMode: int = 2
    logOrLinearMode: int = 1

    def __post_init__(self):
        def require(cond, msg=""):
            if not cond:
                raise ValueError("requirement failed" + (": " + msg if msg else ""))
        require(isPow2(self.refWindowSize) and isPow2(self.fftSize))          # :50
        require(self.refWindowSize > 0 and self.guardWindowSize > 0)          # :51
        require(self.refWindowSize > self.guardWindowSize)                    # :52
        if self.subWindowSize is not None:
            require(self.subWindowSize < self.refWindowSize)                  # :54
        if self.indexLead is not None:
            require(self.indexLead < self.refWindowSize)                      # :57
        if self.indexLagg is not None:
            require(self.indexLagg < self.refWindowSize)                      # :60


_CFAR_MODE = {"Cell Averaging": 0, "Greatest Of": 1, "Smallest Of": 2, "CASH": 3}


class FftMagCfarChainVanilla:
    """GPU stand-in for `LazyModule(new FftMagCfarChainVanilla(params) with ...Pins)`
    (FftMagCfarChain.scala:31-73,119): `ioMem` becomes memWriteWord/memReadWord,
    `in`/`out` become stream()."""

    def __init__(self, params: FftMagCfarVanillaParameters):
        self.params = params
        self._lib = N.lib()
        self._h = C.c_void_p()
        cp = params.to_c()
        _check(self._lib.rsp_chain_create(C.byref(cp), C.byref(self._h)))

    # -- lifetime
    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.rsp_chain_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- control plane (AXI4MasterModel)
    def memWriteWord(self, addr: int, value: int):
        _check(self._lib.rsp_chain_write_reg(self._h, addr & 0xFFFFFFFF, int(value) & 0xFFFFFFFF))

    def memReadWord(self, addr: int) -> int:
        v = C.c_uint32()
        _check(self._lib.rsp_chain_read_reg(self._h, addr & 0xFFFFFFFF, C.byref(v)))
        return v.value

    def configure(self, rt: RunTimeRspChainParams):
        """The CSR write sequence of FftMagCfarChainTester.scala:82-132, in its order."""
        p = self.params
        bb = p.beatBytes
        self.memWriteWord(p.fftAddress.base, log2Up(rt.fftSize))                       # :82
        self.memWriteWord(p.magAddress.base, rt.magMode)                               # :84 (2 = JPL)
        binPointThr = p.cfarParams.protoThreshold.binaryPoint                          # :94-97
        base = p.cfarAddress.base
        self.memWriteWord(base, rt.fftSize)                                            # :100
        self.memWriteWord(base + bb, int(rt.thresholdScaler * 2.0 ** binPointThr))     # :101
        self.memWriteWord(base + 2 * bb, rt.logOrLinearMode)                           # :104
        alg = p.cfarParams.CFARAlgorithm
        if alg != GOSCFARType:                                                         # :105-108
            if rt.divSum is None:
                raise ValueError("requirement failed: divSum")
            self.memWriteWord(base + 3 * bb, rt.divSum)
        self.memWriteWord(base + 4 * bb, rt.peakGrouping)                              # :109
        if alg == GOSCACFARType:                                                       # :110-118
            if rt.CFARAlgorithm is None:
                raise ValueError("requirement failed: CFARAlgorithm")
            self.memWriteWord(base + 5 * bb, {"CA": 0, "GOS": 1}.get(rt.CFARAlgorithm, 0))
        self.memWriteWord(base + 6 * bb, _CFAR_MODE.get(rt.CFARMode, 0))               # :119
        self.memWriteWord(base + 7 * bb, rt.refWindowSize)                             # :120
        self.memWriteWord(base + 8 * bb, rt.guardWindowSize)                           # :121
        if alg != CACFARType:                                                          # :123-127
            if rt.indexLagg is None or rt.indexLead is None:
                raise ValueError("requirement failed: indexLagg/indexLead")
            self.memWriteWord(base + 9 * bb, rt.indexLagg)
            self.memWriteWord(base + 10 * bb, rt.indexLead)
        if alg == CACFARType and p.cfarParams.includeCASH:                             # :129-132
            if rt.subWindowSize is None:
                raise ValueError("requirement failed: subWindowSize")
            self.memWriteWord(base + 11 * bb, rt.subWindowSize)

    def check(self):
        _check(self._lib.rsp_chain_check_regs(self._h))

    # -- data plane (AXI4StreamModel)
    @property
    def fftSize(self) -> int:
        return 1 << self.memReadWord(self.params.fftAddress.base)

    @property
    def frameCells(self) -> int:
        """beats per frame: fftSize (1-D chain) or dopplerPoints x fftSize (one channel's 2-D map)"""
        return self.fftSize * (self.params.dopplerPoints or 1)

    def _as_beats(self, beats) -> np.ndarray:
        if self.params.dtype == FIXED16:
            a = np.ascontiguousarray(beats, dtype=np.uint32).ravel()
            cells = a.size
        else:
            a = np.ascontiguousarray(beats, dtype=np.complex64).ravel()
            cells = a.size
        n = self.frameCells
        if cells % n:
            raise ValueError(f"requirement failed: {cells} beats is not a whole number of {n}-beat frames "
                             "(TLAST closes every frame)")
        return a

    def stream(self, beats, out: Optional[np.ndarray] = None) -> np.ndarray:
        """Enqueue whole frames (TLAST on each frame's final beat, Tester:137) and
        collect fftSize output words per frame (Tester:145-151).  `out`: a uint32 array to receive the words
        (e.g. a HostBuffer's: pinned memory moves over the link in place, include/rspchain.h)."""
        a = self._as_beats(beats)
        n = self.frameCells
        cut = bool(self.params.cfarParams.sendCut)   # 64-bit output beat: {word, cut} per cell
        words = a.size * (2 if cut else 1)
        if out is None:
            out = np.empty(words, np.uint32)
        else:
            if out.dtype != np.uint32 or out.size != words or not out.flags.c_contiguous:
                raise ValueError(f"requirement failed: out must be a contiguous uint32 array of {words} words")
            out = out.reshape(-1)
        _check(self._lib.rsp_chain_process(self._h, a.ctypes.data_as(C.c_void_p), a.size // n,
                                           out.ctypes.data_as(C.c_void_p)))
        if self.params.dopplerPoints:
            return out.reshape(-1, self.params.dopplerPoints, self.fftSize)
        return out.reshape(-1, n, 2) if cut else out.reshape(-1, n)

    def detections(self, beats, cap: int = 1 << 20):
        """Host-buffer convenience: (sorted detection records, number of peaks found).  Complete: frames
        with more than RSP_FRAME_DET_CAP peaks are finished from the dense words on the device."""
        a = self._as_beats(beats)
        n = self.frameCells
        lst = (N.Detection * cap)()
        found = C.c_size_t()
        _check(self._lib.rsp_chain_process_detections(self._h, a.ctypes.data_as(C.c_void_p), a.size // n,
                                                      lst, cap, C.byref(found)))
        k = min(found.value, cap)
        arr = np.frombuffer(lst, dtype=np.dtype([("frame", "<u4"), ("bin", "<u4"), ("doppler", "<u4"),
                                                 ("word", "<u4")]), count=k).copy()
        return arr, found.value

    # device-resident entry points (pointers are plain integers, e.g. tensor.data_ptr())
    def process_device(self, d_in: int, n_frames: int, d_out: int):
        _check(self._lib.rsp_chain_process_device(self._h, C.c_void_p(d_in), n_frames, C.c_void_p(d_out)))

    def process_detect_device(self, d_in: int, n_frames: int, d_out: int, d_list: int, cap: int, d_count: int):
        """Fused dense words (d_out may be 0 = skip) + compact detection list.  d_count -> two device
        uint32: {peaks found, entries stored}."""
        _check(self._lib.rsp_chain_process_detect_device(self._h, C.c_void_p(d_in), n_frames,
                                                         C.c_void_p(d_out) if d_out else None,
                                                         C.c_void_p(d_list) if d_list else None, cap, C.c_void_p(d_count)))

    def detections_device(self, d_words: int, n_frames: int, d_list: int, cap: int, d_count: int):
        _check(self._lib.rsp_chain_detections_device(self._h, C.c_void_p(d_words), n_frames,
                                                     C.c_void_p(d_list), cap, C.c_void_p(d_count)))

    MAX_FRAMES_PER_LAUNCH, FORCE_TILED_CFAR2D, FORCE_GENERIC_TAIL, RD_CHUNK_BYTES, EXPERIMENT, HOST_CHUNK_BYTES = 1, 2, 3, 4, 5, 6   # RSP_OPT_* of include/rspchain.h

    def set_option(self, option: int, value: int):
        _check(self._lib.rsp_chain_set_option(self._h, option, value))

    def set_stream(self, hip_stream: int):
        _check(self._lib.rsp_chain_set_stream(self._h, C.c_void_p(hip_stream)))

    def synchronize(self):
        _check(self._lib.rsp_chain_synchronize(self._h))

    def profile_enable(self, on: bool = True):
        _check(self._lib.rsp_chain_profile_enable(self._h, int(on)))

    def profile_read(self):
        """(summed chain-kernel ms, launches) since enable / the last read."""
        ms, k = C.c_float(), C.c_uint32()
        _check(self._lib.rsp_chain_profile_read(self._h, C.byref(ms), C.byref(k)))
        return ms.value, k.value

    def timer_start(self):
        _check(self._lib.rsp_chain_timer_start(self._h))

    def timer_stop(self) -> float:
        ms = C.c_float()
        _check(self._lib.rsp_chain_timer_stop(self._h, C.byref(ms)))
        return ms.value


# ----------------------------------------------------------------- the full chain: PLFG -> NCO -> FFT -> mag -> CFAR

@dataclass
class FixedPLFGParams:
    """plfg.FixedPLFGParams, RspChain.scala:84-93"""
    maxNumOfSegments: int = 4
    maxNumOfDifferentChirps: int = 8
    maxNumOfRepeatedChirps: int = 8
    maxChirpOrdinalNum: int = 4
    maxNumOfFrames: int = 4
    maxNumOfSamplesWidth: int = 8
    outputWidthInt: int = 16
    outputWidthFrac: int = 0


@dataclass
class FixedNCOParams:
    """nco.FixedNCOParams, RspChain.scala:94-106"""
    tableSize: int = 128
    tableWidth: int = 16
    phaseWidth: int = 9
    rasterizedMode: bool = False
    nInterpolationTerms: int = 0
    ditherEnable: bool = False
    syncROMEnable: bool = False
    phaseAccEnable: bool = True
    roundingMode: str = "RoundHalfUp"
    pincType: str = "Streaming"
    poffType: str = "Fixed"


@dataclass
class RspChainVanillaParameters:
    """rspChain.RspChainVanillaParameters, RspChain.scala:24-37"""
    plfgParams: FixedPLFGParams
    ncoParams: FixedNCOParams
    fftParams: FFTParams
    magParams: MAGParams
    cfarParams: CFARParams
    plfgAddress: AddressSet = AddressSet(0x30000000, 0xFF)
    plfgRAM: AddressSet = AddressSet(0x30001000, 0xFFF)
    ncoAddress: AddressSet = AddressSet(0x30000300, 0xF)
    fftAddress: AddressSet = AddressSet(0x30000100, 0xFF)
    magAddress: AddressSet = AddressSet(0x30000200, 0xFF)
    cfarAddress: AddressSet = AddressSet(0x30002000, 0xFFF)
    beatBytes: int = 4
    device: int = 0


def _check_s(rc: int):
    if rc == N.RSP_OK:
        return
    msg = N.lib().rsp_stimulus_last_error().decode()
    if rc == N.RSP_ERR_INVALID:
        raise ValueError(f"requirement failed: {msg}")
    if rc == N.RSP_ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    if rc == N.RSP_ERR_ADDRESS:
        raise IndexError(msg)
    raise RspError(msg)


class RspChainVanilla:
    """GPU stand-in for `LazyModule(new RspChainVanilla(params) with RspChainVanillaPins)`
    (RspChain.scala:39-78): no stream input -- the PLFG/NCO pair generates the chain's beats on the
    device; `ioMem` becomes memWriteWord, `outStream` becomes run()."""

    def __init__(self, params: RspChainVanillaParameters):
        self.params = params
        lib = N.lib()
        sp = N.StimulusParamsC()
        for k in N.PlfgParamsC._fields_:
            setattr(sp.plfgParams, k[0], int(getattr(params.plfgParams, k[0])))
        n = params.ncoParams
        for k in ("tableSize", "tableWidth", "phaseWidth", "nInterpolationTerms"):
            setattr(sp.ncoParams, k, int(getattr(n, k)))
        for k in ("rasterizedMode", "ditherEnable", "syncROMEnable", "phaseAccEnable"):
            setattr(sp.ncoParams, k, int(bool(getattr(n, k))))
        sp.ncoParams.roundingMode = 0 if n.roundingMode == "RoundHalfUp" else 1
        sp.ncoParams.pincType = 0 if n.pincType == "Streaming" else 1
        sp.ncoParams.poffType = 0 if n.poffType == "Fixed" else 1
        for name in ("plfgAddress", "plfgRAM", "ncoAddress"):
            a = getattr(params, name)
            getattr(sp, name).base, getattr(sp, name).mask = a.base, a.mask
        sp.beatBytes, sp.device = params.beatBytes, params.device
        self._s = C.c_void_p()
        _check_s(lib.rsp_stimulus_create(C.byref(sp), C.byref(self._s)))
        self.chain = FftMagCfarChainVanilla(FftMagCfarVanillaParameters(
            fftParams=params.fftParams, magParams=params.magParams, cfarParams=params.cfarParams,
            fftAddress=params.fftAddress, magAddress=params.magAddress, cfarAddress=params.cfarAddress,
            beatBytes=params.beatBytes, dtype=FIXED16, device=params.device))
        self._lib = lib

    def close(self):
        if getattr(self, "_s", None):
            self._lib.rsp_stimulus_destroy(self._s)
            self._s = C.c_void_p()
        if getattr(self, "chain", None) is not None:
            self.chain.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def memWriteWord(self, addr: int, value: int):
        """one AXI4 crossbar in front of all five blocks (RspChain.scala:48-54)"""
        p = self.params
        for a in (p.plfgAddress, p.plfgRAM, p.ncoAddress):
            if (addr & ~a.mask & 0xFFFFFFFF) == a.base:
                _check_s(self._lib.rsp_stimulus_write_reg(self._s, addr & 0xFFFFFFFF, int(value) & 0xFFFFFFFF))
                return
        self.chain.memWriteWord(addr, value)

    def configure_plfg_tester(self, numFrames: int = 4, startValue: int = 16):
        """the PLFG program of RspChainVanillaTester.scala:80-94, in its order"""
        p, bb = self.params, self.params.beatBytes
        seg = 6 * bb
        rep = seg + 4 * bb
        ordn = rep + 8 * bb
        self.memWriteWord(p.plfgRAM.base, 0x24000000)                 # :86
        self.memWriteWord(p.plfgAddress.base + 2 * bb, numFrames * 2)  # :87 number of frames
        self.memWriteWord(p.plfgAddress.base + 4 * bb, 1)              # :88 number of chirps
        self.memWriteWord(p.plfgAddress.base + 5 * bb, startValue)     # :89 start value
        self.memWriteWord(p.plfgAddress.base + seg, 1)                 # :90 segments of the first chirp
        self.memWriteWord(p.plfgAddress.base + rep, 1)                 # :91 repeated chirps
        self.memWriteWord(p.plfgAddress.base + ordn, 0)                # :92
        self.memWriteWord(p.plfgAddress.base + bb, 0)                  # :93 reset bit = 0
        self.memWriteWord(p.plfgAddress.base, 1)                       # :94 enable

    def configure(self, rt: "RunTimeRspChainParams"):
        self.chain.configure(rt)   # RspChainVanillaTester.scala:96-146 == the stand-alone chain's sequence

    def stimulus(self, n_samples: int) -> np.ndarray:
        out = np.empty(n_samples, np.uint32)
        _check_s(self._lib.rsp_stimulus_generate(self._s, out.ctypes.data_as(C.c_void_p), n_samples))
        return out

    def run(self, n_frames: int = 1) -> np.ndarray:
        """collect n_frames x fftSize output words from outStream (RspChainVanillaTester.scala:157-163)"""
        n = self.chain.fftSize
        d_beats = DeviceBuffer(4 * n * n_frames, self.params.device)
        d_out = DeviceBuffer(4 * n * n_frames, self.params.device)
        try:
            self.chain.synchronize()
            _check_s(self._lib.rsp_stimulus_generate_device(self._s, C.c_void_p(d_beats.ptr), n * n_frames, None))
            # generation ran on the null stream: make it visible to the chain's stream
            import ctypes
            N.lib().rsp_chain_synchronize(self.chain._h)
            _sync_device(self.params.device)
            self.chain.process_device(d_beats.ptr, n_frames, d_out.ptr)
            self.chain.synchronize()
            return d_out.download(np.uint32, n * n_frames).reshape(n_frames, n)
        finally:
            d_beats.free()
            d_out.free()


def _sync_device(device: int):
    # a 4-byte D2H copy on the null stream orders everything queued on it before the host returns
    b = DeviceBuffer(4, device)
    b.download(np.uint32, 1)
    b.free()


# ----------------------------------------------------------------- word formats

def unpack_output(words, fftSize: int):
    """FftMagCfarChainTester.scala:153,163-167: threshold = word >> (fftBinWidth + 1) on a signed
    Int, peak = word & 1; the bits between are the bin."""
    w = np.asarray(words, np.uint32)
    bw = log2Up(fftSize)
    thr = w.astype(np.int32) >> (bw + 1)
    bins = (w >> 1) & ((1 << bw) - 1)
    return thr, bins.astype(np.int32), (w & 1).astype(np.uint8)


def unpack_output_f32(words):
    """F32 chain: fp32 threshold with the mantissa LSB carrying the peak flag."""
    w = np.asarray(words, np.uint32)
    return (w & np.uint32(0xFFFFFFFE)).view(np.float32), (w & 1).astype(np.uint8)


def device_count() -> int:
    n = C.c_int()
    rc = N.lib().rsp_device_count(C.byref(n))
    return n.value if rc == 0 else 0


class HostBuffer:
    """Pinned host memory from rsp_host_alloc, viewed as a NumPy array: stream buffers the host-buffer entry
    points DMA in place (the JVM-side counterpart wraps the same allocation in a direct ByteBuffer)."""

    def __init__(self, shape, dtype, device: int = 0):
        self.dtype = np.dtype(dtype)
        self.shape = tuple(np.atleast_1d(shape))
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        self._p = C.c_void_p()
        _check(N.lib().rsp_host_alloc(device, C.byref(self._p), self.nbytes))
        buf = (C.c_char * self.nbytes).from_address(self._p.value)
        self.array = np.frombuffer(buf, dtype=self.dtype).reshape(self.shape)

    def free(self):
        if self._p:
            self.array = None
            N.lib().rsp_host_free(self._p)
            self._p = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DeviceBuffer:
    """hipMalloc'd buffer owned through the C ABI (tests/bench without torch)."""

    def __init__(self, nbytes: int, device: int = 0):
        self.device, self.nbytes = device, nbytes
        self._p = C.c_void_p()
        _check(N.lib().rsp_device_malloc(device, C.byref(self._p), nbytes))

    @property
    def ptr(self) -> int:
        return self._p.value

    def upload(self, a: np.ndarray):
        a = np.ascontiguousarray(a)
        assert a.nbytes <= self.nbytes
        _check(N.lib().rsp_memcpy_h2d(self.device, self._p, a.ctypes.data_as(C.c_void_p), a.nbytes))

    def download(self, dtype, count: int) -> np.ndarray:
        out = np.empty(count, dtype)
        assert out.nbytes <= self.nbytes
        _check(N.lib().rsp_memcpy_d2h(self.device, out.ctypes.data_as(C.c_void_p), self._p, out.nbytes))
        return out

    def free(self):
        if self._p:
            N.lib().rsp_device_free(self.device, self._p)
            self._p = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
