"""Loads librspchain.so (the C ABI of include/rspchain.h) through ctypes.

There is no Python or CPU fallback: if the library is missing and cannot be
built, importing this module raises.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# RSP_CHAIN_LIB: A/B-time an alternative build of the same ABI (tools/ab.sh); never a fallback
LIB_PATH = os.environ.get("RSP_CHAIN_LIB") or os.path.join(_HERE, "librspchain.so")
CSRC = os.path.join(_HERE, "csrc")

RSP_MAX_STAGES = 16
RSP_OK, RSP_ERR_INVALID, RSP_ERR_UNSUPPORTED, RSP_ERR_DEVICE, RSP_ERR_ADDRESS, RSP_ERR_NOMEM = 0, -1, -2, -3, -4, -5


class FixedProto(C.Structure):
    _fields_ = [("width", C.c_int32), ("binaryPoint", C.c_int32)]


class AddressSetC(C.Structure):
    _fields_ = [("base", C.c_uint32), ("mask", C.c_uint32)]


class FftParamsC(C.Structure):
    _fields_ = [("dataWidth", C.c_int32), ("twiddleWidth", C.c_int32), ("numPoints", C.c_int32),
                ("useBitReverse", C.c_int32), ("runTime", C.c_int32), ("numAddPipes", C.c_int32),
                ("numMulPipes", C.c_int32), ("expandLogic", C.c_int32 * RSP_MAX_STAGES),
                ("keepMSBorLSB", C.c_int32 * RSP_MAX_STAGES), ("minSRAMdepth", C.c_int32),
                ("binPoint", C.c_int32), ("trimType", C.c_int32)]


class MagParamsC(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("dataWidth", "binPoint", "dataWidthLog", "binPointLog",
                                         "log2LookUpWidth", "useLast", "numAddPipes", "numMulPipes")]


class CfarParamsC(C.Structure):
    _fields_ = [("protoIn", FixedProto), ("protoThreshold", FixedProto), ("protoScaler", FixedProto)] + [
        (n, C.c_int32) for n in ("leadLaggWindowSize", "guardWindowSize", "sendCut", "fftSize",
                                 "minSubWindowSize", "includeCASH", "CFARAlgorithm", "numMulPipes",
                                 "edgeMode")]


class ChainParamsC(C.Structure):
    _fields_ = [("fftParams", FftParamsC), ("magParams", MagParamsC), ("cfarParams", CfarParamsC),
                ("fftAddress", AddressSetC), ("magAddress", AddressSetC), ("cfarAddress", AddressSetC),
                ("beatBytes", C.c_int32), ("dtype", C.c_int32), ("device", C.c_int32),
                ("dopplerPoints", C.c_int32), ("refDoppler", C.c_int32), ("guardDoppler", C.c_int32),
                ("window", C.c_int32), ("windowDoppler", C.c_int32), ("reserved", C.c_int32 * 6)]


class PlfgParamsC(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("maxNumOfSegments", "maxNumOfDifferentChirps", "maxNumOfRepeatedChirps",
                                         "maxChirpOrdinalNum", "maxNumOfFrames", "maxNumOfSamplesWidth",
                                         "outputWidthInt", "outputWidthFrac")]


class NcoParamsC(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("tableSize", "tableWidth", "phaseWidth", "rasterizedMode",
                                         "nInterpolationTerms", "ditherEnable", "syncROMEnable", "phaseAccEnable",
                                         "roundingMode", "pincType", "poffType")]


class StimulusParamsC(C.Structure):
    _fields_ = [("plfgParams", PlfgParamsC), ("ncoParams", NcoParamsC), ("plfgAddress", AddressSetC),
                ("plfgRAM", AddressSetC), ("ncoAddress", AddressSetC), ("beatBytes", C.c_int32),
                ("device", C.c_int32)]


class Detection(C.Structure):
    _fields_ = [("frame", C.c_uint32), ("bin", C.c_uint32), ("doppler", C.c_uint32), ("word", C.c_uint32)]


# every symbol include/rspchain.h declares: (restype, argtypes)
_P = C.POINTER
SIGNATURES = {
    "rsp_abi_version": (C.c_uint32, []),
    "rsp_last_error": (C.c_char_p, []),
    "rsp_chain_default_params": (None, [_P(ChainParamsC)]),
    "rsp_chain_validate_params": (C.c_int, [_P(ChainParamsC)]),
    "rsp_chain_create": (C.c_int, [_P(ChainParamsC), _P(C.c_void_p)]),
    "rsp_chain_destroy": (None, [C.c_void_p]),
    "rsp_chain_write_reg": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32]),
    "rsp_chain_read_reg": (C.c_int, [C.c_void_p, C.c_uint32, _P(C.c_uint32)]),
    "rsp_chain_check_regs": (C.c_int, [C.c_void_p]),
    "rsp_chain_process": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "rsp_chain_process_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "rsp_chain_process_detect_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]),
    "rsp_chain_detections_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_uint32, C.c_void_p]),
    "rsp_chain_process_detections": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, _P(C.c_size_t)]),
    "rsp_chain_set_option": (C.c_int, [C.c_void_p, C.c_int, C.c_int64]),
    "rsp_chain_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rsp_chain_synchronize": (C.c_int, [C.c_void_p]),
    "rsp_chain_timer_start": (C.c_int, [C.c_void_p]),
    "rsp_chain_timer_stop": (C.c_int, [C.c_void_p, _P(C.c_float)]),
    "rsp_chain_profile_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "rsp_chain_profile_read": (C.c_int, [C.c_void_p, _P(C.c_float), _P(C.c_uint32)]),
    "rsp_device_count": (C.c_int, [_P(C.c_int)]),
    "rsp_device_malloc": (C.c_int, [C.c_int, _P(C.c_void_p), C.c_size_t]),
    "rsp_device_free": (C.c_int, [C.c_int, C.c_void_p]),
    "rsp_host_alloc": (C.c_int, [C.c_int, _P(C.c_void_p), C.c_size_t]),
    "rsp_host_free": (C.c_int, [C.c_void_p]),
    "rsp_host_register": (C.c_int, [C.c_int, C.c_void_p, C.c_size_t]),
    "rsp_host_unregister": (C.c_int, [C.c_void_p]),
    "rsp_memcpy_h2d": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]),
    "rsp_memcpy_d2h": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]),
    "rsp_stimulus_default_params": (None, [_P(StimulusParamsC)]),
    "rsp_stimulus_create": (C.c_int, [_P(StimulusParamsC), _P(C.c_void_p)]),
    "rsp_stimulus_destroy": (None, [C.c_void_p]),
    "rsp_stimulus_write_reg": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32]),
    "rsp_stimulus_read_reg": (C.c_int, [C.c_void_p, C.c_uint32, _P(C.c_uint32)]),
    "rsp_stimulus_generate_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "rsp_stimulus_generate": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "rsp_stimulus_last_error": (C.c_char_p, []),
    "rsp_pack_iq": (C.c_uint32, [C.c_int32, C.c_int32]),
    "rsp_unpack_word": (None, [C.c_uint32, C.c_int32, _P(C.c_int32), _P(C.c_uint32), _P(C.c_uint32)]),
    "rsp_unpack_word_f32": (None, [C.c_uint32, _P(C.c_float), _P(C.c_uint32)]),
}


def build(force: bool = False) -> str:
    """hipcc --offload-arch=gfx950 build of the library (csrc/Makefile)."""
    if force or not os.path.exists(LIB_PATH) or _stale():
        subprocess.run(["make", "-C", CSRC, "-j4"] + (["-B"] if force else []), check=True,
                       stdout=subprocess.DEVNULL)
    return LIB_PATH


def _stale() -> bool:
    if os.environ.get("RSP_CHAIN_LIB") or not os.path.isdir(CSRC):
        return False
    t = os.path.getmtime(LIB_PATH)
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp", ".hpp"))]
    srcs.append(os.path.join(_HERE, "..", "include", "rspchain.h"))
    return any(os.path.exists(s) and os.path.getmtime(s) > t for s in srcs)


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    try:
        # torch bundles its own libamdhip64 (same SONAME); loading it first makes
        # this library and torch share ONE HIP runtime, so torch streams and
        # tensors can be handed across the C ABI.
        import torch  # noqa: F401
    except Exception:
        pass
    if not os.path.exists(LIB_PATH) or _stale():
        try:
            build()
        except Exception as e:  # pragma: no cover
            if not os.path.exists(LIB_PATH):
                raise ImportError(
                    f"librspchain.so is missing and could not be built ({e}); "
                    "the rsp-chains GPU path has no fallback") from e
    L = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(L, name)  # AttributeError = ABI mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L
