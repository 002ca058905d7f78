"""MI355X-native sdf-fft -> logMagMux -> CFAR path behind the reference's chain
interface (milovanovic/rsp-chains).  The arithmetic lives in hand-written HIP
(csrc/) behind the C ABI of include/rspchain.h; this package is the host-side
mirror of the reference's parameter / register / stream API.  There is no CPU
fallback: importing works without a GPU (so configs can be built and validated),
creating a chain does not.
"""
from . import _native
from .chain import (  # noqa: F401
    AddressSet, CACFARType, CFARParams, DeviceBuffer, F32, FFTParams, FIXED16, FixedNCOParams, FixedPLFGParams,
    FftMagCfarChainVanilla, FftMagCfarVanillaParameters, FixedPoint, HostBuffer, GOSCACFARType, GOSCFARType,
    MAGParams, RspChainVanilla, RspChainVanillaParameters, RspError, RunTimeRspChainParams, device_count, isPow2, log2Up, unpack_output,
    unpack_output_f32)
from . import stimulus  # noqa: F401
from . import dist  # noqa: F401
from . import dumps  # noqa: F401

__all__ = [n for n in dir() if not n.startswith("_")]


def build(force: bool = False) -> str:
    return _native.build(force)
