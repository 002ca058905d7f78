"""The C-ABI library: loads without a GPU, exports every symbol include/rspchain.h declares,
struct layouts agree between the C header and the ctypes mirror, and the product never
routes through the oracle or a CPU fallback."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

import rsp_chains_amd as R
from rsp_chains_amd import _native as N

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "rspchain.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rsp_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound():
    lib = N.lib()
    names = declared_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in rspchain.h but not exported by librspchain.so"
        assert n in N.SIGNATURES, f"{n} has no ctypes signature in _native.py"
    assert set(N.SIGNATURES) == set(names)
    assert lib.rsp_abi_version() == 3   # v3: rsp_host_alloc / _free / _register / _unregister, RSP_OPT_EXPERIMENT / _HOST_CHUNK_BYTES


def test_struct_layout_matches_header(tmp_path):
    prog = tmp_path / "sz.c"
    prog.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "rspchain.h"\n'
                    'int main(void){printf("%zu %zu %zu %zu %zu %zu %zu\\n", sizeof(rsp_chain_params),'
                    'sizeof(rsp_fft_params), sizeof(rsp_mag_params), sizeof(rsp_cfar_params),'
                    'offsetof(rsp_chain_params, beatBytes), offsetof(rsp_chain_params, dtype),'
                    'sizeof(rsp_detection));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), str(prog), "-o", str(exe)], check=True)
    got = [int(v) for v in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    want = [C.sizeof(N.ChainParamsC), C.sizeof(N.FftParamsC), C.sizeof(N.MagParamsC), C.sizeof(N.CfarParamsC),
            N.ChainParamsC.beatBytes.offset, N.ChainParamsC.dtype.offset, C.sizeof(N.Detection)]
    assert got == want


def test_header_is_plain_c_and_cites_the_reference():
    src = open(HEADER).read()
    assert "torch" not in src.lower() and "#include <hip" not in src
    for cite in ("FftMagCfarChain.scala:21-29", "FftMagCfarChainTester.scala:82-132",
                 "RspChainTesterUtils.scala:105-109", "RspChainVanillaTester.scala:35-48"):
        assert cite in src
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c", HEADER], check=True)


def test_defaults_equal_reference_app_parameters():
    """rsp_chain_default_params == FftMagCfarChainVanillaApp (FftMagCfarChain.scala:77-116) == the
    Python mirror's defaults."""
    d = N.ChainParamsC()
    N.lib().rsp_chain_default_params(C.byref(d))
    assert (d.fftParams.numPoints, d.fftParams.dataWidth, d.fftParams.twiddleWidth, d.fftParams.binPoint) == (1024, 16, 16, 12)
    assert list(d.fftParams.expandLogic[:10]) == [0] * 10 and list(d.fftParams.keepMSBorLSB[:10]) == [1] * 10
    assert (d.magParams.dataWidthLog, d.magParams.binPointLog, d.magParams.log2LookUpWidth) == (16, 9, 9)
    assert (d.cfarParams.leadLaggWindowSize, d.cfarParams.guardWindowSize, d.cfarParams.sendCut) == (64, 4, 0)
    assert (d.fftAddress.base, d.magAddress.base, d.cfarAddress.base) == (0x30000100, 0x30000200, 0x30002000)
    assert d.beatBytes == 4
    py = R.FftMagCfarVanillaParameters(fftParams=R.FFTParams.fixed(numPoints=1024), magParams=R.MAGParams.fixed(),
                                       cfarParams=R.CFARParams()).to_c()
    assert bytes(py) == bytes(d)


@pytest.mark.skipif(R.device_count() > 0, reason="this box has a GPU")
def test_no_cpu_fallback_without_a_device():
    p = R.FftMagCfarVanillaParameters(fftParams=R.FFTParams.fixed(numPoints=1024), magParams=R.MAGParams.fixed(),
                                      cfarParams=R.CFARParams())
    with pytest.raises(R.RspError, match="no HIP device"):
        R.FftMagCfarChainVanilla(p)


def test_product_never_touches_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may use oracle/."""
    pkg = os.path.join(ROOT, "rsp-chains_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                # comments may cite the spec's location; code may not include, import or link it
                assert not re.search(r'#\s*include[^\n]*oracle', txt), f
                assert not re.search(r'^\s*(from|import)\s+oracle\b', txt, flags=re.M), f
                assert "librsp_oracle" not in txt and "orc_chain" not in txt, f
    out = subprocess.run(["nm", "-D", os.path.join(pkg, "librspchain.so")], capture_output=True, text=True).stdout
    assert not re.search(r"\borc_[a-z0-9_]+", out)
    assert "oracle" not in open(os.path.join(ROOT, "include", "rspchain.h")).read().lower()


def test_wire_helpers_in_library():
    lib = N.lib()
    assert lib.rsp_pack_iq(-1, 2) == 0xFFFF0002
    thr, b, p = C.c_int32(), C.c_uint32(), C.c_uint32()
    lib.rsp_unpack_word(C.c_uint32((77 << 11) | (5 << 1) | 1), 10, C.byref(thr), C.byref(b), C.byref(p))
    assert (thr.value, b.value, p.value) == (77, 5, 1)
    lib.rsp_unpack_word(C.c_uint32(((-3) << 11) & 0xFFFFFFFF), 10, C.byref(thr), C.byref(b), C.byref(p))
    assert thr.value == -3
    f = C.c_float()
    import numpy as np
    bits = int(np.float32(1.5).view(np.uint32)) | 1
    lib.rsp_unpack_word_f32(C.c_uint32(bits), C.byref(f), C.byref(p))
    assert f.value == 1.5 and p.value == 1
