#!/usr/bin/env python3
"""cfg-3 driver: n_ch x nd x nr range-Doppler maps through the 2-D chain, timing per pass."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rsp_chains_amd as R
nr = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
nd = int(sys.argv[2]) if len(sys.argv) > 2 else 512
n_ch = int(sys.argv[3]) if len(sys.argv) > 3 else 8
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
fixed = os.environ.get("RSP_PROF_FIXED") == "1"     # FIXED16 data path (int16 I/Q beats)
params = R.FftMagCfarVanillaParameters(
    fftParams=R.FFTParams.fixed(numPoints=nr), magParams=R.MAGParams.fixed(),
    cfarParams=R.CFARParams(fftSize=nr, leadLaggWindowSize=16), dtype=R.FIXED16 if fixed else R.F32, dopplerPoints=nd,
    refDoppler=8, guardDoppler=2)
rt = R.RunTimeRspChainParams(fftSize=nr, CFARMode="Cell Averaging", refWindowSize=8, guardWindowSize=2, divSum=8 if fixed else 4,
                             thresholdScaler=4.0)
dut = R.FftMagCfarChainVanilla(params); dut.configure(rt)
if os.environ.get("RSP_RD_CHUNK_MB") is not None: dut.set_option(dut.RD_CHUNK_BYTES, int(os.environ["RSP_RD_CHUNK_MB"]) << 20)
rng = np.random.default_rng(2345)
x = (0.05 * (rng.standard_normal((nd, nr)) + 1j * rng.standard_normal((nd, nr)))).astype(np.complex64)
x = np.tile(x, (n_ch, 1, 1))
if fixed:
    x = ((np.rint(x.real * 2e4).astype(np.int64) & 0xffff) << 16 | (np.rint(x.imag * 2e4).astype(np.int64) & 0xffff)).astype(np.uint32)
sets = 3
ins, outs = [], []
for s in range(sets):
    b = R.DeviceBuffer(x.nbytes); b.upload(x); ins.append(b); outs.append(R.DeviceBuffer(x.size * 4))
with_list = len(sys.argv) > 5 and sys.argv[5] == "list"   # + compaction of the dense words into a detection list
fused = len(sys.argv) > 5 and sys.argv[5] == "fused"       # the CFAR kernel appends the list itself
cap = 1 << 16
lst, cnt = R.DeviceBuffer(cap * 16), R.DeviceBuffer(8)
def step(i):
    if fused: dut.process_detect_device(ins[i % sets].ptr, n_ch, outs[i % sets].ptr, lst.ptr, cap, cnt.ptr)
    else: dut.process_device(ins[i % sets].ptr, n_ch, outs[i % sets].ptr)
    if with_list: dut.detections_device(outs[i % sets].ptr, n_ch, lst.ptr, cap, cnt.ptr)
for i in range(3): step(i)
dut.synchronize(); dut.timer_start()
for i in range(reps): step(i)
ms = dut.timer_stop() / reps
cells = x.size
if with_list or fused: print("detections {found, stored}:", cnt.download(np.uint32, 2).tolist())
print(f"rd2d nr={nr} nd={nd} ch={n_ch}: {ms*1e3:.1f} us/batch  {cells/ms/1e6:.1f} Gcells/s  {cells*28/ms/1e9:.2f} TB/s (28 B/cell algorithmic)")
