"""Parity hook for REAL Chisel simulation dumps.

For every directory under tests/golden/chisel/ that holds inputDataReal.txt, inputDataImag.txt and
outputData.txt (the files /root/reference/src/test/scala/FftMagCfarChainTester.scala:56-68,155-175
writes), + an optional config.json, the oracle (CPU) and the HIP path (GPU) must reproduce outputData.txt
word for word.  The repository ships none (the reference cannot run in this pipeline): skipped until
someone drops a directory in.  tests/golden/chisel/README.md has the recipe and the config schema."""
import glob
import json
import os

import numpy as np
import pytest

import rsp_chains_amd as R
from helpers import make_params, oracle_cfg

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "golden", "chisel")
NEEDED = ("inputDataReal.txt", "inputDataImag.txt", "outputData.txt")


def dump_dirs(root=ROOT):
    return sorted(d for d in glob.glob(os.path.join(root, "*")) if all(os.path.exists(os.path.join(d, f)) for f in NEEDED))


def load_case(d):
    """(params, rt, beats [frames, N], expected words [frames, N]) of one dump directory"""
    cfg = json.load(open(os.path.join(d, "config.json"))) if os.path.exists(os.path.join(d, "config.json")) else {}
    pk = dict(cfg.get("params", {}))
    n = int(pk.pop("numPoints", 1024))
    proto = None
    if any(k in pk for k in ("protoIn", "protoThreshold", "protoScaler")):
        bp = int(pk.get("binPoint", 12))
        proto = tuple(R.FixedPoint(*pk.pop(k, [16, bp])) for k in ("protoIn", "protoThreshold", "protoScaler"))
    params = make_params(n, bp=int(pk.pop("binPoint", 12)), alg=pk.pop("CFARAlgorithm", R.CACFARType),
                         edge=pk.pop("edgeMode", "zero"), trim=pk.pop("trimType", "Convergent"),
                         leadLagg=int(pk.pop("leadLaggWindowSize", 64)), guard=int(pk.pop("guardWindowSize", 4)),
                         proto=proto, includeCASH=bool(pk.pop("includeCASH", False)), sendCut=bool(pk.pop("sendCut", False)),
                         useBitReverse=bool(pk.pop("useBitReverse", True)), expandLogic=pk.pop("expandLogic", ()),
                         keepMSBorLSB=pk.pop("keepMSBorLSB", ()))
    assert not pk, f"{d}/config.json: unknown params keys {sorted(pk)}"
    rt = R.RunTimeRspChainParams(**{"fftSize": n, **cfg.get("runtime", {})})
    z = R.dumps.read_input_dumps(d)
    beats = R.stimulus.formAXI4StreamComplexData(z).reshape(-1, rt.fftSize)
    want = R.dumps.read_output_words(d)
    per_cell = 2 if params.cfarParams.sendCut else 1
    want = want.reshape(-1, rt.fftSize, per_cell) if per_cell == 2 else want.reshape(-1, rt.fftSize)
    assert want.shape[0] == beats.shape[0] == int(cfg.get("frames", beats.shape[0])), "frames in / out differ"
    return params, rt, beats, want


def explain(got, want, n):
    bad = np.argwhere(got != want)
    f, k = bad[0][:2]
    tg, bg, pg = R.unpack_output(np.asarray(got).reshape(got.shape[0], n, -1)[f, k, 0], n)
    tw, bw, pw = R.unpack_output(np.asarray(want).reshape(want.shape[0], n, -1)[f, k, 0], n)
    return (f"{len(bad)} of {got.size} words differ; first: frame {f} bin {k}: threshold {int(tg)} vs {int(tw)}, "
            f"bin field {int(bg)} vs {int(bw)}, peak {int(pg)} vs {int(pw)} (got vs Chisel)")


DIRS = dump_dirs()


@pytest.mark.skipif(not DIRS, reason="no Chisel dumps under tests/golden/chisel/ (see its README.md)")
@pytest.mark.parametrize("d", DIRS or ["-"], ids=lambda d: os.path.basename(d))
def test_oracle_reproduces_chisel_dumps(d):
    from oracle import oracle as O
    params, rt, beats, want = load_case(d)
    got = O.chain_fixed(beats, oracle_cfg(params, rt)).reshape(want.shape)
    assert np.array_equal(got, want), explain(got, want, rt.fftSize)


@pytest.mark.gpu
@pytest.mark.skipif(not DIRS, reason="no Chisel dumps under tests/golden/chisel/ (see its README.md)")
@pytest.mark.parametrize("d", DIRS or ["-"], ids=lambda d: os.path.basename(d))
def test_gpu_reproduces_chisel_dumps(gpu, d):
    params, rt, beats, want = load_case(d)
    with R.FftMagCfarChainVanilla(params) as dut:
        dut.configure(rt)
        got = dut.stream(beats)
    assert np.array_equal(got, want), explain(got, want, rt.fftSize)


def make_dump_dir(tmp_path, name, cfg, seed):
    """what a Chisel run would leave behind, written from the oracle's output by the dump writer"""
    from oracle import oracle as O
    d = tmp_path / name
    d.mkdir()
    json.dump(cfg, open(d / "config.json", "w"))
    n = cfg.get("runtime", {}).get("fftSize", cfg.get("params", {}).get("numPoints", 1024))
    frames = cfg.get("frames", 1)
    z = np.concatenate([R.stimulus.getComplexTones(n, 0.125, 0.25, 0.5, shiftRangeFactor=12, seed=seed + f) for f in range(frames)])
    R.dumps.write_input_dumps(str(d), z)
    params, rt, beats, _ = load_case_inputs_only(str(d), cfg)
    words = O.chain_fixed(beats, oracle_cfg(params, rt))
    R.dumps.write_output_dumps(str(d), words, n)
    return str(d)


def load_case_inputs_only(d, cfg):
    open(os.path.join(d, "outputData.txt"), "w").write("\n".join(["0000"] * (cfg.get("frames", 1) * cfg.get(
        "runtime", {}).get("fftSize", cfg.get("params", {}).get("numPoints", 1024)) * (2 if cfg.get("params", {}).get("sendCut") else 1))))
    return load_case(d)


CASES = [("tester_defaults", {}, 1),
         ("go_512_wrap", {"params": {"numPoints": 1024, "edgeMode": "wrap", "trimType": "RoundHalfUp"},
                          "runtime": {"fftSize": 512, "CFARMode": "Smallest Of", "refWindowSize": 16, "divSum": 4, "peakGrouping": 1},
                          "frames": 3}, 2),
         ("gos_sendcut", {"params": {"numPoints": 256, "CFARAlgorithm": "GOSCFARType", "sendCut": True},
                          "runtime": {"fftSize": 256, "refWindowSize": 8, "guardWindowSize": 2, "divSum": None, "indexLagg": 3, "indexLead": 5},
                          "frames": 2}, 3)]


@pytest.mark.parametrize("name,cfg,seed", CASES, ids=[c[0] for c in CASES])
def test_hook_machinery_on_oracle_made_dumps(tmp_path, name, cfg, seed):
    """The hook itself: directories written in the tester's format (from the oracle's output: NOT reference data)
    are found, parsed with their config.json and reproduced; a corrupted word is reported with its bin."""
    from oracle import oracle as O
    d = make_dump_dir(tmp_path, name, cfg, seed)
    assert dump_dirs(str(tmp_path)) == [d]
    params, rt, beats, want = load_case(d)
    got = O.chain_fixed(beats, oracle_cfg(params, rt)).reshape(want.shape)
    assert np.array_equal(got, want)
    want.reshape(-1)[7] ^= 1 << 20
    assert "first: frame 0 bin" in explain(got, want, rt.fftSize)


@pytest.mark.gpu
@pytest.mark.parametrize("name,cfg,seed", CASES, ids=[c[0] for c in CASES])
def test_hook_machinery_gpu(gpu, tmp_path, name, cfg, seed):
    d = make_dump_dir(tmp_path, name, cfg, seed)
    params, rt, beats, want = load_case(d)
    with R.FftMagCfarChainVanilla(params) as dut:
        dut.configure(rt)
        assert np.array_equal(dut.stream(beats), want)
