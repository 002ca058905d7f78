"""Synthetic stimulus and wire packing: seeded restatements of the reference's
test utilities (/root/reference/src/test/scala/RspChainTesterUtils.scala).  The
reference draws from an UNSEEDED scala.util.Random (:59), so values cannot be
reproduced; the distribution and every deterministic part are.
"""
from __future__ import annotations

import numpy as np


def getComplexTones(numSamples: int, f1r: float, f2r: float, f3r: float, shiftRangeFactor: int = 0,
                    scale: int = 1, seed: int = 1234) -> np.ndarray:
    """RspChainTesterUtils.scala:56-67: tones of amplitude 0.4/0.2/0.1 plus real-only noise
    sqrt(U1 + U2), times 2^shiftRangeFactor / scale, truncated toward zero (.toInt)."""
    shiftRange = int(2.0 ** shiftRangeFactor / scale)
    rng = np.random.default_rng(seed)
    i = np.arange(numSamples)
    noise = np.sqrt(rng.random(numSamples) + rng.random(numSamples))
    s = (noise + 0.4 * np.exp(2j * np.pi * f1r * i) + 0.2 * np.exp(2j * np.pi * f2r * i)
         + 0.1 * np.exp(2j * np.pi * f3r * i))
    return np.trunc(s.real * shiftRange) + 1j * np.trunc(s.imag * shiftRange)


def calcExpectedNcoOut(fftSize: int, binWithPeak: int) -> np.ndarray:
    """RspChainTesterUtils.scala:174-181: NCO tone, amplitude 2^14, sample index 1..N."""
    if not binWithPeak < fftSize:
        raise ValueError("requirement failed: Index of expected peak can not be larger than fft size")
    i = np.arange(1, fftSize + 1)
    ang = 2 * np.pi * binWithPeak / fftSize * i
    return np.trunc(np.cos(ang) * 2 ** 14) + 1j * np.trunc(np.sin(ang) * 2 ** 14)


def formAXI4StreamComplexData(inData, dataWidth: int = 16) -> np.ndarray:
    """RspChainTesterUtils.scala:105-109: {re[31:16], im[15:0]}, two's complement."""
    assert dataWidth == 16
    z = np.asarray(inData)
    re = np.trunc(z.real).astype(np.int64)
    im = np.trunc(z.imag).astype(np.int64)
    return (((re & 0xFFFF) << 16) | (im & 0xFFFF)).astype(np.uint32)


def formAXI4StreamRealData(inData, dataWidth: int = 16) -> np.ndarray:
    """RspChainTesterUtils.scala:96-100: data in the upper half, zeros below."""
    assert dataWidth == 16
    re = np.asarray(inData).astype(np.int64)
    return ((re & 0xFFFF) << 16).astype(np.uint32)


def jplMag(z) -> np.ndarray:
    """RspChainTesterUtils.scala:120-127 (float model, truncated)."""
    z = np.asarray(z)
    u = np.maximum(np.abs(z.real), np.abs(z.imag))
    v = np.minimum(np.abs(z.real), np.abs(z.imag))
    return np.trunc(np.maximum(u + v / 8, 7 * u / 8 + v / 2))


def chirp_frames(n_frames: int, n: int, seed: int, n_targets: int = 3, sigma: float = 0.05,
                 amps=(0.4, 0.2, 0.1)) -> np.ndarray:
    """fp32 synthetic frames of SURVEY 8(d): K point targets (complex exponentials at random
    range bins, amplitudes 0.4/0.2/0.1) + complex white noise, sigma per component."""
    rng = np.random.default_rng(seed)
    t = np.arange(n, dtype=np.float64)
    out = np.empty((n_frames, n), np.complex64)
    bins = rng.integers(0, n, size=(n_frames, n_targets))
    for f in range(n_frames):
        s = sigma * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
        for j in range(n_targets):
            s = s + amps[j % len(amps)] * np.exp(2j * np.pi * bins[f, j] * t / n)
        out[f] = s.astype(np.complex64)
    return out
