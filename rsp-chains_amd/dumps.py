"""Reader / writer of the reference tester's text dumps
(/root/reference/src/test/scala/FftMagCfarChainTester.scala:56-68,155-175): one value per line
as Scala's f"$x%04x" of an Int -- at least 4 hex digits, a negative Int printed as its 32-bit
two's complement (8 digits).  With these, dumps produced by the real Chisel simulation on a
machine that has a JVM can be dropped into this repo's tests to pin parity."""
from __future__ import annotations

import os

import numpy as np

FILES = ("inputDataReal.txt", "inputDataImag.txt", "outputData.txt", "thresholdData.txt")


def _fmt(v: int) -> str:
    return "%04x" % (int(v) & 0xFFFFFFFF)


def write_hex(path: str, values) -> None:
    with open(path, "w") as f:
        for v in np.asarray(values).ravel():
            f.write(_fmt(int(v)) + "\n")


def read_hex(path: str, signed_bits: int = 32) -> np.ndarray:
    vals = [int(line, 16) for line in open(path).read().split()]
    a = np.array(vals, dtype=np.int64)
    if signed_bits:
        # %04x of a non-negative Int below 2^16 is ambiguous only if it was meant as int16: the
        # tester prints Ints, so values >= 2^31 are negative Ints
        a = np.where(a >= (1 << 31), a - (1 << 32), a)
    return a


def write_input_dumps(directory: str, inData) -> None:
    """Tester:56-68: inputDataReal.txt / inputDataImag.txt from the complex stimulus."""
    z = np.asarray(inData)
    write_hex(os.path.join(directory, FILES[0]), np.trunc(z.real).astype(np.int64))
    write_hex(os.path.join(directory, FILES[1]), np.trunc(z.imag).astype(np.int64))


def read_input_dumps(directory: str) -> np.ndarray:
    re = read_hex(os.path.join(directory, FILES[0]))
    im = read_hex(os.path.join(directory, FILES[1]))
    re = ((re + (1 << 15)) % (1 << 16)) - (1 << 15)   # the stream carries the low 16 bits (Utils:21-31)
    im = ((im + (1 << 15)) % (1 << 16)) - (1 << 15)
    return re + 1j * im


def write_output_dumps(directory: str, words, fftSize: int) -> None:
    """Tester:155-175: outputData.txt (the raw words) and thresholdData.txt (word >> (log2 N + 1))."""
    w = np.asarray(words, np.uint32).ravel()
    bw = max(1, (int(fftSize) - 1).bit_length())
    write_hex(os.path.join(directory, FILES[2]), w.astype(np.int64))
    write_hex(os.path.join(directory, FILES[3]), (w.astype(np.int32) >> (bw + 1)).astype(np.int64))


def read_output_words(directory: str) -> np.ndarray:
    return (read_hex(os.path.join(directory, FILES[2])) & 0xFFFFFFFF).astype(np.uint32)
