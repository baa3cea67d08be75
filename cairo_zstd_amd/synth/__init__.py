"""ctypes binding of the synthetic frame generator (synth.c): benchmark / parity inputs."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libczsynth.so")

RAW_RLE, HUF_LITERALS, FULL_4A, FULL_4B, MIX = 2, 3, 4, 41, 5
KINDS = {"raw_rle": RAW_RLE, "huf_literals": HUF_LITERALS, "full_4a": FULL_4A, "full_4b": FULL_4B, "mix": MIX}
SEEDS = {RAW_RLE: 0x5EED0002, HUF_LITERALS: 0x5EED0003, FULL_4A: 0x5EED0004, FULL_4B: 0x5EED0004, MIX: 0x5EED0005}

_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "synth.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libczsynth.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.czs_slot_bytes.restype = C.c_size_t
        L.czs_slot_bytes.argtypes = [C.c_int]
        L.czs_generate.restype = C.c_int
        L.czs_generate.argtypes = [C.c_int, C.c_uint64, C.c_uint64, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p,
                                   C.c_void_p, C.c_int]
        _lib = L
    return _lib


class Batch:
    """A batch of frames in fixed-stride slots: frame i = base[off[i] : off[i]+length[i]]."""

    def __init__(self, base, off, length, regen):
        self.base, self.off, self.length, self.regen = base, off, length, regen

    @property
    def n(self):
        return int(self.off.size)

    def frame(self, i: int) -> bytes:
        return self.base[int(self.off[i]): int(self.off[i] + self.length[i])].tobytes()

    def out_layout(self, align: int = 256):
        """Output offsets / capacities: every frame gets its regenerated size, aligned."""
        cap = self.regen.astype(np.uint64)
        padded = (cap + np.uint64(align - 1)) // np.uint64(align) * np.uint64(align)
        off = np.zeros(self.n, dtype=np.uint64)
        if self.n > 1:
            off[1:] = np.cumsum(padded[:-1])
        total = int(padded.sum())
        return off, cap, total

    def compact(self) -> "Batch":
        """Copy into a dense arena (frames back to back)."""
        off = np.zeros(self.n, dtype=np.uint64)
        if self.n > 1:
            off[1:] = np.cumsum(self.length[:-1])
        total = int(self.length.sum())
        base = np.empty(total + 64, dtype=np.uint8)
        for i in range(self.n):
            base[int(off[i]): int(off[i] + self.length[i])] = self.base[int(self.off[i]): int(self.off[i] + self.length[i])]
        base[total:] = 0
        return Batch(base, off, self.length.copy(), self.regen.copy())


def generate(kind, n: int, seed: int | None = None, first_index: int = 0, nthreads: int | None = None) -> Batch:
    if isinstance(kind, str):
        kind = KINDS[kind]
    if seed is None:
        seed = SEEDS[kind]
    if nthreads is None:
        nthreads = min(32, os.cpu_count() or 1)
    L = lib()
    stride = int(L.czs_slot_bytes(kind))
    base = np.zeros(n * stride + 64, dtype=np.uint8)
    length = np.zeros(n, dtype=np.uint64)
    regen = np.zeros(n, dtype=np.uint64)
    fails = L.czs_generate(kind, seed, first_index, n, base.ctypes.data, stride, length.ctypes.data, regen.ctypes.data,
                           nthreads)
    if fails:
        raise RuntimeError(f"synth: {fails} frames fell back to the placeholder")
    off = (np.arange(n, dtype=np.uint64) * np.uint64(stride)).astype(np.uint64)
    return Batch(base, off, length, regen)
