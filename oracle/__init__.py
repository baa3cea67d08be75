"""ctypes binding of the CPU oracle (oracle/zstd_oracle.c).

TEST INFRASTRUCTURE ONLY.  May be imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg — never by anything under cairo_zstd_amd/.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libzstd_oracle.so")

u8p = C.POINTER(C.c_uint8)
u64p = C.POINTER(C.c_uint64)
i32p = C.POINTER(C.c_int32)


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "zstd_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libzstd_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.czo_xxh64.restype = C.c_uint64
        L.czo_xxh64.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64]
        L.czo_decode_frame.restype = C.c_int
        L.czo_decode_frame.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, u64p]
        L.czo_decode_batch.restype = C.c_int
        L.czo_decode_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.czo_decode_single_block.restype = C.c_int
        L.czo_decode_single_block.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, u64p, u64p, C.c_size_t]
        L.czo_fd_create.restype = C.c_void_p
        L.czo_fd_destroy.argtypes = [C.c_void_p]
        for name in ("czo_fd_new", "czo_fd_reset"):
            f = getattr(L, name)
            f.restype = C.c_int
            f.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), u64p]
        L.czo_fd_content_size.restype = C.c_uint64
        L.czo_fd_content_size.argtypes = [C.c_void_p]
        L.czo_fd_checksum_from_data.restype = C.c_int
        L.czo_fd_checksum_from_data.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
        L.czo_fd_calculated_checksum.restype = C.c_uint32
        L.czo_fd_calculated_checksum.argtypes = [C.c_void_p]
        L.czo_fd_bytes_read_from_source.restype = C.c_uint64
        L.czo_fd_bytes_read_from_source.argtypes = [C.c_void_p]
        L.czo_fd_is_finished.restype = C.c_int
        L.czo_fd_is_finished.argtypes = [C.c_void_p]
        L.czo_fd_blocks_decoded.restype = C.c_size_t
        L.czo_fd_blocks_decoded.argtypes = [C.c_void_p]
        L.czo_fd_decode_blocks.restype = C.c_int
        L.czo_fd_decode_blocks.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_size_t,
                                           C.POINTER(C.c_size_t), C.POINTER(C.c_int)]
        L.czo_fd_can_collect.restype = C.c_size_t
        L.czo_fd_can_collect.argtypes = [C.c_void_p]
        L.czo_fd_collect.restype = C.c_int
        L.czo_fd_collect.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        L.czo_fd_read.restype = C.c_size_t
        L.czo_fd_read.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.czo_fd_decode_from_to.restype = C.c_int
        L.czo_fd_decode_from_to.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                            C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        L.czo_dict_decode.restype = C.c_int
        L.czo_dict_decode.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p), u64p]
        L.czo_dict_destroy.argtypes = [C.c_void_p]
        L.czo_dict_info.argtypes = [C.c_void_p, u64p]
        L.czo_fd_init_from_dict.restype = C.c_int
        L.czo_fd_init_from_dict.argtypes = [C.c_void_p, C.c_void_p]
        L.czo_kat_reverse_reads.restype = C.c_int
        L.czo_kat_reverse_reads.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, u64p, C.POINTER(C.c_int64)]
        L.czo_kat_forward_reads.restype = C.c_int
        L.czo_kat_forward_reads.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, u64p]
        L.czo_kat_fse_table.restype = C.c_int
        L.czo_kat_fse_table.argtypes = [C.c_int, C.c_uint8, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.POINTER(C.c_uint32)]
        L.czo_kat_fse_read.restype = C.c_int
        L.czo_kat_fse_read.argtypes = [C.c_void_p, C.c_size_t, C.c_uint8, C.c_void_p, C.POINTER(C.c_uint32),
                                       C.POINTER(C.c_uint8), C.POINTER(C.c_size_t)]
        L.czo_kat_huf_table.restype = C.c_int
        L.czo_kat_huf_table.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32),
                                        C.POINTER(C.c_uint32), C.c_void_p, C.POINTER(C.c_uint32)]
        L.czo_set_d1_reference_nibbles.argtypes = [C.c_int]
        L.czo_libzstd_batch.restype = C.c_long
        L.czo_libzstd_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        _lib = L
    return _lib


def _buf(b) -> np.ndarray:
    if isinstance(b, np.ndarray):
        return np.ascontiguousarray(b, dtype=np.uint8)
    return np.frombuffer(bytes(b), dtype=np.uint8)


def xxh64(data, seed: int = 0) -> int:
    a = _buf(data)
    return lib().czo_xxh64(a.ctypes.data if a.size else None, a.size, seed)


def decode_frame(src, cap: int | None = None):
    """_test_decode (src/tests/decoding.cairo:4-21).  Returns (status, out_bytes, info dict)."""
    a = _buf(src)
    if cap is None:
        cap = max(1 << 16, a.size * 64)
    out = np.empty(cap, dtype=np.uint8)
    info = (C.c_uint64 * 7)()
    st = lib().czo_decode_frame(a.ctypes.data if a.size else None, a.size, out.ctypes.data, cap, info)
    d = dict(written=info[0], consumed=info[1], checksum=info[2], has_checksum=bool(info[3]), blocks=info[4],
             window_size=info[5], content_size=info[6])
    return st, out[: info[0]].tobytes(), d


def decode_single_block(src, cap: int = 1 << 20, window: int = 1 << 17):
    a = _buf(src)
    out = np.empty(cap, dtype=np.uint8)
    w = C.c_uint64()
    c = C.c_uint64()
    st = lib().czo_decode_single_block(a.ctypes.data, a.size, out.ctypes.data, cap, C.byref(w), C.byref(c), window)
    return st, out[: w.value].tobytes(), c.value


def decode_batch(in_base: np.ndarray, in_off, in_len, out_off, out_cap, out_total: int, nthreads: int = 1,
                 out_base: np.ndarray | None = None):
    """Same argument meaning as cairo_zstd_amd.decode_batch.  Returns (out_base, out_len, status)."""
    in_base = np.ascontiguousarray(in_base, dtype=np.uint8)
    in_off = np.ascontiguousarray(in_off, dtype=np.uint64)
    in_len = np.ascontiguousarray(in_len, dtype=np.uint64)
    out_off = np.ascontiguousarray(out_off, dtype=np.uint64)
    out_cap = np.ascontiguousarray(out_cap, dtype=np.uint64)
    n = in_off.size
    if out_base is None:
        out_base = np.zeros(out_total, dtype=np.uint8)
    out_len = np.zeros(n, dtype=np.uint64)
    status = np.zeros(n, dtype=np.int32)
    lib().czo_decode_batch(in_base.ctypes.data, in_off.ctypes.data, in_len.ctypes.data, n, out_base.ctypes.data,
                           out_off.ctypes.data, out_cap.ctypes.data, out_len.ctypes.data, status.ctypes.data, nthreads)
    return out_base, out_len, status


class FrameDecoder:
    """Oracle twin of the reference FrameDecoder (src/frame_decoder.cairo:107-335)."""

    ALL, UPTO_BLOCKS, UPTO_BYTES = 0, 1, 2

    def __init__(self):
        self._h = lib().czo_fd_create()
        self._keep = None

    def __del__(self):
        if getattr(self, "_h", None):
            lib().czo_fd_destroy(self._h)
            self._h = None

    def _init(self, fn, src):
        a = _buf(src)
        consumed = C.c_size_t()
        detail = (C.c_uint64 * 2)()
        st = fn(self._h, a.ctypes.data if a.size else None, a.size, C.byref(consumed), detail)
        return st, consumed.value, (detail[0], detail[1])

    def new(self, src):
        return self._init(lib().czo_fd_new, src)

    def reset(self, src):
        return self._init(lib().czo_fd_reset, src)

    def content_size(self):
        return lib().czo_fd_content_size(self._h)

    def get_checksum_from_data(self):
        v = C.c_uint32()
        return v.value if lib().czo_fd_checksum_from_data(self._h, C.byref(v)) else None

    def get_calculated_checksum(self):
        return lib().czo_fd_calculated_checksum(self._h)

    def bytes_read_from_source(self):
        return lib().czo_fd_bytes_read_from_source(self._h)

    def is_finished(self):
        return bool(lib().czo_fd_is_finished(self._h))

    def blocks_decoded(self):
        return lib().czo_fd_blocks_decoded(self._h)

    def decode_blocks(self, src, strategy=0, n=0):
        a = _buf(src)
        consumed = C.c_size_t()
        fin = C.c_int()
        st = lib().czo_fd_decode_blocks(self._h, a.ctypes.data if a.size else None, a.size, strategy, n,
                                        C.byref(consumed), C.byref(fin))
        return st, consumed.value, bool(fin.value)

    def can_collect(self):
        return lib().czo_fd_can_collect(self._h)

    def collect(self, cap: int = 1 << 24):
        out = np.empty(cap, dtype=np.uint8)
        w = C.c_size_t()
        r = lib().czo_fd_collect(self._h, out.ctypes.data, cap, C.byref(w))
        if r <= 0:
            return None
        return out[: w.value].tobytes()

    def read(self, cap: int = 1 << 24):
        out = np.empty(cap, dtype=np.uint8)
        n = lib().czo_fd_read(self._h, out.ctypes.data, cap)
        return out[:n].tobytes()

    def decode_from_to(self, src, cap: int = 1 << 24):
        a = _buf(src)
        out = np.empty(cap, dtype=np.uint8)
        r = C.c_size_t()
        w = C.c_size_t()
        st = lib().czo_fd_decode_from_to(self._h, a.ctypes.data if a.size else None, a.size, out.ctypes.data, cap,
                                         C.byref(r), C.byref(w))
        return st, r.value, out[: w.value].tobytes()


class Dictionary:
    """DictionaryTrait::decode_dict (src/decoding/dictionary.cairo:35-91).  `status` != 0: the parse failed."""

    INFO = ("id", "content_off", "content_len", "hist0", "hist1", "hist2", "huf_max_bits", "ll_log", "ml_log", "of_log")

    def __init__(self, raw):
        self._raw = _buf(raw)
        self._h = C.c_void_p()
        detail = (C.c_uint64 * 2)()
        self.status = lib().czo_dict_decode(self._raw.ctypes.data, self._raw.size, C.byref(self._h), detail)
        self.detail = int(detail[0])
        self.info = {}
        if self.status == 0:
            v = (C.c_uint64 * 10)()
            lib().czo_dict_info(self._h, v)
            self.info = dict(zip(self.INFO, (int(x) for x in v)))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().czo_dict_destroy(self._h)
            self._h = None


def decode_frame_with_dict(src, dictionary: Dictionary, cap: int = 1 << 24):
    """new -> init_from_dict (scratch.cairo:60-65, the call the reference never makes) -> decode_blocks(All) -> collect.
    Returns (status, bytes)."""
    fd = FrameDecoder()
    st, used, _ = fd.new(src)
    if st:
        return st, b""
    st = lib().czo_fd_init_from_dict(fd._h, dictionary._h)
    if st:
        return st, b""
    a = _buf(src)
    st, _, _ = fd.decode_blocks(a[used:])
    if st:
        return st, b""
    return 0, fd.collect(cap)


def libzstd_batch(in_base: np.ndarray, in_off, in_len, out_off, out_cap, out_total: int, expected_len, nthreads: int = 1):
    """bench.py's second CPU baseline: the HOST's libzstd (ZSTD_decompress, dlopen) over the batch on `nthreads` pthreads.
    Returns the number of frames that decoded to their expected length, or -1 when the host has no libzstd.so.1."""
    in_base = np.ascontiguousarray(in_base, dtype=np.uint8)
    a = [np.ascontiguousarray(x, dtype=np.uint64) for x in (in_off, in_len, out_off, out_cap, expected_len)]
    out = np.empty(out_total, dtype=np.uint8)
    return int(lib().czo_libzstd_batch(in_base.ctypes.data, a[0].ctypes.data, a[1].ctypes.data, a[0].size, out.ctypes.data,
                                       a[2].ctypes.data, a[3].ctypes.data, a[4].ctypes.data, nthreads))
