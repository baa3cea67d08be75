"""cairo_zstd_amd — MI355X-native zstd frame/block decoder with the frame_decoder API surface of
NethermindEth/cairo_zstd.

Python here is a thin mirror over the C ABI (include/cairo_zstd_amd.h, libcairo_zstd_amd.so):
  * Context / decode_batch* : many independent frames per launch (the GPU hot path)
  * FrameDecoder            : FrameDecoderTrait call for call (src/frame_decoder.cairo:107-335)
  * read_frame_header / read_block_header : stateless parsers
All decoding runs in the HIP kernels; nothing here decodes on the CPU.
"""
from __future__ import annotations

import ctypes as C
import weakref

import numpy as np

from . import status
from ._lib import (RESULT_CHECKSUM_COMPUTED, RESULT_CHECKSUM_MATCH, RESULT_DTYPE, RESULT_FINISHED, RESULT_HAS_CHECKSUM,
                   BlockHeader, FrameHeader, build, lib)

DEBUG_CHAIN_CPP_STEP, DEBUG_NO_HUF1, DEBUG_WX_POISON, DEBUG_EXEC_FIRST = 1, 2, 4, 8     # cz_context_set_debug_flags
from .status import CzError

__all__ = ["Context", "FrameDecoder", "BlockDecodingStrategy", "decode_batch_host", "read_frame_header",
           "read_block_header", "RESULT_DTYPE", "RESULT_FINISHED", "RESULT_HAS_CHECKSUM", "RESULT_CHECKSUM_COMPUTED",
           "RESULT_CHECKSUM_MATCH", "status", "CzError", "build", "lib"]


def _as_u8(b) -> np.ndarray:
    if isinstance(b, np.ndarray):
        return np.ascontiguousarray(b, dtype=np.uint8)
    return np.frombuffer(bytes(b), dtype=np.uint8)


class Context:
    """Device context (cz_context_*).  `stream` is a raw hipStream_t (e.g.
    torch.cuda.current_stream().cuda_stream) or None for a private stream."""

    def __init__(self, device: int = 0, stream: int | None = None):
        self._h = C.c_void_p()
        st = lib().cz_context_create(C.byref(self._h), device, C.c_void_p(stream) if stream else None)
        if st:
            self._h = None
            raise CzError(st, "cz_context_create")
        self.device = device
        self._decoders = weakref.WeakSet()

    def close(self):
        if getattr(self, "_h", None):
            for fd in list(self._decoders):      # frame decoders hold device memory of this context
                fd.close()
            lib().cz_context_destroy(self._h)
            self._h = None

    __del__ = close

    def synchronize(self):
        st = lib().cz_context_synchronize(self._h)
        if st:
            raise CzError(st, f"hip error {lib().cz_context_last_hip_error(self._h)}")

    def launch_info(self):
        wg, th, cu = C.c_int(), C.c_int(), C.c_int()
        lib().cz_context_launch_info(self._h, C.byref(wg), C.byref(th), C.byref(cu))
        return dict(workgroups=wg.value, threads_per_workgroup=th.value, compute_units=cu.value)

    def execute_grid(self):
        """(waves of cz_execute_frames_kernel, waves of its 8-waves build) a batch launch runs with."""
        a, b = C.c_int(), C.c_int()
        lib().cz_context_execute_grid(self._h, C.byref(a), C.byref(b))
        return a.value, b.value

    def set_chain_arena(self, nbytes: int, min_sequences: int | None = None):
        """Enable (nbytes > 0) / disable (0) the FSE-chain pre-pass (cz_chain_kernel) for batch decodes."""
        if min_sequences is not None:
            lib().cz_context_set_chain_min_sequences(self._h, min_sequences)
        st = lib().cz_context_set_chain_arena(self._h, nbytes)
        if st:
            raise CzError(st, f"hip error {lib().cz_context_last_hip_error(self._h)}")

    def set_verify_checksum(self, on: bool = True):
        """Compute and compare the XXH64 content checksum of every checksummed frame on the device."""
        lib().cz_context_set_verify_checksum(self._h, 1 if on else 0)

    def set_dictionary(self, dictionary: "Dictionary | None"):
        """Batch decodes start every frame from `dictionary` (None: from nothing): init_from_dict, src/decoding/scratch.cairo:60-65."""
        self._dict = dictionary
        st = lib().cz_context_set_dictionary(self._h, dictionary._h if dictionary is not None else None)
        if st:
            raise CzError(st, "cz_context_set_dictionary")

    def last_chain_ms(self) -> float:
        """Milliseconds of the last launch spent in the FSE-chain pre-pass kernel (0 when it is off)."""
        ms = C.c_float(0)
        st = lib().cz_context_last_chain_ms(self._h, C.byref(ms))
        if st:
            raise CzError(st, "cz_context_last_chain_ms")
        return float(ms.value)

    def set_literal_arena(self, nbytes: int):
        """Enable (nbytes > 0) / disable (0) the block-parallel huff0 / tile kernels and the execute-only frame kernel (cz_context_set_literal_arena)."""
        st = lib().cz_context_set_literal_arena(self._h, nbytes)
        if st:
            raise CzError(st, f"hip error {lib().cz_context_last_hip_error(self._h)}")

    def last_prepass_counts(self, n: int):
        """(frames with chain records, frames with literal nodes) of the last batch launch of n frames."""
        a, b = C.c_size_t(), C.c_size_t()
        lib().cz_context_last_prepass_counts(self._h, n, C.byref(a), C.byref(b))
        return int(a.value), int(b.value)

    def last_literals_tail_ms(self) -> float:
        """Milliseconds the last launch went on with the huff0 / tile kernels after the chain kernel was done."""
        ms = C.c_float(0)
        lib().cz_context_last_literals_tail_ms(self._h, C.byref(ms))
        return float(ms.value)

    def set_exec_kernel(self, on: bool = True):
        """Frames the pre-pass finished (chain records and literals) run on cz_execute_frames_kernel (default); off: like every
        other frame, on cz_decode_frames_kernel."""
        lib().cz_context_set_exec_kernel(self._h, 1 if on else 0)

    def set_wexec_kernel(self, on: bool = True, cus: int = 0, leave_per_cu: int = 0, force: bool = False):
        """cz_wexec_kernel (a workgroup per frame, the block in hand in an LDS window) side by side with cz_execute_frames_kernel on
        batches whose far offsets outweigh the near ones (decided on the device; default on).  cus / leave_per_cu / force: the A/B
        knobs of cz_context_set_wexec_tuning (force: side by side whatever the offsets look like)."""
        lib().cz_context_set_wexec_kernel(self._h, 1 if on else 0)
        st = lib().cz_context_set_wexec_tuning(self._h, int(cus), int(leave_per_cu), int(force))
        if st:
            raise CzError(st, "cz_context_set_wexec_tuning")

    def measure_batch(self, d_in_base: int, d_in_off: int, d_in_len: int, n: int, d_out_cap: int):
        """(chain arena bytes, literal arena bytes) a batch on the device needs, from its headers (cz_context_measure_batch)."""
        a, b = C.c_size_t(), C.c_size_t()
        st = lib().cz_context_measure_batch(self._h, d_in_base, d_in_off, d_in_len, n, d_out_cap, C.byref(a), C.byref(b))
        if st:
            raise CzError(st, "cz_context_measure_batch")
        return int(a.value), int(b.value)

    def set_debug_flags(self, flags: int):
        """Test knobs (cz_context_set_debug_flags): DEBUG_CHAIN_CPP_STEP, DEBUG_NO_HUF1, DEBUG_WX_POISON."""
        lib().cz_context_set_debug_flags(self._h, int(flags))

    def debug_read_chain_arena(self, nbytes: int):
        """(first nbytes of the chain arena as a uint64 array, arena units in use) after the last launch."""
        import numpy as _np
        buf = _np.zeros(nbytes // 8, dtype=_np.uint64)
        used = C.c_uint64(0)
        st = lib().cz_context_debug_read_chain_arena(self._h, buf.ctypes.data, buf.nbytes, C.byref(used))
        if st:
            raise CzError(st, "cz_context_debug_read_chain_arena")
        return buf, int(used.value)

    def last_sequence_stats(self):
        """(near-offset, far-offset, long-run) sequence sums of the last launch (x 4), as cz_chain_kernel took them from the code tables."""
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        lib().cz_context_last_sequence_stats(self._h, C.byref(a), C.byref(b), C.byref(c))
        return int(a.value), int(b.value), int(c.value)

    def set_early_execute(self, on: bool):
        """Two chain launches (large blocks / all others) with the early execute launches behind the second, or one (default)."""
        lib().cz_context_set_early_execute(self._h, 1 if on else 0)

    def set_graph_replay(self, on: bool):
        """on: a launch that repeats the one before it is captured as a hipGraph and replayed from then on; off (default): never."""
        lib().cz_context_set_graph_replay(self._h, 1 if on else 0)

    def last_launch_was_replay(self) -> bool:
        """Whether the last batch launch was the replay of a captured graph."""
        return bool(lib().cz_context_last_launch_was_replay(self._h))

    def last_small_ms(self) -> float:
        """When the small blocks' chains and all literals of the last launch were done, ms from its start (0: not a split launch)."""
        ms = C.c_float(0)
        lib().cz_context_last_small_ms(self._h, C.byref(ms))
        return float(ms.value)

    def last_fallback_count(self) -> int:
        """Frames the pre-pass and execute kernels of the last launch handed to cz_decode_frames_kernel (each listed once)."""
        a = C.c_size_t()
        lib().cz_context_last_fallback_count(self._h, C.byref(a))
        return int(a.value)

    def last_wexec_counts(self):
        """(frames listed for cz_wexec_kernel, frames it finished, frames it handed on) in the last launch."""
        a, b, c = C.c_size_t(), C.c_size_t(), C.c_size_t()
        lib().cz_context_last_wexec_counts(self._h, C.byref(a), C.byref(b), C.byref(c))
        return int(a.value), int(b.value), int(c.value)

    def last_wexec_ms(self) -> float:
        """Milliseconds of the last launch spent in cz_wexec_kernel (0 when it did not run)."""
        ms = C.c_float(0)
        st = lib().cz_context_last_wexec_ms(self._h, C.byref(ms))
        if st:
            raise CzError(st, "cz_context_last_wexec_ms")
        return float(ms.value)

    def last_exec_ms(self) -> float:
        """Milliseconds of the last launch spent in cz_execute_frames_kernel (0 when it did not run)."""
        ms = C.c_float(0)
        st = lib().cz_context_last_exec_ms(self._h, C.byref(ms))
        if st:
            raise CzError(st, "cz_context_last_exec_ms")
        return float(ms.value)

    def last_kernel_ms(self) -> float:
        ms = C.c_float()
        st = lib().cz_context_last_kernel_ms(self._h, C.byref(ms))
        if st:
            raise CzError(st, "cz_context_last_kernel_ms")
        return ms.value

    def decode_batch_device(self, in_base: int, in_off: int, in_len: int, n: int, out_base: int, out_off: int,
                            out_cap: int, results: int):
        """All arguments are raw DEVICE pointers (tensor.data_ptr()).  Asynchronous on the context's stream — which is a stream of
        the context's own when it was created with handle 0 (torch's default stream): synchronise between torch's writes to these
        buffers and this call, and before reading the results."""
        st = lib().cz_decode_batch_device(self._h, in_base, in_off, in_len, n, out_base, out_off, out_cap, results)
        if st:
            raise CzError(st, f"hip error {lib().cz_context_last_hip_error(self._h)}")

    def decode_batch_host(self, in_base, in_off, in_len, out_off, out_cap, out_total: int):
        """Host buffers in, host buffers out (PCIe-inclusive).  Returns (out_base, results)."""
        in_base = _as_u8(in_base)
        in_off = np.ascontiguousarray(in_off, dtype=np.uint64)
        in_len = np.ascontiguousarray(in_len, dtype=np.uint64)
        out_off = np.ascontiguousarray(out_off, dtype=np.uint64)
        out_cap = np.ascontiguousarray(out_cap, dtype=np.uint64)
        n = int(in_off.size)
        out = np.zeros(max(out_total, 1), dtype=np.uint8)
        res = np.zeros(n, dtype=RESULT_DTYPE)
        st = lib().cz_decode_batch_host(self._h, in_base.ctypes.data, in_base.size, in_off.ctypes.data, in_len.ctypes.data,
                                        n, out.ctypes.data, out_total, out_off.ctypes.data, out_cap.ctypes.data,
                                        res.ctypes.data)
        if st:
            raise CzError(st, f"hip error {lib().cz_context_last_hip_error(self._h)}")
        return out, res


def decode_batch_host(frames, caps, ctx: Context | None = None):
    """Convenience: list of frame byte strings -> list of (result record, decoded bytes)."""
    own = ctx is None
    ctx = ctx or Context()
    try:
        lens = np.array([len(f) for f in frames], dtype=np.uint64)
        in_off = np.zeros(len(frames), dtype=np.uint64)
        if len(frames) > 1:
            in_off[1:] = np.cumsum(lens[:-1])
        in_base = np.frombuffer(b"".join(frames) + b"\0" * 16, dtype=np.uint8)
        caps = np.array(caps, dtype=np.uint64)
        pad = (caps + np.uint64(255)) // np.uint64(256) * np.uint64(256)
        out_off = np.zeros(len(frames), dtype=np.uint64)
        if len(frames) > 1:
            out_off[1:] = np.cumsum(pad[:-1])
        total = int(pad.sum())
        out, res = ctx.decode_batch_host(in_base, in_off, lens, out_off, caps, total)
        return [(res[i], out[int(out_off[i]): int(out_off[i]) + min(int(res[i]["bytes_produced"]), int(caps[i]))].tobytes())
                for i in range(len(frames))]
    finally:
        if own:
            ctx.close()


def partition_balanced(weights, parts: int) -> np.ndarray:
    """cz_partition_balanced: deals items to `parts` parts by weight (heaviest first, to the lightest part).  No device needed."""
    w = np.ascontiguousarray(weights, dtype=np.uint64)
    out = np.zeros(w.size, dtype=np.uint32)
    st = lib().cz_partition_balanced(w.ctypes.data if w.size else None, w.size, parts, out.ctypes.data if w.size else None)
    if st:
        raise CzError(st, "cz_partition_balanced")
    return out


def decode_batch_multi(frames, caps, ctxs):
    """cz_decode_batch_multi: one batch over several contexts (one per GPU).  Returns ([(result, bytes)], device_of)."""
    lens = np.array([len(f) for f in frames], dtype=np.uint64)
    in_off = np.zeros(len(frames), dtype=np.uint64)
    if len(frames) > 1:
        in_off[1:] = np.cumsum(lens[:-1])
    in_base = np.frombuffer(b"".join(frames) + b"\0" * 16, dtype=np.uint8)
    caps = np.array(caps, dtype=np.uint64)
    pad = (caps + np.uint64(255)) // np.uint64(256) * np.uint64(256)
    out_off = np.zeros(len(frames), dtype=np.uint64)
    if len(frames) > 1:
        out_off[1:] = np.cumsum(pad[:-1])
    total = int(pad.sum())
    out = np.zeros(total + 16, dtype=np.uint8)
    res = np.zeros(len(frames), dtype=RESULT_DTYPE)
    dev = np.zeros(len(frames), dtype=np.uint32)
    handles = (C.c_void_p * len(ctxs))(*[c._h for c in ctxs])
    st = lib().cz_decode_batch_multi(handles, len(ctxs), in_base.ctypes.data, in_base.size, in_off.ctypes.data, lens.ctypes.data, len(frames),
                                     out.ctypes.data, total, out_off.ctypes.data, caps.ctypes.data, res.ctypes.data, dev.ctypes.data)
    if st:
        raise CzError(st, "cz_decode_batch_multi")
    return [(res[i], out[int(out_off[i]): int(out_off[i]) + min(int(res[i]["bytes_produced"]), int(caps[i]))].tobytes()) for i in range(len(frames))], dev


def decode_batch_multi_device(ctxs, shares):
    """cz_decode_batch_multi_device: shares = per context (d_in_base, d_in_off, d_in_len, n, d_out_base, d_out_off, d_out_cap, d_results),
    all device pointers of that context's device.  Only enqueues."""
    from ._lib import DeviceShare
    arr = (DeviceShare * len(ctxs))(*[DeviceShare(*[int(v) for v in sh]) for sh in shares])
    handles = (C.c_void_p * len(ctxs))(*[c._h for c in ctxs])
    st = lib().cz_decode_batch_multi_device(handles, len(ctxs), arr)
    if st:
        raise CzError(st, "cz_decode_batch_multi_device")


def gather_to_root(ctxs, root, d_src, nbytes, d_dst_on_root):
    """cz_gather_to_root: concurrent peer copies of the other contexts' decoded arenas to the root context's device."""
    k = len(ctxs)
    handles = (C.c_void_p * k)(*[c._h for c in ctxs])
    src = (C.c_void_p * k)(*[int(v) for v in d_src])
    dst = (C.c_void_p * k)(*[int(v) for v in d_dst_on_root])
    nb = (C.c_size_t * k)(*[int(v) for v in nbytes])
    st = lib().cz_gather_to_root(handles, k, int(root), src, nb, dst)
    if st:
        raise CzError(st, "cz_gather_to_root")


def read_frame_header(src):
    """read_frame_header (src/frame.cairo:152-284).  Returns (status, FrameHeader, detail)."""
    a = _as_u8(src)
    fh = FrameHeader()
    detail = (C.c_uint64 * 2)()
    st = lib().cz_read_frame_header(a.ctypes.data if a.size else None, a.size, C.byref(fh), detail)
    return st, fh, (detail[0], detail[1])


def read_block_header(src):
    """BlockDecoderTrait::read_block_header (src/decoding/block_decoder.cairo:237-278)."""
    a = _as_u8(src)
    bh = BlockHeader()
    st = lib().cz_read_block_header(a.ctypes.data if a.size else None, a.size, C.byref(bh))
    return st, bh


STREAM_ENTRY_DTYPE = np.dtype([("offset", "<u8"), ("length", "<u8"), ("kind", "<u4"), ("magic", "<u4"), ("content_size", "<u8"),
                               ("out_bound", "<u8"), ("window_size", "<u8")])
STREAM_FRAME, STREAM_SKIPPABLE = 0, 1


def stream_split(src):
    """cz_stream_split: cuts a stream of concatenated zstd / skippable frames (the caller's side of SkipFrame,
    src/frame.cairo:160-166) into entries.  Returns (status, entries, consumed)."""
    a = _as_u8(src)
    cap = 64
    while True:
        ents = np.zeros(cap, dtype=STREAM_ENTRY_DTYPE)
        count, consumed = C.c_size_t(), C.c_size_t()
        st = lib().cz_stream_split(a.ctypes.data if a.size else None, a.size, ents.ctypes.data, cap, C.byref(count), C.byref(consumed))
        if st == status.CZ_E_TARGET_TOO_SMALL:
            cap = count.value
            continue
        return st, ents[: min(count.value, cap)], consumed.value


def decode_stream(src, ctx: "Context | None" = None) -> bytes:
    """Decodes every zstd frame of a multi-frame stream in ONE batch launch (skippable frames are skipped) and
    returns the concatenated output.  Raises CzError on a malformed stream or frame."""
    a = _as_u8(src)
    st, ents, consumed = stream_split(a)
    if st:
        raise CzError(st, f"cz_stream_split stopped at byte {consumed}")
    fr = ents[ents["kind"] == STREAM_FRAME]
    frames = [a[int(e["offset"]): int(e["offset"] + e["length"])] for e in fr]
    caps = [int(e["out_bound"]) + 64 for e in fr]
    out = []
    for (r, o), e in zip(decode_batch_host(frames, caps, ctx), fr):
        if int(r["status"]):
            raise CzError(int(r["status"]), f"frame at byte {int(e['offset'])}")
        out.append(o)
    return b"".join(out)


class DecoderScratch:
    """DecoderScratch (src/decoding/scratch.cairo:11-67) resident on the device."""

    def __init__(self, ctx: "Context", window_size: int = 0, _borrowed=None):
        self._ctx, self._own = ctx, _borrowed is None
        if _borrowed is not None:
            self._h = _borrowed
            return
        self._h = C.c_void_p()
        st = lib().cz_decoder_scratch_create(ctx._h, window_size, C.byref(self._h))
        if st:
            self._h = None
            raise CzError(st, "cz_decoder_scratch_create")
        ctx._decoders.add(self)

    def close(self):
        if getattr(self, "_h", None) and self._own:
            lib().cz_decoder_scratch_destroy(self._h)
        self._h = None

    __del__ = close

    def reset(self, window_size: int):
        return lib().cz_decoder_scratch_reset(self._h, window_size)

    def buffer_len(self) -> int:
        return int(lib().cz_decoder_scratch_buffer_len(self._h))

    def total_output(self) -> int:
        return int(lib().cz_decoder_scratch_total_output(self._h))

    def drain(self, cap: int = 1 << 24) -> bytes:
        out = np.empty(max(cap, 1), dtype=np.uint8)
        w = C.c_size_t()
        st = lib().cz_decoder_scratch_drain(self._h, out.ctypes.data, cap, C.byref(w))
        if st:
            raise CzError(st, "cz_decoder_scratch_drain")
        return out[: w.value].tobytes()

    def drain_to_window_size(self, cap: int = 1 << 24):
        out = np.empty(max(cap, 1), dtype=np.uint8)
        w = C.c_size_t()
        rc = lib().cz_decoder_scratch_drain_to_window_size(self._h, out.ctypes.data, cap, C.byref(w))
        if rc < 0:
            raise CzError(-rc, "cz_decoder_scratch_drain_to_window_size")
        return out[: w.value].tobytes() if rc == 1 else None

    def hash_digest(self) -> int:
        return int(lib().cz_decoder_scratch_hash_digest(self._h))

    def init_from_dict(self, dictionary: "Dictionary") -> int:
        """DecoderScratchTrait::init_from_dict (src/decoding/scratch.cairo:60-65)."""
        self._dict = dictionary                                        # keep it alive while the workspace uses it
        return lib().cz_decoder_scratch_init_from_dict(self._h, dictionary._h)


class Dictionary:
    """Dictionary / DictionaryTrait::decode_dict (src/decoding/dictionary.cairo:11-91), parsed on and resident in the device."""

    def __init__(self, ctx: "Context", raw):
        a = _as_u8(raw)
        self._ctx, self._h = ctx, C.c_void_p()
        detail = (C.c_uint64 * 2)()
        st = lib().cz_dictionary_decode(ctx._h, a.ctypes.data if a.size else None, a.size, C.byref(self._h), detail)
        if st:
            self._h = None
            raise CzError(st, f"cz_dictionary_decode (detail {int(detail[0]):#x})")
        ctx._decoders.add(self)

    def close(self):
        if getattr(self, "_h", None):
            lib().cz_dictionary_destroy(self._h)
        self._h = None

    __del__ = close

    @property
    def id(self) -> int:
        return int(lib().cz_dictionary_id(self._h))

    @property
    def content_len(self) -> int:
        return int(lib().cz_dictionary_content_len(self._h))

    @property
    def offset_hist(self):
        v = (C.c_uint32 * 3)()
        lib().cz_dictionary_offset_hist(self._h, v)
        return tuple(int(x) for x in v)


def decode_frame_with_dict(src, dictionary: Dictionary, ctx: "Context") -> bytes:
    """One frame, block by block, on a workspace seeded from `dictionary`: read_frame_header -> DecoderScratch::new ->
    init_from_dict -> {read_block_header, decode_block_content} until the last block -> drain.  (The reference's
    FrameDecoder never calls init_from_dict, so neither does cz_frame_decoder_*.)  Raises CzError."""
    a = _as_u8(src)
    st, fh, _ = read_frame_header(a)
    if st:
        raise CzError(st, "read_frame_header")
    ws = DecoderScratch(ctx, int(fh.window_size))
    try:
        st = ws.init_from_dict(dictionary)
        if st:
            raise CzError(st, "init_from_dict")
        bd, pos = BlockDecoder(), int(fh.header_len)
        while True:
            st, bh, used = bd.read_block_header(a[pos:])
            if st:
                raise CzError(st, f"read_block_header at byte {pos}")
            pos += used
            st, used = bd.decode_block_content(bh, ws, a[pos:])
            if st:
                raise CzError(st, f"decode_block_content at byte {pos}")
            pos += used
            if bh.last_block:
                return ws.drain(max(ws.buffer_len(), 1))
    finally:
        ws.close()


class BlockDecoder:
    """BlockDecoder (src/decoding/block_decoder.cairo:20-30, :69-137, :237-278)."""

    class _State(C.Structure):
        _fields_ = [("internal_state", C.c_uint8), ("header_buffer", C.c_uint8 * 3)]

    READY_FOR_HEADER, READY_FOR_BODY, FAILED = 0, 1, 2

    def __init__(self):
        self._s = BlockDecoder._State()
        lib().cz_block_decoder_new(C.byref(self._s))

    @property
    def internal_state(self) -> int:
        return int(self._s.internal_state)

    def read_block_header(self, src):
        """Returns (status, BlockHeader, consumed)."""
        a = _as_u8(src)
        bh = BlockHeader()
        used = C.c_uint8()
        st = lib().cz_block_decoder_read_block_header(C.byref(self._s), a.ctypes.data if a.size else None, a.size, C.byref(bh), C.byref(used))
        return st, bh, int(used.value)

    def decode_block_content(self, header, workspace: DecoderScratch, src):
        """Returns (status, bytes of src the block took)."""
        a = _as_u8(src)
        used = C.c_uint64()
        st = lib().cz_block_decoder_decode_block_content(C.byref(self._s), C.byref(header), workspace._h, a.ctypes.data if a.size else None, a.size, C.byref(used))
        return st, int(used.value)


class BlockDecodingStrategy:
    """src/frame_decoder.cairo:33-37"""
    ALL, UPTO_BLOCKS, UPTO_BYTES = 0, 1, 2


class FrameDecoder:
    """Mirror of FrameDecoder / FrameDecoderTrait (src/frame_decoder.cairo:17-335).  Methods
    return the status code where the reference returns Result<_, FrameDecoderError>."""

    def __init__(self, ctx: Context):
        self._ctx = ctx
        self._h = C.c_void_p()
        st = lib().cz_frame_decoder_create(ctx._h, C.byref(self._h))
        if st:
            self._h = None
            raise CzError(st, "cz_frame_decoder_create")
        ctx._decoders.add(self)

    def close(self):
        if getattr(self, "_h", None):
            lib().cz_frame_decoder_destroy(self._h)
            self._h = None

    __del__ = close

    def scratch(self) -> "DecoderScratch":
        """FrameDecoderState.decoder_scratch (src/frame_decoder.cairo:25), owned by this frame decoder."""
        return DecoderScratch(self._ctx, _borrowed=C.c_void_p(lib().cz_frame_decoder_scratch(self._h)))

    def _init(self, fn, src):
        a = _as_u8(src)
        consumed = C.c_size_t()
        detail = (C.c_uint64 * 2)()
        st = fn(self._h, a.ctypes.data if a.size else None, a.size, C.byref(consumed), detail)
        return st, consumed.value, (detail[0], detail[1])

    def new(self, src):
        return self._init(lib().cz_frame_decoder_new, src)

    def reset(self, src):
        return self._init(lib().cz_frame_decoder_reset, src)

    def content_size(self):
        return lib().cz_frame_decoder_content_size(self._h)

    def get_checksum_from_data(self):
        v = C.c_uint32()
        return v.value if lib().cz_frame_decoder_checksum_from_data(self._h, C.byref(v)) else None

    def get_calculated_checksum(self):
        return lib().cz_frame_decoder_calculated_checksum(self._h)

    def bytes_read_from_source(self):
        return lib().cz_frame_decoder_bytes_read_from_source(self._h)

    def is_finished(self):
        return bool(lib().cz_frame_decoder_is_finished(self._h))

    def blocks_decoded(self):
        return lib().cz_frame_decoder_blocks_decoded(self._h)

    def decode_blocks(self, src, strategy=BlockDecodingStrategy.ALL, n=0):
        a = _as_u8(src)
        consumed = C.c_size_t()
        fin = C.c_int()
        st = lib().cz_frame_decoder_decode_blocks(self._h, a.ctypes.data if a.size else None, a.size, strategy, n,
                                                  C.byref(consumed), C.byref(fin))
        return st, consumed.value, bool(fin.value)

    def can_collect(self):
        return lib().cz_frame_decoder_can_collect(self._h)

    def collect(self, cap: int = 1 << 24):
        out = np.empty(cap, dtype=np.uint8)
        w = C.c_size_t()
        r = lib().cz_frame_decoder_collect(self._h, out.ctypes.data, cap, C.byref(w))
        if r < 0:
            raise CzError(-r, "collect")
        return out[: w.value].tobytes() if r == 1 else None

    def read(self, cap: int = 1 << 24):
        out = np.empty(cap, dtype=np.uint8)
        n = lib().cz_frame_decoder_read(self._h, out.ctypes.data, cap)
        return out[:n].tobytes()

    def decode_from_to(self, src, cap: int = 1 << 24):
        a = _as_u8(src)
        out = np.empty(cap, dtype=np.uint8)
        r, w = C.c_size_t(), C.c_size_t()
        st = lib().cz_frame_decoder_decode_from_to(self._h, a.ctypes.data if a.size else None, a.size, out.ctypes.data, cap,
                                                   C.byref(r), C.byref(w))
        return st, r.value, out[: w.value].tobytes()
