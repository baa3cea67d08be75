"""Loader of libcairo_zstd_amd.so (the C ABI of include/cairo_zstd_amd.h).

Fails loudly when the library is missing: there is no CPU decode path in this package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.environ.get("CAIRO_ZSTD_AMD_LIB") or os.path.join(CSRC, "libcairo_zstd_amd.so")   # override: diagnostic builds

RESULT_DTYPE = np.dtype([("status", "<i4"), ("blocks_decoded", "<u4"), ("bytes_consumed", "<u8"),
                         ("bytes_produced", "<u8"), ("checksum_from_data", "<u4"), ("flags", "<u4"),
                         ("detail", "<u8", (2,)), ("calculated_checksum", "<u4"), ("reserved", "<u4")])
RESULT_FINISHED, RESULT_HAS_CHECKSUM, RESULT_CHECKSUM_COMPUTED, RESULT_CHECKSUM_MATCH = 1, 2, 4, 8


class FrameHeader(C.Structure):
    _fields_ = [("descriptor", C.c_uint8), ("window_descriptor", C.c_uint8), ("has_dict_id", C.c_uint8),
                ("header_len", C.c_uint8), ("dict_id", C.c_uint32), ("frame_content_size", C.c_uint64),
                ("window_size", C.c_uint64)]


class DeviceShare(C.Structure):
    _fields_ = [("d_in_base", C.c_void_p), ("d_in_off", C.c_void_p), ("d_in_len", C.c_void_p), ("n", C.c_size_t),
                ("d_out_base", C.c_void_p), ("d_out_off", C.c_void_p), ("d_out_cap", C.c_void_p), ("d_results", C.c_void_p)]


class BlockHeader(C.Structure):
    _fields_ = [("last_block", C.c_uint8), ("block_type", C.c_uint8), ("decompressed_size", C.c_uint32),
                ("content_size", C.c_uint32)]


def build(force: bool = False) -> str:
    """Compile the library in-tree with hipcc for gfx950 (csrc/Makefile)."""
    srcs = [os.path.join(CSRC, f) for f in ("czstd_host.hip", "czstd_kernels.hip", "czstd_chain.hip", "czstd_pre.hip", "czstd_wexec.hip", "czstd_types.h")]
    srcs += [os.path.join(_HERE, "..", "include", f) for f in ("cairo_zstd_amd.h", "cairo_zstd_amd_status.h")]
    stale = not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", CSRC, "libcairo_zstd_amd.so"], stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def lib() -> C.CDLL:
    """The loaded C ABI.  Import torch first when it is used in the same process, so that both
    share one HIP runtime (the library's DT_NEEDED libamdhip64.so.7 then resolves to the copy
    torch already mapped)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  cairo_zstd_amd has no CPU fallback.")
    try:
        import torch  # noqa: F401  (maps torch's HIP runtime first)
    except Exception:  # pragma: no cover - torch is optional for the C ABI itself
        pass
    L = C.CDLL(LIB_PATH)
    vp, sz, u64p = C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64)
    L.cz_abi_version.restype = C.c_int
    L.cz_context_create.restype = C.c_int
    L.cz_context_create.argtypes = [C.POINTER(vp), C.c_int, vp]
    L.cz_context_destroy.argtypes = [vp]
    L.cz_context_synchronize.restype = C.c_int
    L.cz_context_synchronize.argtypes = [vp]
    L.cz_context_last_hip_error.restype = C.c_int
    L.cz_context_last_hip_error.argtypes = [vp]
    if hasattr(L, "cz_context_execute_grid"):                          # (scripts/ab.sh also loads diagnostic builds of earlier sources)
        L.cz_context_execute_grid.restype = C.c_int
        L.cz_context_execute_grid.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.cz_context_launch_info.restype = C.c_int
    L.cz_context_launch_info.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.cz_context_last_kernel_ms.restype = C.c_int
    L.cz_context_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float)]
    L.cz_context_last_chain_ms.restype = C.c_int
    L.cz_context_last_chain_ms.argtypes = [vp, C.POINTER(C.c_float)]
    L.cz_context_set_chain_arena.restype = C.c_int
    L.cz_context_set_chain_arena.argtypes = [vp, sz]
    L.cz_context_set_verify_checksum.restype = C.c_int
    L.cz_context_set_verify_checksum.argtypes = [vp, C.c_int]
    L.cz_context_last_exec_ms.restype = C.c_int
    L.cz_context_last_exec_ms.argtypes = [vp, C.POINTER(C.c_float)]
    L.cz_context_set_exec_kernel.restype = C.c_int
    L.cz_context_set_exec_kernel.argtypes = [vp, C.c_int]
    L.cz_context_set_wexec_kernel.restype = C.c_int
    L.cz_context_set_wexec_kernel.argtypes = [vp, C.c_int]
    L.cz_context_set_wexec_tuning.restype = C.c_int
    L.cz_context_set_wexec_tuning.argtypes = [vp, C.c_int, C.c_int, C.c_int]
    L.cz_context_measure_batch.restype = C.c_int
    L.cz_context_measure_batch.argtypes = [vp, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    L.cz_decode_batch_multi_device.restype = C.c_int
    L.cz_decode_batch_multi_device.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
    L.cz_gather_to_root.restype = C.c_int
    L.cz_gather_to_root.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
    L.cz_context_set_debug_flags.restype = C.c_int
    L.cz_context_set_debug_flags.argtypes = [vp, C.c_uint32]
    L.cz_context_debug_read_chain_arena.restype = C.c_int
    L.cz_context_debug_read_chain_arena.argtypes = [vp, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64)]
    L.cz_context_last_sequence_stats.restype = C.c_int
    L.cz_context_last_sequence_stats.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.cz_context_set_early_execute.restype = C.c_int
    L.cz_context_set_early_execute.argtypes = [vp, C.c_int]
    L.cz_context_set_graph_replay.restype = C.c_int
    L.cz_context_set_graph_replay.argtypes = [vp, C.c_int]
    L.cz_context_last_launch_was_replay.restype = C.c_int
    L.cz_context_last_launch_was_replay.argtypes = [vp]
    L.cz_context_last_small_ms.restype = C.c_int
    L.cz_context_last_small_ms.argtypes = [vp, C.POINTER(C.c_float)]
    L.cz_context_last_fallback_count.restype = C.c_int
    L.cz_context_last_fallback_count.argtypes = [vp, C.POINTER(C.c_size_t)]
    L.cz_context_last_wexec_counts.restype = C.c_int
    L.cz_context_last_wexec_counts.argtypes = [vp, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    L.cz_context_last_wexec_ms.restype = C.c_int
    L.cz_context_last_wexec_ms.argtypes = [vp, C.POINTER(C.c_float)]
    L.cz_context_last_prepass_counts.restype = C.c_int
    L.cz_context_last_prepass_counts.argtypes = [vp, sz, C.POINTER(sz), C.POINTER(sz)]
    L.cz_context_last_literals_tail_ms.restype = C.c_int
    L.cz_context_last_literals_tail_ms.argtypes = [vp, C.POINTER(C.c_float)]
    L.cz_context_set_literal_arena.restype = C.c_int
    L.cz_context_set_literal_arena.argtypes = [vp, sz]
    L.cz_context_set_chain_min_sequences.restype = C.c_int
    L.cz_context_set_chain_min_sequences.argtypes = [vp, C.c_uint32]
    L.cz_context_read_profile.restype = C.c_int
    L.cz_context_read_profile.argtypes = [vp, C.POINTER(C.c_uint64), C.c_int]
    L.cz_decode_batch_device.restype = C.c_int
    L.cz_decode_batch_device.argtypes = [vp, vp, vp, vp, sz, vp, vp, vp, vp]
    L.cz_decode_batch_host.restype = C.c_int
    L.cz_decode_batch_host.argtypes = [vp, vp, sz, vp, vp, sz, vp, sz, vp, vp, vp]
    L.cz_partition_balanced.restype = C.c_int
    L.cz_partition_balanced.argtypes = [vp, sz, sz, vp]
    L.cz_decode_batch_multi.restype = C.c_int
    L.cz_decode_batch_multi.argtypes = [vp, sz, vp, sz, vp, vp, sz, vp, sz, vp, vp, vp, vp]
    L.cz_read_frame_header.restype = C.c_int
    L.cz_read_frame_header.argtypes = [vp, sz, C.POINTER(FrameHeader), u64p]
    L.cz_read_block_header.restype = C.c_int
    L.cz_read_block_header.argtypes = [vp, sz, C.POINTER(BlockHeader)]
    L.cz_frame_decoder_create.restype = C.c_int
    L.cz_frame_decoder_create.argtypes = [vp, C.POINTER(vp)]
    L.cz_frame_decoder_destroy.argtypes = [vp]
    for name in ("cz_frame_decoder_new", "cz_frame_decoder_reset"):
        f = getattr(L, name)
        f.restype = C.c_int
        f.argtypes = [vp, vp, sz, C.POINTER(sz), u64p]
    L.cz_frame_decoder_content_size.restype = C.c_uint64
    L.cz_frame_decoder_content_size.argtypes = [vp]
    L.cz_frame_decoder_checksum_from_data.restype = C.c_int
    L.cz_frame_decoder_checksum_from_data.argtypes = [vp, C.POINTER(C.c_uint32)]
    L.cz_frame_decoder_calculated_checksum.restype = C.c_uint32
    L.cz_frame_decoder_calculated_checksum.argtypes = [vp]
    L.cz_frame_decoder_bytes_read_from_source.restype = C.c_uint64
    L.cz_frame_decoder_bytes_read_from_source.argtypes = [vp]
    L.cz_frame_decoder_is_finished.restype = C.c_int
    L.cz_frame_decoder_is_finished.argtypes = [vp]
    L.cz_frame_decoder_blocks_decoded.restype = sz
    L.cz_frame_decoder_blocks_decoded.argtypes = [vp]
    L.cz_frame_decoder_decode_blocks.restype = C.c_int
    L.cz_frame_decoder_decode_blocks.argtypes = [vp, vp, sz, C.c_int, sz, C.POINTER(sz), C.POINTER(C.c_int)]
    L.cz_frame_decoder_can_collect.restype = sz
    L.cz_frame_decoder_can_collect.argtypes = [vp]
    L.cz_frame_decoder_collect.restype = C.c_int
    L.cz_frame_decoder_collect.argtypes = [vp, vp, sz, C.POINTER(sz)]
    L.cz_frame_decoder_read.restype = sz
    L.cz_frame_decoder_read.argtypes = [vp, vp, sz]
    L.cz_frame_decoder_decode_from_to.restype = C.c_int
    L.cz_frame_decoder_decode_from_to.argtypes = [vp, vp, sz, vp, sz, C.POINTER(sz), C.POINTER(sz)]
    L.cz_stream_split.restype = C.c_int
    L.cz_stream_split.argtypes = [vp, sz, vp, sz, C.POINTER(sz), C.POINTER(sz)]
    L.cz_decoder_scratch_create.restype = C.c_int
    L.cz_decoder_scratch_create.argtypes = [vp, C.c_uint64, C.POINTER(vp)]
    L.cz_decoder_scratch_reset.restype = C.c_int
    L.cz_decoder_scratch_reset.argtypes = [vp, C.c_uint64]
    L.cz_decoder_scratch_destroy.argtypes = [vp]
    L.cz_decoder_scratch_buffer_len.restype = sz
    L.cz_decoder_scratch_buffer_len.argtypes = [vp]
    L.cz_decoder_scratch_total_output.restype = C.c_uint64
    L.cz_decoder_scratch_total_output.argtypes = [vp]
    L.cz_decoder_scratch_drain.restype = C.c_int
    L.cz_decoder_scratch_drain.argtypes = [vp, vp, sz, C.POINTER(sz)]
    L.cz_decoder_scratch_drain_to_window_size.restype = C.c_int
    L.cz_decoder_scratch_drain_to_window_size.argtypes = [vp, vp, sz, C.POINTER(sz)]
    L.cz_decoder_scratch_hash_digest.restype = C.c_uint64
    L.cz_decoder_scratch_hash_digest.argtypes = [vp]
    L.cz_block_decoder_new.argtypes = [vp]
    L.cz_block_decoder_read_block_header.restype = C.c_int
    L.cz_block_decoder_read_block_header.argtypes = [vp, vp, sz, C.POINTER(BlockHeader), C.POINTER(C.c_uint8)]
    L.cz_block_decoder_decode_block_content.restype = C.c_int
    L.cz_block_decoder_decode_block_content.argtypes = [vp, C.POINTER(BlockHeader), vp, vp, sz, u64p]
    L.cz_dictionary_decode.restype = C.c_int
    L.cz_dictionary_decode.argtypes = [vp, vp, sz, C.POINTER(vp), u64p]
    L.cz_dictionary_destroy.argtypes = [vp]
    L.cz_dictionary_id.restype = C.c_uint32
    L.cz_dictionary_id.argtypes = [vp]
    L.cz_dictionary_content_len.restype = sz
    L.cz_dictionary_content_len.argtypes = [vp]
    L.cz_dictionary_offset_hist.restype = C.c_int
    L.cz_dictionary_offset_hist.argtypes = [vp, C.POINTER(C.c_uint32)]
    L.cz_context_set_dictionary.restype = C.c_int
    L.cz_context_set_dictionary.argtypes = [vp, vp]
    L.cz_decoder_scratch_init_from_dict.restype = C.c_int
    L.cz_decoder_scratch_init_from_dict.argtypes = [vp, vp]
    L.cz_frame_decoder_scratch.restype = vp
    L.cz_frame_decoder_scratch.argtypes = [vp]
    _lib = L
    return L
