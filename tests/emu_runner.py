"""Runs the kernel source on the CPU SIMT emulator (tests/emu) — sanitizer coverage for the
HIP kernels without a GPU.  Test infrastructure only."""
import os
import struct
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
EMU_DIR = os.path.join(HERE, "emu")
RESULT_DTYPE = np.dtype([("status", "<i4"), ("blocks_decoded", "<u4"), ("bytes_consumed", "<u8"),
                         ("bytes_produced", "<u8"), ("checksum_from_data", "<u4"), ("flags", "<u4"),
                         ("detail", "<u8", (2,)), ("calculated_checksum", "<u4"), ("reserved", "<u4")])


def build(target="emu_decode"):
    subprocess.check_call(["make", "-C", EMU_DIR, target], stdout=subprocess.DEVNULL)
    return os.path.join(EMU_DIR, target)


def run(frames, caps, target="emu_decode", timeout=900, chain_bytes=0, exec_kernel=False, lit_bytes=0, dict_path=None, wexec_waves=0, verify=True, wexec_auto=False, debug_flags=0):
    exe = build(target)
    with tempfile.TemporaryDirectory() as td:
        inp, outp = os.path.join(td, "in.bin"), os.path.join(td, "out.bin")
        with open(inp, "wb") as f:
            f.write(struct.pack("<Q", len(frames)))
            for fr, cap in zip(frames, caps):
                f.write(struct.pack("<QQ", len(fr), cap))
                f.write(fr)
        env = dict(os.environ, EMU_CHAIN=str(int(chain_bytes)), EMU_EXEC="1" if exec_kernel else "0", EMU_LIT=str(int(lit_bytes)), ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
        if dict_path:
            env["EMU_DICT"] = dict_path
        if wexec_waves:
            env["EMU_WEXEC"] = str(int(wexec_waves))
            if wexec_auto:
                env["EMU_WX_AUTO"] = "1"
        env["EMU_VERIFY"] = "1" if verify else "0"
        env["EMU_DEBUG_FLAGS"] = str(int(debug_flags))
        p = subprocess.run([exe, inp, outp], capture_output=True, timeout=timeout, env=env)
        run.last_stderr = p.stderr.decode()[-2000:]
        if p.returncode != 0:
            raise RuntimeError(f"emu_decode failed rc={p.returncode}\n{p.stderr.decode()[-4000:]}")
        raw = open(outp, "rb").read()
    out, pos = [], 0
    for cap in caps:
        r = np.frombuffer(raw, dtype=RESULT_DTYPE, count=1, offset=pos)[0]
        pos += RESULT_DTYPE.itemsize
        w = min(int(r["bytes_produced"]), cap)
        out.append((r, raw[pos:pos + w]))
        pos += w
    return out
