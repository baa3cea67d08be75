import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT+"/tests")
os.environ["CAIRO_ZSTD_AMD_LIB"] = os.path.join(ROOT, "cairo_zstd_amd", "csrc", "libcairo_zstd_amd_prof.so")
import numpy as np, torch
import cairo_zstd_amd as cz
from conftest import corpus_pairs
L = C.CDLL("libzstd.so.1")
L.ZSTD_compress.restype = C.c_size_t
L.ZSTD_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
L.ZSTD_compressBound.restype = C.c_size_t; L.ZSTD_compressBound.argtypes = [C.c_size_t]
rng = np.random.default_rng(7)
words = [bytes(rng.integers(97, 123, int(rng.integers(2, 10)), dtype=np.uint8)) for _ in range(3000)]
SZ = 131072
PH = ["hdr", "huf_build", "huf_decode", "seq_tables", "ring", "chain", "extract", "lit_copy", "match", "raw_rle", "other"]
for kind in (0, 1):
    frames, origs = [], []
    for i in range(8):
        if kind == 0:
            idx = np.minimum(rng.integers(0, len(words), SZ // 4), rng.integers(0, len(words), SZ // 4))
            d = b" ".join(words[int(j)] for j in idx)[:SZ]
        else:
            rec = bytearray(); t = 0
            while len(rec) < SZ:
                t += int(rng.integers(1, 50))
                rec += b"%010d host%02d GET /api/v1/item/%06d status=%d bytes=%d\n" % (t, int(rng.integers(0, 40)), int(rng.integers(0, 50000)), (200, 200, 200, 404, 500)[int(rng.integers(0, 5))], int(rng.integers(100, 90000)))
            d = bytes(rec[:SZ])
        d = d.ljust(SZ, b".")
        cap = L.ZSTD_compressBound(SZ); dst = C.create_string_buffer(cap)
        m = L.ZSTD_compress(dst, cap, d, len(d), 3)
        frames.append(dst.raw[:m]); origs.append(d)
    ctx = cz.Context(0)
    ctx.set_chain_arena(64 << 20); ctx.set_literal_arena(32 << 20)
    for it in range(2):
        got = cz.decode_batch_host(frames, [SZ] * 8, ctx)
    buf = (C.c_uint64 * 64)()
    cz.lib().cz_context_read_profile(ctx._h, buf, 64)
    ok = all(int(r["status"]) == 0 and out == o for (r, out), o in zip(got, origs))
    vals = [buf[i] for i in range(11)]; tot = sum(vals) or 1
    print(("words" if kind == 0 else "records"), "ok", ok, "compressed", sum(len(f) for f in frames) // 8, "exec ms", round(ctx.last_exec_ms(), 3))
    print("   " + "  ".join(f"{nm} {100.0 * v / tot:.1f}%" for nm, v in zip(PH, vals) if v))
    print(f"   per frame: {buf[14] // 8} LDS-path chunks, {buf[15] // 8} general chunks with {buf[16] // 8} rounds and {buf[17] // 8} wave-wide copies")
    ctx.close()
