"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle and the committed
golden corpus.  Run on the MI355X box with `pytest -m gpu`."""
import hashlib
import json
import os

import numpy as np
import pytest

import oracle
from conftest import GOLDEN, add_checksum, corpus_pairs, raw_frame_with_checksum

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cz():
    import torch  # noqa: F401
    import cairo_zstd_amd as m
    assert os.path.exists(m._lib.LIB_PATH), "libcairo_zstd_amd.so missing: run __graft_entry__.build()"
    return m


@pytest.fixture(scope="module")
def ctx(cz):
    c = cz.Context(0)
    yield c
    c.close()


def _diff(a: bytes, b: bytes):
    if len(a) != len(b):
        return f"len {len(a)} != {len(b)}"
    x, y = np.frombuffer(a, np.uint8), np.frombuffer(b, np.uint8)
    bad = np.nonzero(x != y)[0]
    return None if bad.size == 0 else f"{bad.size} bytes differ, first at {int(bad[0])}"


def _check_against_oracle(cz, ctx, frames, caps, expect_ok=False, label=""):
    got = cz.decode_batch_host(frames, caps, ctx)
    bad = []
    for i, (fr, cap, (r, out)) in enumerate(zip(frames, caps, got)):
        st, ref, info = oracle.decode_frame(fr, cap=cap)
        if expect_ok:
            assert st == 0, f"{label}[{i}] oracle status {st}"
        if int(r["status"]) != st:
            bad.append(f"{label}[{i}] status gpu={cz.status.name(r['status'])} oracle={cz.status.name(st)} detail={r['detail']}")
            continue
        if st == 0:
            d = _diff(out, ref)
            if d:
                bad.append(f"{label}[{i}] output {d}")
            elif int(r["bytes_consumed"]) != info["consumed"] or int(r["blocks_decoded"]) != info["blocks"]:
                bad.append(f"{label}[{i}] consumed/blocks {r['bytes_consumed']}/{r['blocks_decoded']} vs {info['consumed']}/{info['blocks']}")
            elif bool(r["flags"] & 2) != info["has_checksum"] or (info["has_checksum"] and int(r["checksum_from_data"]) != info["checksum"]):
                bad.append(f"{label}[{i}] checksum field")
    assert not bad, "\n".join(bad[:20]) + f"\n({len(bad)} of {len(frames)} frames differ)"
    return got


def test_corpus_golden_batch(cz, ctx):
    """_test_decode (src/tests/decoding.cairo:4-21) for every committed corpus pair, on the GPU."""
    pairs = corpus_pairs()
    got = cz.decode_batch_host([z for _, z, _ in pairs], [len(o) + 32 for _, _, o in pairs], ctx)
    for (name, z, orig), (r, out) in zip(pairs, got):
        assert int(r["status"]) == 0, f"{name}: {cz.status.name(r['status'])} {r['detail']}"
        assert _diff(out, orig) is None, f"{name}: {_diff(out, orig)}"
        assert int(r["bytes_consumed"]) == len(z)
        assert r["flags"] & 1 and r["flags"] & 2
        assert int(r["checksum_from_data"]) == oracle.xxh64(orig) & 0xFFFFFFFF, name


def test_content_checksum_on_device(cz, ctx):
    """get_calculated_checksum == get_checksum_from_data (src/tests/decoding.cairo:16-19) computed
    by the decode kernel: the corpus (every frame carries one), synthetic frames patched to carry
    one, every tail-length class, and a corrupted stored value."""
    from cairo_zstd_amd import synth
    pairs = corpus_pairs()
    frames = [z for _, z, _ in pairs]
    origs = [o for _, _, o in pairs]
    b = synth.generate("mix", 300, first_index=77)
    for i in range(b.n):
        st, ref, _ = oracle.decode_frame(b.frame(i), cap=int(b.regen[i]))
        assert st == 0
        frames.append(add_checksum(b.frame(i), ref))
        origs.append(ref)
    rng = np.random.default_rng(3)
    for n in [0, 1, 4, 7, 8, 31, 32, 33, 511, 512, 513, 544, 1025, 131072, 300000]:
        d = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        frames.append(raw_frame_with_checksum(d))
        origs.append(d)
    bad_at = len(frames)
    frames.append(raw_frame_with_checksum(origs[0], corrupt=True))
    origs.append(origs[0])
    ctx.set_verify_checksum(True)
    try:
        got = cz.decode_batch_host(frames, [len(o) + 8 for o in origs], ctx)
    finally:
        ctx.set_verify_checksum(False)
    for i, ((r, out), o) in enumerate(zip(got, origs)):
        assert int(r["status"]) == 0 and out == o, i
        assert r["flags"] & cz.RESULT_CHECKSUM_COMPUTED, i
        assert int(r["calculated_checksum"]) == oracle.xxh64(o) & 0xFFFFFFFF, i
        assert bool(r["flags"] & cz.RESULT_CHECKSUM_MATCH) == (i != bad_at), i
    # off again: the flag must not be set
    got = cz.decode_batch_host(frames[:4], [len(o) + 8 for o in origs[:4]], ctx)
    assert all(not (r["flags"] & cz.RESULT_CHECKSUM_COMPUTED) for r, _ in got)


@pytest.mark.parametrize("kind,n", [("raw_rle", 64), ("huf_literals", 48), ("full_4a", 24), ("full_4b", 8), ("mix", 1500)])
def test_synthetic_vs_oracle(cz, ctx, kind, n):
    from cairo_zstd_amd import synth
    b = synth.generate(kind, n)
    frames = [b.frame(i) for i in range(n)]
    _check_against_oracle(cz, ctx, frames, [int(r) + 16 for r in b.regen], expect_ok=True, label=kind)


def test_output_capacity_exact_and_too_small(cz, ctx):
    from cairo_zstd_amd import synth
    b = synth.generate("mix", 200, first_index=5000)
    frames = [b.frame(i) for i in range(b.n)]
    exact = cz.decode_batch_host(frames, [int(r) for r in b.regen], ctx)
    for i, (r, out) in enumerate(exact):
        assert int(r["status"]) == 0 and len(out) == int(b.regen[i]), i
    short = cz.decode_batch_host(frames, [max(int(r) - 1, 0) for r in b.regen], ctx)
    for i, (r, _) in enumerate(short):
        if int(b.regen[i]) > 0:
            assert int(r["status"]) == cz.status.CZ_E_OUTPUT_TOO_SMALL, (i, cz.status.name(r["status"]))


def _mutations(z: bytes, seed: int):
    rng = np.random.default_rng(seed)
    out = [z[: len(z) // 2], z[:-1], z[:5], z[:3], b"", z + b"\x00"]
    for _ in range(10):
        a = bytearray(z)
        k = int(rng.integers(0, len(a)))
        a[k] ^= 1 << int(rng.integers(0, 8))
        out.append(bytes(a))
    return out


@pytest.mark.parametrize("pipeline", ["single_kernel", "prepass"])
def test_malformed_inputs_status_parity(cz, ctx, pipeline):
    """Truncated / bit-flipped frames: same status as the oracle, never a fault, and a failing
    frame leaves its neighbours intact — through the one complete kernel, and through the pre-pass kernels (scan, chain,
    huff0, tile, execute), which must hand every irregular frame to the complete kernel for the reference's status."""
    pairs = corpus_pairs(max_orig=20000)
    frames, caps = [], []
    for idx, (name, z, orig) in enumerate(pairs):
        for m in _mutations(z, idx):
            frames.append(m)
            caps.append(len(orig) * 2 + 4096)
    if pipeline == "prepass":
        c2 = cz.Context(0)
        c2.set_chain_arena(128 << 20, min_sequences=0)
        c2.set_literal_arena(64 << 20)
        try:
            got = cz.decode_batch_host(frames, caps, c2)
            assert c2.last_chain_ms() > 0.0
        finally:
            c2.close()
    else:
        got = cz.decode_batch_host(frames, caps, ctx)
    mismatch = []
    for i, (fr, cap, (r, out)) in enumerate(zip(frames, caps, got)):
        st, ref, info = oracle.decode_frame(fr, cap=cap)
        if st != int(r["status"]):
            mismatch.append((i, cz.status.name(r["status"]), cz.status.name(st)))
        elif st == 0:
            assert _diff(out, ref) is None, i
    assert not mismatch, f"{len(mismatch)} of {len(frames)}: {mismatch[:15]}"


def test_frame_decoder_api_matches_oracle(cz, ctx):
    """FrameDecoder mirror (src/frame_decoder.cairo:107-335): UptoBlocks stepping + collect."""
    pairs = corpus_pairs()
    for name, z, orig in pairs[::4]:
        fd, od = cz.FrameDecoder(ctx), oracle.FrameDecoder()
        st, hl, _ = fd.new(z)
        ost, ohl, _ = od.new(z)
        assert (st, hl) == (ost, ohl) == (0, ohl)
        assert fd.content_size() == od.content_size()
        pos, out = hl, b""
        while not fd.is_finished():
            st, used, fin = fd.decode_blocks(z[pos:], cz.BlockDecodingStrategy.UPTO_BLOCKS, 2)
            ost, oused, ofin = od.decode_blocks(z[pos:], oracle.FrameDecoder.UPTO_BLOCKS, 2)
            assert (st, used, fin) == (ost, oused, ofin), name
            assert st == 0
            pos += used
            assert fd.blocks_decoded() == od.blocks_decoded() and fd.can_collect() == od.can_collect()
            a, b = fd.collect(cap=len(orig) + 64), od.collect(cap=len(orig) + 64)
            assert a == b, name
            out += a or b""
        assert out == orig, name
        assert fd.bytes_read_from_source() == od.bytes_read_from_source() == len(z)
        assert fd.get_checksum_from_data() == od.get_checksum_from_data() == fd.get_calculated_checksum()
        fd.close()


def test_decode_from_to_streaming_matches_oracle(cz, ctx):
    pairs = corpus_pairs()
    for name, z, orig in pairs[1::6]:
        fd, od = cz.FrameDecoder(ctx), oracle.FrameDecoder()
        st, hl, _ = fd.new(z)
        od.new(z)
        pos, out, guard, step = hl, b"", 0, 3000
        while not fd.is_finished() and guard < 5000:
            guard += 1
            chunk = z[pos:pos + step]
            st, r, got = fd.decode_from_to(chunk, cap=len(orig) + 64)
            ost, orr, ogot = od.decode_from_to(chunk, cap=len(orig) + 64)
            assert (st, r, got) == (ost, orr, ogot), (name, guard)
            assert st == 0
            if r == 0 and not got:
                assert len(chunk) < len(z) - pos, name      # a whole block did not fit: feed more
                step *= 2
            pos += r
            out += got
        st, r, got = fd.decode_from_to(b"", cap=len(orig) + 64)
        out += got
        assert out == orig, name
        fd.close()


def test_frame_header_errors(cz, ctx):
    fd = cz.FrameDecoder(ctx)
    skip = bytes.fromhex("502a4d18") + (7).to_bytes(4, "little") + b"\0" * 7
    st, _, detail = fd.new(skip)
    assert st == cz.status.CZ_E_FH_SKIP_FRAME and detail == (0x184D2A50, 7)   # frame.cairo:160-166
    assert fd.new(b"\x01\x02\x03")[0] == cz.status.CZ_E_FH_MAGIC_READ
    assert fd.new(b"\x01\x02\x03\x04\x05")[0] == cz.status.CZ_E_FH_BAD_MAGIC
    big = bytes.fromhex("28b52ffd") + bytes([0x00, 0xFF])               # window log 41 -> ~3.7 TiB
    assert fd.new(big)[0] == cz.status.CZ_E_WINDOW_TOO_BIG == oracle.FrameDecoder().new(big)[0]   # frame.cairo:118-127
    z = corpus_pairs(max_orig=2000)[0][1]
    assert fd.new(z)[0] == 0
    w100 = bytes.fromhex("28b52ffd") + bytes([0x00, (17 << 3)])         # window 2^27 > 100 MiB
    assert fd.new(w100)[0] == 0                                          # D4: new() has no cap
    assert fd.reset(w100)[0] == cz.status.CZ_E_WINDOW_SIZE_TOO_BIG       # frame_decoder.cairo:92
    fd.close()


def test_device_pointer_batch_full_size(cz, ctx):
    """BASELINE config sizes through cz_decode_batch_device with torch-owned HBM buffers;
    checked by per-frame XXH64 against the oracle on a sample and by size/status on all."""
    import torch
    from cairo_zstd_amd import synth
    n = 2048
    b = synth.generate("full_4a", n)
    out_off, out_cap, total = b.out_layout()
    dev = torch.device("cuda:0")
    t_in = torch.from_numpy(b.base).to(dev)
    t_off, t_len = torch.from_numpy(b.off.astype(np.int64)).to(dev), torch.from_numpy(b.length.astype(np.int64)).to(dev)
    t_ooff, t_ocap = torch.from_numpy(out_off.astype(np.int64)).to(dev), torch.from_numpy(out_cap.astype(np.int64)).to(dev)
    t_out = torch.zeros(total, dtype=torch.uint8, device=dev)
    t_res = torch.zeros(n * cz.RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    c2 = cz.Context(0, torch.cuda.current_stream().cuda_stream)
    c2.decode_batch_device(t_in.data_ptr(), t_off.data_ptr(), t_len.data_ptr(), n, t_out.data_ptr(), t_ooff.data_ptr(),
                           t_ocap.data_ptr(), t_res.data_ptr())
    torch.cuda.synchronize()
    res = t_res.cpu().numpy().view(cz.RESULT_DTYPE)
    out = t_out.cpu().numpy()
    assert (res["status"] == 0).all(), np.unique(res["status"], return_counts=True)
    assert (res["bytes_produced"] == b.regen).all() and (res["bytes_consumed"] == b.length).all()
    for i in range(0, n, 97):
        st, ref, _ = oracle.decode_frame(b.frame(i), cap=int(b.regen[i]) + 16)
        got = out[int(out_off[i]): int(out_off[i] + b.regen[i])].tobytes()
        assert st == 0 and oracle.xxh64(got) == oracle.xxh64(ref), i
    assert c2.last_kernel_ms() > 0
    c2.close()


@pytest.mark.parametrize("arena_mb,lit_mb", [(0.02, 0), (256, 0), (256, 256), (256, 0.05)])
def test_chain_prepass_matches_oracle(cz, arena_mb, lit_mb):
    """cz_chain_kernel + cz_decode_frames_kernel (two-pass pipeline) == oracle, also when the record
    arena is far too small (frames fall back to in-kernel chains) and on malformed frames."""
    from cairo_zstd_amd import synth
    c = cz.Context(0)
    c.set_chain_arena(int(arena_mb * (1 << 20)), min_sequences=0)      # pre-pass every frame, however short its chains
    c.set_literal_arena(int(lit_mb * (1 << 20)))                       # huff0 / tile kernels and the execute kernel with it (0 = off; 0.05 MB: most frames do not fit)
    try:
        frames, caps = [], []
        for kind, n in (("full_4a", 12), ("full_4b", 4), ("mix", 800), ("huf_literals", 4), ("raw_rle", 4)):
            b = synth.generate(kind, n, first_index=77)
            frames += [b.frame(i) for i in range(n)]
            caps += [int(r) + 16 for r in b.regen]
        for name, z, orig in corpus_pairs():
            frames.append(z)
            caps.append(len(orig) + 32)
        for idx, (name, z, orig) in enumerate(corpus_pairs(max_orig=6000)):
            for m in _mutations(z, idx)[:10]:
                frames.append(m)
                caps.append(len(orig) * 2 + 4096)
        c.set_verify_checksum(True)                                      # the record-driven path feeds the XXH64 pass too
        got = cz.decode_batch_host(frames, caps, c)
        assert c.last_chain_ms() > 0.0 and c.last_kernel_ms() > c.last_chain_ms()   # both kernels ran and were timed
        bad = []
        for i, (fr, cap, (r, out)) in enumerate(zip(frames, caps, got)):
            st, ref, info = oracle.decode_frame(fr, cap=cap)
            if st != int(r["status"]) or (st == 0 and out != ref):
                bad.append((i, cz.status.name(r["status"]), cz.status.name(st)))
            elif st == 0 and info["has_checksum"]:
                ok = bool(r["flags"] & cz.RESULT_CHECKSUM_COMPUTED) and int(r["calculated_checksum"]) == oracle.xxh64(ref) & 0xFFFFFFFF
                if not ok:
                    bad.append((i, "XXH64"))
        assert not bad, bad[:10]
        if arena_mb == 256 and lit_mb == 0:
            # without a literal arena cz_scan_kernel must list the same frames for the chain kernel as with one (its literal and
            # copy counts are not computed then, and must not be looked at)
            n_chain = c.last_prepass_counts(len(frames))[0]
            assert n_chain >= 600, n_chain
            c.set_literal_arena(256 << 20)
            cz.decode_batch_host(frames, caps, c)
            with_chain, with_lits = c.last_prepass_counts(len(frames))
            assert with_chain == n_chain and with_lits > 0, (n_chain, with_chain, with_lits)
        c.set_chain_arena(0)
        cz.decode_batch_host(frames[:4], caps[:4], c)
        assert c.last_chain_ms() == 0.0
    finally:
        c.close()


def test_decode_stream_multi_frame_with_skippable_frames(cz, ctx):
    """A multi-frame .zst with skippable frames in between goes through ONE batch launch (cz_stream_split +
    cz_decode_batch_host); src/frame.cairo:160-166 leaves the skipping to the caller."""
    pairs = corpus_pairs(max_orig=70000)[::5]
    skip = bytes.fromhex("532a4d18") + (11).to_bytes(4, "little") + b"not a frame"
    data, want = b"", b""
    for i, (name, z, orig) in enumerate(pairs):
        data += z + (skip if i % 3 == 0 else b"")
        want += orig
    assert cz.decode_stream(data, ctx) == want
    with pytest.raises(cz.CzError):
        cz.decode_stream(data[:-2], ctx)


def test_block_decoder_walks_frames_block_by_block(cz, ctx):
    """BlockDecoder::read_block_header + decode_block_content (src/decoding/block_decoder.cairo:237-278, :77-137)
    against a DecoderScratch on the device, one block per call, next to the oracle's frame decoder stepping one block;
    the first block also against the oracle's fresh-scratch single-block entry."""
    for name, z, orig in corpus_pairs()[3::7]:
        st, fh, _ = cz.read_frame_header(z)
        assert st == 0
        ws = cz.DecoderScratch(ctx, fh.window_size)
        bd = cz.BlockDecoder()
        od = oracle.FrameDecoder()
        od.new(z)
        pos, out, first = fh.header_len, b"", True
        assert bd.decode_block_content(cz.BlockHeader(), ws, b"")[0] == cz.status.CZ_E_BLOCK_EXPECTED_HEADER   # :86-88
        while True:
            st, bh, used = bd.read_block_header(z[pos:])
            assert st == 0 and used == 3, name
            st, took = bd.decode_block_content(bh, ws, z[pos + 3:])
            ost, oused, ofin = od.decode_blocks(z[pos:], oracle.FrameDecoder.UPTO_BLOCKS, 1)
            assert st == 0 == ost and 3 + took == oused - (4 if ofin and (fh.descriptor >> 2) & 1 else 0), (name, pos)
            assert bd.internal_state == cz.BlockDecoder.READY_FOR_HEADER
            if first:
                fst, fout, fused = oracle.decode_single_block(z[pos:], cap=len(orig) + 64, window=int(fh.window_size))
                assert fst == 0 and fused == 3 + took and ws.buffer_len() == len(fout), name
                first = False
            assert ws.total_output() == len(out) + ws.buffer_len()
            pos += 3 + took
            part = ws.drain_to_window_size(cap=len(orig) + 64)
            out += part or b""
            od.collect(cap=len(orig) + 64)
            if bh.last_block:
                break
        out += ws.drain(cap=len(orig) + 64)
        assert out == orig, name
        assert ws.hash_digest() == oracle.xxh64(orig), name
        ws.close()


def test_many_tiny_blocks_stream_through_decode_from_to(cz, ctx):
    """A frame of thousands of small Raw / RLE blocks fed in slices: the resident output stays bounded (exact sizes
    for Raw / RLE, drained bytes are dropped) and the result equals the oracle's."""
    rng = np.random.default_rng(11)
    data = rng.integers(0, 256, 400_000, dtype=np.uint8).tobytes()
    frame = bytearray(b"\x28\xb5\x2f\xfd" + bytes([0x04, 0x00]))          # checksum flag, window 1 KiB
    pos, nblk = 0, 0
    while pos < len(data):
        n = int(rng.integers(1, 200))
        chunk = data[pos:pos + n]
        last = pos + n >= len(data)
        if nblk % 5 == 4:                                                     # an RLE block now and then
            chunk = bytes([chunk[0]]) * len(chunk)
            data = data[:pos] + chunk + data[pos + len(chunk):]
            v = (1 if last else 0) | (1 << 1) | (len(chunk) << 3)
            frame += bytes([v & 255, (v >> 8) & 255, (v >> 16) & 255]) + chunk[:1]
        else:
            v = (1 if last else 0) | (len(chunk) << 3)
            frame += bytes([v & 255, (v >> 8) & 255, (v >> 16) & 255]) + chunk
        pos += len(chunk)
        nblk += 1
    frame += (oracle.xxh64(data) & 0xFFFFFFFF).to_bytes(4, "little")
    z = bytes(frame)
    assert nblk > 3000
    fd, od = cz.FrameDecoder(ctx), oracle.FrameDecoder()
    st, hl, _ = fd.new(z)
    od.new(z)
    assert st == 0
    p, out, guard = hl, b"", 0
    while not fd.is_finished() and guard < 500:
        guard += 1
        chunk = z[p:p + 50_000]
        a = fd.decode_from_to(chunk, cap=len(data) + 64)
        b = od.decode_from_to(chunk, cap=len(data) + 64)
        assert a == b and a[0] == 0, guard
        p += a[1]
        out += a[2]
    out += fd.decode_from_to(b"", cap=len(data) + 64)[2]
    assert out == data and fd.get_calculated_checksum() == fd.get_checksum_from_data() == oracle.xxh64(data) & 0xFFFFFFFF
    fd.close()


def test_bytes_read_counter_after_a_failed_block(cz, ctx):
    """frame_decoder.cairo:172: the 3 header bytes of the block whose body fails are already counted."""
    name, z, orig = [p for p in corpus_pairs(max_orig=20000) if len(p[1]) > 300][0]
    for cut in (len(z) - 9, len(z) // 2):
        a = bytearray(z)
        a[cut] ^= 0x55
        bad = bytes(a)
        fd, od = cz.FrameDecoder(ctx), oracle.FrameDecoder()
        st, hl, _ = fd.new(bad)
        ost, ohl, _ = od.new(bad)
        assert (st, hl) == (ost, ohl)
        if st:
            continue
        ra = fd.decode_blocks(bad[hl:])
        rb = od.decode_blocks(bad[hl:])
        assert ra[0] == rb[0]
        assert fd.bytes_read_from_source() == od.bytes_read_from_source(), (cut, ra, rb)
        assert fd.blocks_decoded() == od.blocks_decoded()
        fd.close()


def _large_corpus():
    m = json.load(open(os.path.join(GOLDEN, "decode_corpus_manifest.json")))
    d = os.path.join(GOLDEN, "decode_corpus_large")
    out = []
    for name in sorted(m):
        if not m[name]["committed"]:
            out.append((name, open(os.path.join(d, name + ".zst"), "rb").read(), m[name]))
    return out


@pytest.mark.parametrize("mode", ["single_kernel", "prepass", "prepass_decode_kernel_only"])
def test_corpus_large_frames_vs_manifest(cz, mode):
    """The 31 corpus frames whose originals exceed 64 KiB (986 blocks in one frame, a 3.5 MiB window, 41 958 sequences
    and 72 521 literals in one block: src/tests/decoding.cairo:4-21 over data/decode_corpus): only the compressed side
    is committed; the decoded bytes are checked by sha256 and XXH64 against the manifest made from the reference's
    originals, and the frame's own content checksum is verified on the device."""
    files = _large_corpus()
    assert len(files) == 31
    c = cz.Context(0)
    try:
        if mode != "single_kernel":
            c.set_chain_arena(512 << 20, min_sequences=0)
            c.set_exec_kernel(mode == "prepass")                        # cz_execute_frames_kernel (default) / everything on cz_decode_frames_kernel
            c.set_literal_arena(64 << 20)
        c.set_verify_checksum(True)
        got = cz.decode_batch_host([z for _, z, _ in files], [e["orig_len"] + 64 for _, _, e in files], c)
        for (name, z, e), (r, out) in zip(files, got):
            assert int(r["status"]) == 0, (name, cz.status.name(r["status"]), r["detail"])
            assert len(out) == e["orig_len"] and int(r["bytes_consumed"]) == len(z), name
            assert hashlib.sha256(out).hexdigest() == e["orig_sha256"], name
            assert f"{oracle.xxh64(out):016x}" == e["xxh64"], name
            assert int(r["checksum_from_data"]) == int(e["xxh64"], 16) & 0xFFFFFFFF, name
            assert r["flags"] & cz.RESULT_CHECKSUM_MATCH, name
        if mode != "single_kernel":
            assert c.last_chain_ms() > 0
            assert (c.last_exec_ms() > 0) == (mode == "prepass")
    finally:
        c.close()


def test_upto_bytes_strategy_matches_oracle(cz, ctx):
    """BlockDecodingStrategy::UptoBytes (src/frame_decoder.cairo:209-213): stop once at least n bytes were produced."""
    for name, z, orig in corpus_pairs()[2::9]:
        for n in (1, 700, 5000, 40000):
            fd, od = cz.FrameDecoder(ctx), oracle.FrameDecoder()
            st, hl, _ = fd.new(z)
            od.new(z)
            pos, out, guard = hl, b"", 0
            while not fd.is_finished() and guard < 3000:
                guard += 1
                a = fd.decode_blocks(z[pos:], cz.BlockDecodingStrategy.UPTO_BYTES, n)
                b = od.decode_blocks(z[pos:], oracle.FrameDecoder.UPTO_BYTES, n)
                assert a == b and a[0] == 0, (name, n, a, b)
                pos += a[1]
                assert fd.blocks_decoded() == od.blocks_decoded() and fd.can_collect() == od.can_collect(), (name, n)
                out += fd.collect(cap=len(orig) + 64) or b""
                od.collect(cap=len(orig) + 64)
            assert out == orig, (name, n)
            fd.close()


def test_d2_huffman_weight_fse_log_10_is_a_pinned_divergence(cz, ctx):
    """the reference accepts any 4-bit accuracy log for the FSE table of the Huffman weights
    (max_log 100, src/huff0/huff0_decoder.cairo:176); the oracle decodes the committed vector (made by
    scripts/gen_d2_vector.py, log 10); the device caps the log at 9 and says CZ_E_UNSUPPORTED — with and without
    the chain pre-pass, and the neighbours of the frame are untouched."""
    d = os.path.join(GOLDEN, "vectors")
    z, want = open(os.path.join(d, "d2_weight_log10.zst"), "rb").read(), open(os.path.join(d, "d2_weight_log10"), "rb").read()
    st, out, _ = oracle.decode_frame(z, cap=64)
    assert st == 0 and out == want
    good = corpus_pairs(max_orig=3000)[0]
    for prepass in (False, True):
        c = cz.Context(0)
        if prepass:
            c.set_chain_arena(8 << 20, min_sequences=0)
        got = cz.decode_batch_host([good[1], z, good[1]], [len(good[2]) + 8, 64, len(good[2]) + 8], c)
        c.close()
        assert int(got[1][0]["status"]) == cz.status.CZ_E_UNSUPPORTED
        assert int(got[0][0]["status"]) == 0 and got[0][1] == good[2] and got[2][1] == good[2]


def test_bench_configuration_all_frames(cz):
    """bench.py's default step — config 4a, chain pre-pass on, default chain_min_sequences — on 2 304 frames, EVERY
    frame compared with the oracle by XXH64, the output buffer poisoned before the launch."""
    import torch
    from cairo_zstd_amd import synth
    n = 2304
    b = synth.generate("full_4a", n, first_index=40000)
    out_off, out_cap, total = b.out_layout(256)
    ref_all, olen, ost = oracle.decode_batch(b.base, b.off, b.length, out_off, out_cap, total, nthreads=os.cpu_count() or 8)
    assert (ost == 0).all() and (olen == b.regen).all()
    dev = torch.device("cuda:0")
    t_in = torch.from_numpy(b.base).to(dev)
    t = [torch.from_numpy(x.astype(np.int64)).to(dev) for x in (b.off, b.length, out_off, out_cap)]
    t_out = torch.full((total,), 0xA5, dtype=torch.uint8, device=dev)
    t_res = torch.zeros(n * cz.RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    c = cz.Context(0, torch.cuda.current_stream().cuda_stream)
    c.set_chain_arena(int(b.length.sum()) * 6 + (64 << 20))
    c.set_literal_arena(int(b.regen.sum()) + (16 << 20))
    try:
        c.decode_batch_device(t_in.data_ptr(), t[0].data_ptr(), t[1].data_ptr(), n, t_out.data_ptr(), t[2].data_ptr(), t[3].data_ptr(), t_res.data_ptr())
        torch.cuda.synchronize()
        assert c.last_chain_ms() > 0
        res = t_res.cpu().numpy().view(cz.RESULT_DTYPE)
        out = t_out.cpu().numpy()
        assert (res["status"] == 0).all() and (res["bytes_produced"] == b.regen).all() and (res["bytes_consumed"] == b.length).all()
        for i in range(n):
            lo, hi = int(out_off[i]), int(out_off[i] + b.regen[i])
            assert oracle.xxh64(out[lo:hi]) == oracle.xxh64(ref_all[lo:hi]), i
    finally:
        c.close()


def test_repeated_launch_is_replayed_as_a_graph(cz):
    """A launch that repeats the one before it (same pointers, sizes, context settings) is captured as a hipGraph and replayed from
    then on (cz_context_set_graph_replay(ctx, 1); off by default).  The replay must be the same work: outputs against the oracle before and
    after the BYTES behind the same pointers are replaced by another batch (a graph holds pointers and grids, never data); a
    change of the context's settings ends the replays until the new launch has repeated; with replay off nothing is captured;
    the library's event times stay available in a replay (its events are event-record nodes of the graph)."""
    import torch
    from cairo_zstd_amd import synth
    dev = torch.device("cuda:0")
    for kind, n in (("mix", 500), ("full_4a", 260), ("raw_rle", 300)):
        batches = [synth.generate(kind, n, first_index=fi) for fi in (5, 7001)]
        in_bytes = max(len(b.base) for b in batches)
        cap_each = np.maximum(batches[0].regen, batches[1].regen).astype(np.uint64) + 256
        out_off = np.concatenate([[0], np.cumsum(cap_each)[:-1]]).astype(np.uint64)
        total = int(cap_each.sum())
        t_in = torch.zeros(in_bytes, dtype=torch.uint8, device=dev)
        t_off, t_len = torch.zeros(n, dtype=torch.int64, device=dev), torch.zeros(n, dtype=torch.int64, device=dev)
        t_ooff, t_ocap = torch.from_numpy(out_off.astype(np.int64)).to(dev), torch.from_numpy(cap_each.astype(np.int64)).to(dev)
        t_out = torch.empty(total, dtype=torch.uint8, device=dev)
        t_res = torch.zeros(n * cz.RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
        refs = []
        for b in batches:
            ref, olen, ost = oracle.decode_batch(b.base, b.off, b.length, out_off, cap_each, total, nthreads=os.cpu_count() or 8)
            assert (ost == 0).all() and (olen == b.regen).all()
            refs.append(ref)

        def load(b):
            t_in[:len(b.base)].copy_(torch.from_numpy(b.base).to(dev))
            t_off.copy_(torch.from_numpy(b.off.astype(np.int64)).to(dev))
            t_len.copy_(torch.from_numpy(b.length.astype(np.int64)).to(dev))

        def run(c, b, ref):
            t_out.fill_(0xA5)
            c.decode_batch_device(t_in.data_ptr(), t_off.data_ptr(), t_len.data_ptr(), n, t_out.data_ptr(), t_ooff.data_ptr(), t_ocap.data_ptr(), t_res.data_ptr())
            torch.cuda.synchronize()
            res, out = t_res.cpu().numpy().view(cz.RESULT_DTYPE), t_out.cpu().numpy()
            assert (res["status"] == 0).all() and (res["bytes_produced"] == b.regen).all(), kind
            for i in range(n):
                lo, hi = int(out_off[i]), int(out_off[i] + b.regen[i])
                assert oracle.xxh64(out[lo:hi]) == oracle.xxh64(ref[lo:hi]), (kind, i)
            return c.last_launch_was_replay()

        c = cz.Context(0, torch.cuda.current_stream().cuda_stream)
        try:
            c.set_chain_arena(int(max(b.length.sum() for b in batches)) * 6 + (64 << 20))
            c.set_literal_arena(int(max(b.regen.sum() for b in batches)) + (16 << 20))
            load(batches[0])
            assert [run(c, batches[0], refs[0]) for _ in range(2)] == [False, False], kind   # (off by default)
            c.set_graph_replay(True)
            assert [run(c, batches[0], refs[0]) for _ in range(4)] == [False, True, True, True], kind
            ms, chain = c.last_kernel_ms(), c.last_chain_ms()
            assert 0.0 < chain <= ms, (kind, ms, chain)                 # (stamped by the replay itself)
            load(batches[1])                                            # other frames behind the same pointers
            assert [run(c, batches[1], refs[1]) for _ in range(2)] == [True, True], kind
            c.set_verify_checksum(True)                                 # the context changed: an ordinary launch, then a new graph
            assert [run(c, batches[1], refs[1]) for _ in range(3)] == [False, True, True], kind
            c.set_graph_replay(False)
            assert [run(c, batches[1], refs[1]) for _ in range(3)] == [False, False, False], kind
            c.set_graph_replay(True)
            load(batches[0])
            assert [run(c, batches[0], refs[0]) for _ in range(3)] == [False, True, True], kind
        finally:
            c.close()


def test_execute_frames_kernel_matches_oracle(cz):
    """cz_execute_frames_kernel (the decode kernel's source without its decoders, for the frames the pre-pass finished) followed
    by cz_decode_frames_kernel on the frames it hands over: same results as the oracle on both kinds."""
    from cairo_zstd_amd import synth
    c = cz.Context(0)
    c.set_chain_arena(256 << 20, min_sequences=0)
    c.set_literal_arena(128 << 20)
    c.set_exec_kernel(True)
    try:
        frames, caps = [], []
        for kind, n in (("full_4a", 8), ("full_4b", 3), ("mix", 400)):
            b = synth.generate(kind, n, first_index=311)
            frames += [b.frame(i) for i in range(n)]
            caps += [int(r) + 16 for r in b.regen]
        for name, z, orig in corpus_pairs():
            frames.append(z)
            caps.append(len(orig) + 32)
        for idx, (name, z, orig) in enumerate(corpus_pairs(max_orig=6000)):
            for m in _mutations(z, idx)[:6]:
                frames.append(m)
                caps.append(len(orig) * 2 + 4096)
        got = cz.decode_batch_host(frames, caps, c)
        assert c.last_exec_ms() > 0.0
        bad = []
        for i, (fr, cap, (r, out)) in enumerate(zip(frames, caps, got)):
            st, ref, info = oracle.decode_frame(fr, cap=cap)
            if st != int(r["status"]) or (st == 0 and (out != ref or int(r["bytes_consumed"]) != info["consumed"] or int(r["blocks_decoded"]) != info["blocks"])):
                bad.append((i, cz.status.name(r["status"]), cz.status.name(st)))
        assert not bad, bad[:10]
    finally:
        c.close()

def test_wexec_kernel_waits_are_bounded(cz):
    """Every device-side wait of cz_wexec_kernel counts its polls (WX_SPIN_LIMIT).  CZ_DEBUG_WX_POISON: chunk 2 of every block never
    publishes its look-back entry — a protocol error made on purpose —, so the chunks behind it poll to the bound, the workgroup
    gives the frame up, and cz_decode_frames_kernel decodes it (sequence_execution.cairo:12-83 by one wave): same bytes as the
    oracle, nothing hangs.  Frames of at most two chunks per block are not affected."""
    from cairo_zstd_amd import synth
    c = cz.Context(0)
    c.set_chain_arena(256 << 20, min_sequences=0)
    c.set_literal_arena(64 << 20)
    c.set_wexec_kernel(True, force=True)
    try:
        b = synth.generate("full_4a", 40, first_index=31)
        frames, caps = [b.frame(i) for i in range(b.n)], [int(r) for r in b.regen]
        for name, z, orig in corpus_pairs(max_orig=1500)[:8]:
            frames.append(z)
            caps.append(len(orig) + 32)
        refs = [oracle.decode_frame(fr, cap=cap) for fr, cap in zip(frames, caps)]
        for flags in (0, cz.DEBUG_WX_POISON):
            c.set_debug_flags(flags)
            got = cz.decode_batch_host(frames, caps, c)
            for i, ((st, ref, _), (r, out)) in enumerate(zip(refs, got)):
                assert int(r["status"]) == st == 0 and out == ref, (flags, i, cz.status.name(r["status"]))
            listed, finished, handed = c.last_wexec_counts()
            if flags:
                assert listed >= b.n and finished < listed - b.n + 1 and handed >= b.n, (listed, finished, handed)   # every config-4a frame ran into the bound
                assert c.last_fallback_count() >= b.n
            else:
                assert finished + handed <= listed and finished >= 1, (listed, finished, handed)
    finally:
        c.set_debug_flags(0)
        c.close()


def test_frames_handed_back_by_huf_kernel_are_listed_once(cz):
    """a frame listed for cz_wexec_kernel whose literals section cz_huf_kernel hands back (the D5 block — 4 huff0 streams
    that do not split ceil(regen / 4), literals_section_decoder.cairo:95-115 — with a sequence behind it, so that it has chain
    records) was put on the fall-back list by whichever execute kernel met it — both, when their timing allowed — and decoded
    twice at once; with most of a batch handed back the list overflowed.  Now the kernel that sets CZ_PRE_LISTED first lists the
    frame: the list holds every handed-back frame exactly once."""
    import json
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vectors")
    man = json.load(open(os.path.join(d, "manifest_r5.json")))
    name = "d5_uneven_split_with_sequences.zst"
    z, want = open(os.path.join(d, name), "rb").read(), open(os.path.join(d, name[:-4] + ".orig"), "rb").read()
    assert hashlib.sha256(want).hexdigest() == man[name]["orig_sha256"]
    n_bad = 6000
    good = [(zz, orig) for _, zz, orig in corpus_pairs(max_orig=4000)][:40]
    frames = [z] * n_bad + [zz for zz, _ in good]
    caps = [len(want) + 16] * n_bad + [len(o) + 32 for _, o in good]
    c = cz.Context(0)
    c.set_chain_arena(256 << 20, min_sequences=0)
    c.set_literal_arena(64 << 20)
    try:
        for force, flags in ((True, cz.DEBUG_NO_HUF1), (True, 0), (False, cz.DEBUG_NO_HUF1)):
            c.set_wexec_kernel(True, force=force)
            c.set_debug_flags(flags)
            for rep in range(3):
                got = cz.decode_batch_host(frames, caps, c)
                for i, (r, out) in enumerate(got):
                    assert int(r["status"]) == 0 and out == (want if i < n_bad else good[i - n_bad][1]), (force, flags, rep, i, cz.status.name(r["status"]))
                fb = c.last_fallback_count()
                if flags:
                    assert n_bad <= fb <= len(frames), (force, rep, fb)     # every D5 frame handed back (cz_huf_kernel alone decodes literals), none twice
                else:
                    assert fb <= len(frames), (force, rep, fb)              # (cz_huf1_kernel keeps the sections it gets to first)
    finally:
        c.set_debug_flags(0)
        c.close()


def test_side_by_side_split_does_not_depend_on_submission_order(cz):
    """Side by side, cz_wexec_kernel and cz_execute_frames_kernel each keep to one half of the CUs by the hardware's CU id (cz_cu_side):
    whichever kernel the host submits first, cz_wexec_kernel gets its CUs and finishes a comparable share of a far-offset batch
    (round 4: submitted second it found every CU held by the other kernel's persistent waves and finished next to nothing).  Outputs
    against the oracle in both orders."""
    from cairo_zstd_amd import synth
    b = synth.generate("full_4a", 3000, first_index=5)
    frames, caps = [b.frame(i) for i in range(b.n)], [int(r) for r in b.regen]
    refs = [oracle.decode_frame(frames[i], cap=caps[i]) for i in range(0, b.n, 100)]
    share = []
    for flags in (0, cz.DEBUG_EXEC_FIRST):
        c = cz.Context(0)
        try:
            c.set_chain_arena(1024 << 20, min_sequences=0)
            c.set_literal_arena(128 << 20)
            c.set_debug_flags(flags)
            for rep in range(3):
                got = cz.decode_batch_host(frames, caps, c)
            assert all(int(r["status"]) == 0 for r, _ in got)
            for k, i in enumerate(range(0, b.n, 100)):
                assert got[i][1] == refs[k][1], (flags, i)
            listed, finished, handed = c.last_wexec_counts()
            assert listed == b.n and handed == 0, (flags, listed, finished, handed)
            share.append(finished)
        finally:
            c.close()
    assert min(share) > 0.15 * b.n and max(share) < 0.6 * b.n and abs(share[0] - share[1]) < 0.15 * b.n, share


def test_split_chain_prepass_and_early_execute_launches(cz):
    """cz_context_set_early_execute(1): the chain pre-pass as two launches of cz_chain_kernel (blocks of 4 096 sequences and more, which
    are published block by block with an agent-scope release; all others) and the execute stage started behind the second — the early
    launch of cz_execute_frames_kernel on the frames without a large block, the early launch of cz_wexec_kernel on the batch's large
    frames, each large block behind its chain's flag — with every frame claimed by whoever gets to it first.  Same bytes and statuses
    as the oracle on a corpus-like batch large enough for the large-frames arrangement, config 4a / 4b frames, the reference corpus
    and damaged frames; several repeats (the arrangement depends on timing)."""
    from cairo_zstd_amd import synth
    c = cz.Context(0)
    c.set_chain_arena(768 << 20, min_sequences=0)
    c.set_literal_arena(384 << 20)
    c.set_early_execute(True)
    try:
        frames, caps = [], []
        for kind, n in (("mix", 2300), ("full_4a", 24), ("full_4b", 6)):
            bb = synth.generate(kind, n, first_index=1234)
            frames += [bb.frame(i) for i in range(n)]
            caps += [int(r) + 16 for r in bb.regen]
        for name, z, orig in corpus_pairs():
            frames.append(z)
            caps.append(len(orig) + 32)
        for idx, (name, z, orig) in enumerate(corpus_pairs(max_orig=20000)):
            for m in _mutations(z, idx)[:4]:
                frames.append(m)
                caps.append(len(orig) * 2 + 4096)
        refs = [oracle.decode_frame(fr, cap=cap) for fr, cap in zip(frames, caps)]
        for rep in range(3):
            got = cz.decode_batch_host(frames, caps, c)
            bad = [(i, cz.status.name(r["status"]), cz.status.name(st)) for i, ((st, ref, _), (r, out)) in enumerate(zip(refs, got)) if st != int(r["status"]) or (st == 0 and out != ref)]
            assert not bad, (rep, bad[:10])
            assert c.last_small_ms() > 0.0
            assert c.last_fallback_count() <= len(frames)
    finally:
        c.close()


@pytest.mark.parametrize("auto", [False, True])
def test_wexec_kernel_side_by_side_matches_oracle(cz, auto):
    """cz_wexec_kernel (a workgroup of 16 waves per frame, the block in hand in an LDS window, chunks of 64 sequences composed by a
    look-back) side by side with cz_execute_frames_kernel, the two claiming frames from one batch: same results as the oracle.
    Forced on (auto = False) every listed frame may go to either kernel — config 4a, config 4b (a block of 224 KiB: the window is
    filled more than once), corpus-like multi-block frames (blocks above the format's 128 KiB, matches into earlier blocks, long
    matches, `offset_value 3 with no literals`), the whole reference corpus (windows up to 3.5 MiB) and damaged frames, which it
    must hand on.  auto = True: the device decides from the batch's offset codes: the far-offset batch side by side, of the near-offset
    batch only the large frames."""
    from cairo_zstd_amd import synth
    c = cz.Context(0)
    c.set_chain_arena(512 << 20, min_sequences=0)
    c.set_literal_arena(256 << 20)
    c.set_wexec_kernel(True, force=not auto)
    try:
        def run(frames, caps):
            got = cz.decode_batch_host(frames, caps, c)
            bad = []
            for i, (fr, cap, (r, out)) in enumerate(zip(frames, caps, got)):
                st, ref, info = oracle.decode_frame(fr, cap=cap)
                if st != int(r["status"]) or (st == 0 and (out != ref or int(r["bytes_consumed"]) != info["consumed"] or int(r["blocks_decoded"]) != info["blocks"]
                                                           or bool(r["flags"] & 2) != info["has_checksum"] or (info["has_checksum"] and int(r["checksum_from_data"]) != info["checksum"]))):
                    bad.append((i, cz.status.name(r["status"]), cz.status.name(st)))
            assert not bad, bad[:10]
            return c.last_wexec_counts()
        b = synth.generate("full_4a", 300, first_index=77)
        listed, finished, handed = run([b.frame(i) for i in range(b.n)], [int(r) for r in b.regen])
        assert listed == b.n and finished > 0 and handed == 0, (listed, finished, handed)   # far offsets: side by side in both modes
        frames, caps = [], []
        for kind, n in (("full_4b", 12), ("mix", 2100 if auto else 700)):    # (auto: a batch large enough for the large-frames arrangement)
            bb = synth.generate(kind, n, first_index=4711)
            frames += [bb.frame(i) for i in range(n)]
            caps += [int(r) + 16 for r in bb.regen]
        for name, z, orig in corpus_pairs():
            frames.append(z)
            caps.append(len(orig) + 32)
        for idx, (name, z, orig) in enumerate(corpus_pairs(max_orig=20000)):
            for m in _mutations(z, idx)[:6]:
                frames.append(m)
                caps.append(len(orig) * 2 + 4096)
        listed, finished, handed = run(frames, caps)
        if auto:
            # near offsets: the large frames (36 000 sequences and more: the config 4b frames, a few of the mix) on cz_wexec_kernel,
            # all others on cz_execute_frames_kernel
            assert 12 <= finished < 140 and handed == 0, (listed, finished, handed)
        else:
            # (the damaged frames are handed on by whichever execute kernel gets to them first: small ones usually by the early launch of cz_execute_frames_kernel)
            assert listed > 100 and finished > 50 and (handed > 0 or c.last_fallback_count() > 0), (listed, finished, handed, c.last_fallback_count())
    finally:
        c.close()


def test_more_frames_than_resident_workgroups_with_prepass(cz):
    """Grid-size regression: both launches of the two-pass pipeline index per-workgroup scratch."""
    from cairo_zstd_amd import synth
    n = 9000
    b = synth.generate("mix", n, first_index=100000)
    c = cz.Context(0)
    c.set_chain_arena(int(b.length.sum()) * 8 + (16 << 20), min_sequences=2048)   # a gate: only frames whose first sequences section is long
    c.set_literal_arena(int(b.regen.sum()) + (16 << 20))
    try:
        out_off, out_cap, total = b.out_layout()
        out, res = c.decode_batch_host(b.base, b.off, b.length, out_off, out_cap, total)
        assert (res["status"] == 0).all() and (res["bytes_produced"] == b.regen).all()
        for i in range(0, n, 331):
            st, ref, _ = oracle.decode_frame(b.frame(i), cap=int(b.regen[i]) + 16)
            assert st == 0 and out[int(out_off[i]): int(out_off[i] + b.regen[i])].tobytes() == ref, i
    finally:
        c.close()


def test_bench_two_ranks_spawned_by_gpus_flag(cz):
    """bench.py --gpus 2 starts two ranks by itself (torch.distributed.run) and reports n_gpus 2: rehearsed here over gloo
    with both ranks on the one GPU of this box (RCCL needs one GPU per rank); mix workload, frames dealt by algorithmic
    bytes, decode-only and decode+gather timed."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--workload", "mix", "--frames", "300",
                        "--steps", "2", "--warmup", "1", "--gather", "--no-other-workloads", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["bit_exact"] is True and line["config"]["frames_total"] == 600
    assert "dealt by algorithmic bytes" in line["config"]["parallelism"]
    assert line["with_gather_to_rank0"]["ms_per_step"] >= line["ms_per_step"] * 0.5 and line["with_gather_to_rank0"]["gathered_bytes_per_step"] > 0


# ---------------------------------------------------------------- dictionaries
def _dict_fixture():
    import glob
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dict")
    raw = open(os.path.join(d, "dict.bin"), "rb").read()
    frames = [(os.path.basename(z), open(z, "rb").read(), open(z[:-4] + ".orig", "rb").read()) for z in sorted(glob.glob(os.path.join(d, "frame_*.zst")))]
    return raw, frames


def test_dictionary_decode_dict_matches_oracle(cz, ctx):
    """DictionaryTrait::decode_dict (src/decoding/dictionary.cairo:35-91) on the device against the oracle: fields of the
    trained dictionary, and the same status for every way of cutting or corrupting it."""
    raw, _ = _dict_fixture()
    od = oracle.Dictionary(raw)
    assert od.status == 0
    gd = cz.Dictionary(ctx, raw)
    assert gd.id == od.info["id"] and gd.content_len == od.info["content_len"]
    assert gd.offset_hist == (od.info["hist0"], od.info["hist1"], od.info["hist2"])
    gd.close()
    rng = np.random.default_rng(7)
    muts = [raw[:n] for n in (0, 3, 7, 8, 9, 20, 40, 60, 90, od.info["content_off"] - 13, od.info["content_off"] - 1, od.info["content_off"])]
    for _ in range(40):
        a = bytearray(raw)
        a[int(rng.integers(0, od.info["content_off"]))] ^= 1 << int(rng.integers(0, 8))
        muts.append(bytes(a))
    for m in muts:
        want = oracle.Dictionary(m).status
        try:
            g = cz.Dictionary(ctx, m)
            got = 0
            g.close()
        except cz.CzError as e:
            got = e.code
        if got == cz.status.CZ_E_UNSUPPORTED:                            # D2: Huffman-weight FSE log above 9
            continue
        assert got == want, f"len {len(m)}: device {cz.status.name(got)} oracle {cz.status.name(want)}"


def test_frames_compressed_with_a_dictionary(cz, ctx):
    """A workspace seeded with init_from_dict (src/decoding/scratch.cairo:60-65) decodes frames made by
    ZSTD_compress_usingDict block by block: the first block repeats the dictionary's tables, repeat offsets start from its
    three, matches reach into its content (src/decoding/decode_buffer.cairo:65-93).  Bit-exact with the originals (which
    libzstd's ZSTD_decompress_usingDict reproduced when the fixtures were made) and with the oracle; corrupted frames give
    the oracle's status."""
    raw, frames = _dict_fixture()
    od, gd = oracle.Dictionary(raw), cz.Dictionary(ctx, raw)
    for name, z, orig in frames:
        assert cz.decode_frame_with_dict(z, gd, ctx) == orig, name
    rng = np.random.default_rng(11)
    checked = 0
    for name, z, orig in frames[:6]:
        for _ in range(12):
            a = bytearray(z)
            a[int(rng.integers(6, len(a)))] ^= 1 << int(rng.integers(0, 8))
            want, wout = oracle.decode_frame_with_dict(bytes(a), od, cap=len(orig) * 4 + 4096)
            try:
                gout, got = cz.decode_frame_with_dict(bytes(a), gd, ctx), 0
            except cz.CzError as e:
                gout, got = None, e.code
            if got in (cz.status.CZ_E_UNSUPPORTED, cz.status.CZ_E_OUTPUT_TOO_SMALL) or want == cz.status.CZ_E_OUTPUT_TOO_SMALL:
                continue
            assert got == want, f"{name}: device {cz.status.name(got)} oracle {cz.status.name(want)}"
            if got == 0:
                assert gout == wout, name
            checked += 1
    assert checked > 40
    gd.close()


@pytest.mark.parametrize("prepass", [False, True])
def test_batch_decode_with_a_context_dictionary(cz, ctx, prepass):
    """cz_context_set_dictionary: one launch decodes many frames that share a dictionary (each starts as init_from_dict
    leaves a workspace), with and without the pre-pass; afterwards the context decodes ordinary frames again."""
    raw, frames = _dict_fixture()
    gd = cz.Dictionary(ctx, raw)
    zs = [z for _, z, _ in frames] * 5
    origs = [o for _, _, o in frames] * 5
    ctx.set_chain_arena((sum(len(z) for z in zs) * 8 + (8 << 20)) if prepass else 0)
    ctx.set_literal_arena((sum(len(o) for o in origs) + (4 << 20)) if prepass else 0)
    ctx.set_dictionary(gd)
    try:
        got = cz.decode_batch_host(zs, [len(o) + 32 for o in origs], ctx)
        for i, ((r, out), orig) in enumerate(zip(got, origs)):
            assert int(r["status"]) == 0, f"frame {i}: {cz.status.name(r['status'])}"
            assert out == orig, f"frame {i}"
    finally:
        ctx.set_dictionary(None)
        ctx.set_chain_arena(0)
        ctx.set_literal_arena(0)
    pairs = corpus_pairs()[:8]
    for (name, z, orig), (r, out) in zip(pairs, cz.decode_batch_host([z for _, z, _ in pairs], [len(o) + 32 for _, _, o in pairs], ctx)):
        assert int(r["status"]) == 0 and out == orig, name
    gd.close()


# ---------------------------------------------------------------- round 3: block-parallel literal / copy pre-pass, new vectors
def _r3_vectors():
    import json
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vectors")
    man = json.load(open(os.path.join(d, "manifest_r3.json")))
    return [(n, open(os.path.join(d, n), "rb").read(), man[n]) for n in sorted(man)]


@pytest.mark.parametrize("prepass", [False, True])
def test_r3_vectors_vs_manifest(cz, prepass):
    """libzstd output at levels 1 / 3 / 19, a frame with OF in Repeat mode, direct Huffman weights with unequal nibbles (D1:
    the device follows the zstd nibble order, like libzstd) and an uneven 4-stream split (D5: accepted like the reference):
    sha256 and XXH64 against the manifest written when the vectors were made.  With the pre-pass on, cz_huf_kernel hands
    the D5 frame back (its streams do not split ceil(regen / 4)) and cz_decode_frames_kernel redoes them back to back."""
    vec = _r3_vectors()
    c = cz.Context(0)
    try:
        if prepass:
            c.set_chain_arena(64 << 20, min_sequences=0)
            c.set_literal_arena(16 << 20)
        for huf1 in ((True, False) if prepass else (True,)):
            # without cz_huf1_kernel (which takes sections off the same list while the chain kernel runs: who gets the D5 section is
            # a matter of timing) every section is cz_huf_kernel's, and the hand-back of the D5 frame is deterministic
            c.set_debug_flags(0 if huf1 else cz.DEBUG_NO_HUF1)
            got = cz.decode_batch_host([z for _, z, _ in vec], [e["orig_len"] + 64 for _, _, e in vec], c)
            for (name, z, e), (r, out) in zip(vec, got):
                assert int(r["status"]) == 0, (name, cz.status.name(r["status"]))
                assert len(out) == e["orig_len"] and int(r["bytes_consumed"]) == len(z), name
                assert hashlib.sha256(out).hexdigest() == e["orig_sha256"] and f"{oracle.xxh64(out):016x}" == e["xxh64"], name
            if prepass:
                with_chain, with_lits = c.last_prepass_counts(len(vec))
                # the D5 frame: cz_huf_kernel hands it back (its streams do not split ceil(regen / 4)); cz_huf1_kernel, when it gets to
                # the section first, redoes the streams back to back itself and keeps it
                if huf1:
                    assert with_lits in (len(vec) - 1, len(vec)), (with_chain, with_lits)
                else:
                    assert with_lits == len(vec) - 1, (with_chain, with_lits)
    finally:
        c.close()

def test_chain_kernel_asm_group_and_cpp_step_leave_the_same_records(cz):
    """cz_chain_kernel's hand-scheduled inline-asm group (czc_group_asm, and its wide variant) against the plain C++ step
    (czc_step), which the CPU emulator runs: the same blocks through both, and the chain arenas — headers, state -> code maps and
    the 8-byte record of every sequence (sequence_section_decoder.cairo:223-286) — compared word for word."""
    from cairo_zstd_amd import synth
    frames, caps = [], []
    for kind, n in (("full_4a", 6), ("full_4b", 2), ("mix", 300)):
        b = synth.generate(kind, n, first_index=901)
        frames += [b.frame(i) for i in range(n)]
        caps += [int(r) + 16 for r in b.regen]
    for name, z, orig in corpus_pairs():
        frames.append(z)
        caps.append(len(orig) + 32)
    arenas = []
    for flags in (0, cz.DEBUG_CHAIN_CPP_STEP):
        c = cz.Context(0)
        try:
            c.set_chain_arena(512 << 20, min_sequences=0)
            c.set_literal_arena(256 << 20)
            c.set_debug_flags(flags | cz.DEBUG_NO_HUF1)
            got = cz.decode_batch_host(frames, caps, c)
            assert all(int(r["status"]) == 0 for r, _ in got)
            with_chain, _ = c.last_prepass_counts(len(frames))
            arena, used = c.debug_read_chain_arena(512 << 20)
            arenas.append((arena[:used].copy(), used, with_chain, [out for _, out in got]))
        finally:
            c.close()
    (a0, u0, w0, o0), (a1, u1, w1, o1) = arenas
    assert u0 == u1 and w0 == w1 and w0 > 200, (u0, u1, w0, w1)
    assert o0 == o1
    # blocks are placed in the arena by atomics whose order differs from run to run: compare block by block, found through the
    # headers {nseq << 32 | maps, bitstream offset, next header of the frame, 0}, keyed by what does not depend on the placement
    def blocks(a, used):
        out, at = {}, 64
        while at < used:
            nseq = int(a[at] >> 32)
            size = 4 + 160 + nseq
            key = (nseq, int(a[at + 1]), int(a[at] & 0xFFFFFFFF), a[at + 4:at + 4 + 160].tobytes()[:64])
            out.setdefault(key, []).append(a[at + 4:at + size].tobytes())
            at += size
        return out
    b0, b1 = blocks(a0, u0), blocks(a1, u1)
    assert set(b0) == set(b1)
    for k in b0:
        assert sorted(b0[k]) == sorted(b1[k]), k[:3]


def test_truncated_sequence_section_status_parity(cz):
    """A sequences section that ends before its last sequences are read: the reference unwraps a negative bits_remaining
    (sequence_section_decoder.cairo:279) and panics before it could return NotEnoughBytesForNumSequences (:281); oracle and
    device both report CZ_E_SEQ_NOT_ENOUGH_BYTES (DESIGN.md divergences), through the complete kernel and through the pre-pass
    kernels."""
    from cairo_zstd_amd import synth
    b = synth.generate("full_4a", 4, first_index=31)
    frames, caps = [], []
    for i in range(b.n):
        fr = bytearray(b.frame(i))
        # the single block's size field: cut bytes off the END of the block (= the start of the reversed sequence bitstream is kept,
        # its tail goes), keeping the header consistent
        hdr = 10
        h = fr[hdr] | (fr[hdr + 1] << 8) | (fr[hdr + 2] << 16)
        size = h >> 3
        for cut in (1, 7, 300):
            g = bytearray(fr[:hdr + 3 + size - cut])
            nh = (h & 7) | ((size - cut) << 3)
            g[hdr:hdr + 3] = bytes((nh & 0xFF, (nh >> 8) & 0xFF, (nh >> 16) & 0xFF))
            frames.append(bytes(g)); caps.append(int(b.regen[i]) + 64)
    want = [oracle.decode_frame(fr, cap=cap)[0] for fr, cap in zip(frames, caps)]
    assert any(w == cz.status.CZ_E_SEQ_NOT_ENOUGH_BYTES for w in want), [cz.status.name(w) for w in want]
    for prepass in (False, True):
        c = cz.Context(0)
        try:
            if prepass:
                c.set_chain_arena(64 << 20, min_sequences=0)
                c.set_literal_arena(16 << 20)
            got = cz.decode_batch_host(frames, caps, c)
            assert [int(r["status"]) for r, _ in got] == want, ([cz.status.name(r["status"]) for r, _ in got], [cz.status.name(w) for w in want])
        finally:
            c.close()


@pytest.mark.parametrize("kind,n", [("raw_rle", 300), ("huf_literals", 40), ("mix", 600)])
def test_prepass_kernels_finish_frames_without_sequences(cz, kind, n):
    """cz_tile_kernel (Raw / RLE runs) and cz_huf_kernel (Huffman literals straight into the output) produce whole frames; the
    scan writes their result records and neither decode kernel walks them: same bytes, sizes and block counts as the oracle,
    also with the content checksum verified on the device (then cz_execute_frames_kernel walks the frames to hash them)."""
    from cairo_zstd_amd import synth
    b = synth.generate(kind, n, first_index=17)
    frames = [b.frame(i) for i in range(n)]
    caps = [int(r) + 8 for r in b.regen]
    refs = [oracle.decode_frame(fr, cap=cap) for fr, cap in zip(frames, caps)]
    for verify in (False, True):
        c = cz.Context(0)
        try:
            c.set_chain_arena(128 << 20, min_sequences=0)
            c.set_literal_arena(int(b.regen.sum()) + (8 << 20))
            c.set_verify_checksum(verify)
            got = cz.decode_batch_host(frames, caps, c)
            for i, ((st, ref, info), (r, out)) in enumerate(zip(refs, got)):
                assert int(r["status"]) == st == 0 and out == ref, (kind, i, cz.status.name(r["status"]))
                assert int(r["bytes_consumed"]) == info["consumed"] and int(r["blocks_decoded"]) == info["blocks"], (kind, i)
                assert bool(r["flags"] & cz.RESULT_FINISHED), (kind, i)
            _, with_lits = c.last_prepass_counts(n)
            assert with_lits == n
        finally:
            c.close()


def test_dictionary_history_is_taken_from_the_dictionary(cz, ctx):
    """dict.bin's repeat offsets equal the reset default (1, 4, 8), so nothing distinguished "history from the
    dictionary" from "history reset".  dict_hist.bin carries (21, 7, 96) and the hist_* frames start with repeat-offset codes:
    block level (init_from_dict), batch (cz_context_set_dictionary) without and with the pre-pass."""
    import glob
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dict")
    raw = open(os.path.join(d, "dict_hist.bin"), "rb").read()
    pairs = [(open(z, "rb").read(), open(z[:-4] + ".orig", "rb").read()) for z in sorted(glob.glob(os.path.join(d, "hist_*.zst")))]
    assert len(pairs) >= 2
    gd = cz.Dictionary(ctx, raw)
    assert gd.offset_hist == (21, 7, 96)
    for z, orig in pairs:
        assert cz.decode_frame_with_dict(z, gd, ctx) == orig
    for prepass in (False, True):
        ctx.set_chain_arena((8 << 20) if prepass else 0)
        ctx.set_literal_arena((4 << 20) if prepass else 0)
        ctx.set_dictionary(gd)
        try:
            got = cz.decode_batch_host([z for z, _ in pairs] * 3, [len(o) + 32 for _, o in pairs] * 3, ctx)
            for i, ((r, out), (_, orig)) in enumerate(zip(got, pairs * 3)):
                assert int(r["status"]) == 0 and out == orig, (prepass, i, cz.status.name(r["status"]))
        finally:
            ctx.set_dictionary(None)
            ctx.set_chain_arena(0)
            ctx.set_literal_arena(0)
    gd.close()


def test_treeless_literals_and_large_frames_through_huf_kernel(cz):
    """cz_huf_kernel on real frames: every block of the corpus' large frames is a unit of its own (Treeless sections rebuild the
    tree of the block that defined it); sha256 against the manifest, content checksum verified on the device."""
    files = _large_corpus()
    c = cz.Context(0)
    try:
        c.set_chain_arena(512 << 20, min_sequences=0)
        c.set_literal_arena(128 << 20)
        c.set_verify_checksum(True)
        got = cz.decode_batch_host([z for _, z, _ in files], [e["orig_len"] + 64 for _, _, e in files], c)
        for (name, z, e), (r, out) in zip(files, got):
            assert int(r["status"]) == 0 and hashlib.sha256(out).hexdigest() == e["orig_sha256"], name
            assert r["flags"] & cz.RESULT_CHECKSUM_MATCH, name
        _, with_lits = c.last_prepass_counts(len(files))
        assert with_lits == len(files)
    finally:
        c.close()


def test_decode_batch_multi_deals_frames_over_contexts(cz):
    """cz_decode_batch_multi: one host batch dealt by algorithmic bytes over several contexts — here two
    contexts on the one GPU of the box, with different options — decoded concurrently, outputs and results back in the
    caller's layout: same as the oracle, and the dealing is cz_partition_balanced's."""
    from cairo_zstd_amd import synth
    frames, caps = [], []
    for kind, n in (("mix", 300), ("full_4a", 6), ("raw_rle", 10), ("huf_literals", 4)):
        b = synth.generate(kind, n, first_index=23)
        frames += [b.frame(i) for i in range(n)]
        caps += [int(r) + 16 for r in b.regen]
    for name, z, orig in corpus_pairs(max_orig=8000):
        frames.append(z[: len(z) // 2] if name.endswith("7") else z)       # a few malformed ones among them
        caps.append(len(orig) + 32)
    c0, c1 = cz.Context(0), cz.Context(0)
    try:
        c0.set_chain_arena(128 << 20, min_sequences=0)
        c0.set_literal_arena(64 << 20)
        got, dev = cz.decode_batch_multi(frames, caps, [c0, c1])
        w = np.array([len(f) + c for f, c in zip(frames, caps)], dtype=np.uint64)
        assert (dev == cz.partition_balanced(w, 2)).all() and 0 < int(dev.sum()) < len(frames)
        for i, (fr, cap, (r, out)) in enumerate(zip(frames, caps, got)):
            st, ref, info = oracle.decode_frame(fr, cap=cap)
            assert int(r["status"]) == st, (i, cz.status.name(r["status"]), cz.status.name(st))
            assert st != 0 or out == ref, i
    finally:
        c0.close()
        c1.close()

def test_decode_batch_multi_device_and_gather_to_root(cz):
    """The device-pointer form of the multi-device entry points (no host buffers, no PCIe in the path): the frames dealt with
    cz_partition_balanced, every share put on its context's device, cz_decode_batch_multi_device (only enqueues), then
    cz_gather_to_root — one peer copy per non-root context, on that context's stream behind its decode — and the root's copy of
    every arena compared with the oracle.  Two contexts on the test box's one GPU; a context listed twice is refused."""
    import torch
    from cairo_zstd_amd import synth
    b = synth.generate("mix", 500, first_index=1234)
    n = b.n
    w = (b.length + b.regen).astype(np.uint64)
    part = np.zeros(n, dtype=np.uint32)
    assert cz.lib().cz_partition_balanced(w.ctypes.data, n, 2, part.ctypes.data) == 0
    dev = torch.device("cuda:0")
    ctxs = [cz.Context(0), cz.Context(0)]
    keep = []
    try:
        shares, outs, metas = [], [], []
        for d, c in enumerate(ctxs):
            c.set_chain_arena(128 << 20, min_sequences=0)
            c.set_literal_arena(64 << 20)
            mine = np.nonzero(part == d)[0]
            lens = b.length[mine].astype(np.int64)
            ioff = np.zeros(len(mine), dtype=np.int64); ioff[1:] = np.cumsum((lens[:-1] + 15) & ~15)
            ibuf = np.zeros(int(ioff[-1] + lens[-1]) + 64, dtype=np.uint8)
            for k, i in enumerate(mine):
                ibuf[ioff[k]:ioff[k] + lens[k]] = b.base[int(b.off[i]):int(b.off[i]) + int(b.length[i])]
            caps = b.regen[mine].astype(np.int64)
            ooff = np.zeros(len(mine), dtype=np.int64); ooff[1:] = np.cumsum((caps[:-1] + 255) & ~255)
            total = int(ooff[-1] + caps[-1]) + 256
            t = [torch.from_numpy(x).to(dev) for x in (ibuf, ioff, lens, ooff, caps)]
            t_out = torch.full((total,), 0xA5, dtype=torch.uint8, device=dev)
            t_res = torch.zeros(len(mine) * cz.RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
            keep += t + [t_out, t_res]
            shares.append((t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), len(mine), t_out.data_ptr(), t[3].data_ptr(), t[4].data_ptr(), t_res.data_ptr()))
            outs.append((t_out, t_res)); metas.append((mine, ooff, caps, total))
        torch.cuda.synchronize()
        cz.decode_batch_multi_device(ctxs, shares)
        root_bufs = [None if d == 0 else torch.zeros(metas[d][3], dtype=torch.uint8, device=dev) for d in range(2)]
        torch.cuda.synchronize()
        cz.gather_to_root(ctxs, 0, [o[0].data_ptr() for o in outs], [m[3] for m in metas], [0 if rb is None else rb.data_ptr() for rb in root_bufs])
        ctxs[0].synchronize()
        with pytest.raises(cz.CzError):
            cz.decode_batch_multi_device([ctxs[0], ctxs[0]], shares)
        for d in range(2):
            mine, ooff, caps, total = metas[d]
            got = (outs[0][0] if d == 0 else root_bufs[d]).cpu().numpy()
            res = outs[d][1].cpu().numpy().view(cz.RESULT_DTYPE)
            for k, i in enumerate(mine):
                st, ref, info = oracle.decode_frame(b.frame(int(i)), cap=int(caps[k]))
                assert st == int(res[k]["status"]) == 0, (d, k)
                assert got[int(ooff[k]):int(ooff[k]) + len(ref)].tobytes() == ref, (d, k)
    finally:
        for c in ctxs:
            c.close()


@pytest.mark.parametrize("pipeline", ["single_kernel", "prepass"])
def test_damaged_synthetic_frames_status_and_output_parity(cz, pipeline):
    """1 800 corpus-like multi-block frames, every third with one flipped bit or a cut tail: status for status and byte for byte
    what the oracle says, with the capacities the oracle gets — through the complete kernel and through the pre-pass kernels."""
    from cairo_zstd_amd import synth
    n = 1800
    b = synth.generate("mix", n, first_index=500000, nthreads=8)
    rng = np.random.default_rng(500000)
    for i in range(0, n, 3):
        o, ln = int(b.off[i]), int(b.length[i])
        if rng.integers(0, 4) == 0:
            b.length[i] = max(1, ln - int(rng.integers(1, min(ln, 4000))))
        else:
            b.base[o + int(rng.integers(4, ln))] ^= np.uint8(1 << int(rng.integers(0, 8)))
    frames = [b.frame(i) for i in range(n)]
    caps = [int(r) for r in b.regen]
    o_off, o_cap, o_total = b.out_layout(64)
    ref_out, olen, ost = oracle.decode_batch(b.base, b.off, b.length, o_off, o_cap, int(o_total) + 256, nthreads=8)
    c = cz.Context(0)
    try:
        if pipeline == "prepass":
            c.set_chain_arena(int(b.length.sum()) * 8 + (64 << 20), min_sequences=0)
            c.set_literal_arena(int(b.regen.sum()) + (16 << 20))
        got = cz.decode_batch_host(frames, caps, c)
    finally:
        c.close()
    assert int((np.asarray(ost) != 0).sum()) > n // 12                   # the damage is seen (a flipped literal or raw byte changes bytes, not the status)
    bad = [(i, cz.status.name(r["status"]), cz.status.name(int(ost[i]))) for i, (r, out) in enumerate(got)
           if int(r["status"]) != int(ost[i]) or (int(ost[i]) == 0 and out != ref_out[int(o_off[i]): int(o_off[i]) + int(olen[i])].tobytes())]
    assert not bad, f"{len(bad)} of {n}: {bad[:10]}"


def test_frame_whose_sections_go_back_twice_is_listed_once(cz):
    """A frame of two blocks that cz_huf_kernel both hands back (the D5 block twice: uneven 4-stream splits): the frame is listed
    for cz_decode_frames_kernel once, however many of its sections fail — a batch of nothing but such frames would otherwise
    overrun the list."""
    d = os.path.join(GOLDEN, "vectors")
    z5 = open(os.path.join(d, "d5_uneven_4stream_split.zst"), "rb").read()
    orig = open(os.path.join(d, "d5_uneven_4stream_split.orig"), "rb").read()
    assert len(orig) == 288 and z5[4] == 0x60                           # single segment, 2-byte content size
    two = z5[:5] + (2 * 288 - 256).to_bytes(2, "little") + bytes([z5[7] & 0xFE]) + z5[8:] + z5[7:]
    st, ref, info = oracle.decode_frame(two, cap=1024)
    assert st == 0 and ref == orig * 2 and info["blocks"] == 2
    c = cz.Context(0)
    try:
        c.set_chain_arena(16 << 20, min_sequences=0)
        c.set_literal_arena(16 << 20)
        got = cz.decode_batch_host([two] * 300, [600] * 300, c)
        assert all(int(r["status"]) == 0 and out == ref and int(r["blocks_decoded"]) == 2 for r, out in got)
    finally:
        c.close()


def _libzstd():
    import ctypes
    try:
        L = ctypes.CDLL("libzstd.so.1")
    except OSError:
        return None
    L.ZSTD_compress.restype = ctypes.c_size_t
    L.ZSTD_compress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    L.ZSTD_isError.restype = ctypes.c_uint
    L.ZSTD_isError.argtypes = [ctypes.c_size_t]
    L.ZSTD_compressBound.restype = ctypes.c_size_t
    L.ZSTD_compressBound.argtypes = [ctypes.c_size_t]
    return L


@pytest.mark.parametrize("pipeline", ["single_kernel", "prepass"])
def test_frames_made_by_libzstd_at_run_time(cz, pipeline):
    """Real encoder output, made on the spot by the box's libzstd (skipped without one): text, structured records, noise with
    long repeats and runs, 1 KB .. 1.5 MB (up to a dozen blocks: Repeat / Treeless modes, predefined and RLE tables, long
    matches, repeat offsets), levels 1 / 3 / 9 / 19 — the originals must come back byte for byte, and the oracle must agree."""
    import ctypes
    L = _libzstd()
    if L is None:
        pytest.skip("no libzstd.so.1 on this box")
    rng = np.random.default_rng(2024)
    text = b"".join(orig for name, z, orig in corpus_pairs(max_orig=20000))
    words = [bytes(rng.integers(97, 123, int(rng.integers(2, 9)), dtype=np.uint8)) for _ in range(300)]
    datas = []
    for size in (1000, 20000, 131072, 400000, 1500000):
        datas.append((text * (size // len(text) + 1))[:size])
        datas.append(b" ".join(words[int(i)] for i in rng.integers(0, len(words), size // 5))[:size])
        rec = bytearray()
        while len(rec) < size:
            rec += b"id=%08d;val=%05d;flag=%d\n" % (len(rec), int(rng.integers(0, 99999)), int(rng.integers(0, 2)))
        datas.append(bytes(rec[:size]))
        noise = bytearray(rng.integers(0, 256, size, dtype=np.uint8).tobytes())
        pos = 0
        while pos + 3000 < size:                                         # long repeats at random distances, runs of one byte
            ln, dist = int(rng.integers(4, 2500)), int(rng.integers(1, max(2, min(pos, 200000))))
            if pos > dist:
                for k in range(ln):
                    noise[pos + k] = noise[pos + k - dist]
            pos += ln + int(rng.integers(0, 700))
            if rng.integers(0, 5) == 0:
                run = int(rng.integers(10, 900)); noise[pos:pos + run] = bytes([int(rng.integers(0, 256))]) * run; pos += run
        datas.append(bytes(noise[:size]))
    frames, origs = [], []
    for i, d in enumerate(datas):
        for lvl in ((1, 3, 9, 19) if len(d) <= 400000 else (1, 9)):
            cap = L.ZSTD_compressBound(len(d))
            dst = ctypes.create_string_buffer(cap)
            n = L.ZSTD_compress(dst, cap, d, len(d), lvl)
            assert not L.ZSTD_isError(n)
            frames.append(dst.raw[:n]); origs.append(d)
    caps = [len(o) for o in origs]
    c = cz.Context(0)
    try:
        if pipeline == "prepass":
            c.set_chain_arena(sum(len(f) for f in frames) * 8 + (64 << 20), min_sequences=0)
            c.set_literal_arena(sum(caps) + (16 << 20))
        got = cz.decode_batch_host(frames, caps, c)
    finally:
        c.close()
    bad = [(i, cz.status.name(r["status"]), len(out), len(o)) for i, ((r, out), o) in enumerate(zip(got, origs)) if int(r["status"]) != 0 or out != o]
    assert not bad, bad[:10]
    for fr, o in list(zip(frames, origs))[::7]:                          # and the oracle sees the same
        st, ref, info = oracle.decode_frame(fr, cap=len(o))
        assert st == 0 and ref == o
