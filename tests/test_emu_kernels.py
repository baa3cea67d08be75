"""Runs the UNMODIFIED kernel source (cairo_zstd_amd/csrc/czstd_kernels.hip) on the CPU SIMT
emulator under AddressSanitizer + UBSan (tests/emu) and checks it against the oracle.
GPU sanitizers are not available on the pool, so this is where out-of-bounds accesses, hangs
and barrier races in the kernels get caught before they reach a device.  CPU-only, small."""
import os

import numpy as np
import pytest

import emu_runner
import oracle
from cairo_zstd_amd import status, synth
from conftest import corpus_pairs, raw_frame_with_checksum


def _run_and_compare(frames, caps, chain_bytes=0, exec_kernel=False, lit_bytes=0, wexec_waves=0, verify=True, wexec_auto=False, debug_flags=0):
    res = emu_runner.run(frames, caps, chain_bytes=chain_bytes, exec_kernel=exec_kernel, lit_bytes=lit_bytes, wexec_waves=wexec_waves, verify=verify, wexec_auto=wexec_auto, debug_flags=debug_flags)
    bad = []
    for i, (fr, cap, (r, out)) in enumerate(zip(frames, caps, res)):
        st, ref, info = oracle.decode_frame(fr, cap=cap)
        if st != int(r["status"]):
            bad.append((i, status.name(r["status"]), status.name(st)))
        elif st == 0 and (out != ref or int(r["bytes_consumed"]) != info["consumed"]):
            bad.append((i, "DATA"))
        elif st == 0 and info["has_checksum"] and verify:
            # the emulator runs with content-checksum verification on
            want = oracle.xxh64(ref) & 0xFFFFFFFF
            if not (r["flags"] & 4) or int(r["calculated_checksum"]) != want or bool(r["flags"] & 8) != (want == info["checksum"]):
                bad.append((i, "XXH64", hex(int(r["calculated_checksum"])), hex(want)))
    assert not bad, bad[:10]


def test_emu_corpus_small():
    pairs = corpus_pairs(max_orig=6000)
    _run_and_compare([z for _, z, _ in pairs], [len(o) + 16 for _, _, o in pairs])


def test_emu_synthetic_mix():
    b = synth.generate("mix", 30, first_index=900, nthreads=2)
    keep = [i for i in range(b.n) if b.regen[i] < 60000][:9]
    _run_and_compare([b.frame(i) for i in keep], [int(b.regen[i]) + 8 for i in keep])


def test_emu_malformed_inputs():
    frames, caps = [], []
    for idx, (name, z, orig) in enumerate(corpus_pairs(max_orig=1200)):
        rng = np.random.default_rng(idx)
        muts = [z[: len(z) // 2], z[:-1], z[:5], z[:3], b"", z + b"\x00"]
        for _ in range(4):
            a = bytearray(z)
            a[int(rng.integers(0, len(a)))] ^= 1 << int(rng.integers(0, 8))
            muts.append(bytes(a))
        frames += muts
        caps += [len(orig) * 2 + 4096] * len(muts)
    _run_and_compare(frames, caps)


def test_emu_content_checksum_lengths():
    """XXH64 on the device (src/utils/xxhash64.cairo:31-163): every tail length class, block
    boundaries of the 512-byte staging, one wrong stored checksum."""
    rng = np.random.default_rng(7)
    frames, caps = [], []
    for n in [0, 1, 3, 4, 5, 7, 8, 9, 31, 32, 33, 40, 63, 64, 95, 511, 512, 513, 543, 544, 545, 1023, 1024, 1025, 1536, 5000, 70001]:
        frames.append(raw_frame_with_checksum(rng.integers(0, 256, n, dtype=np.uint8).tobytes()))
        caps.append(n + 8)
    frames.append(raw_frame_with_checksum(b"abc" * 400, corrupt=True))
    caps.append(1300)
    _run_and_compare(frames, caps)


def test_emu_chain_prepass():
    """cz_chain_kernel + the record-driven path of cz_decode_frames_kernel under ASan/UBSan: corpus frames
    (multi-block, Repeat/RLE/predefined tables), synthetic frames, malformed frames (the pre-pass must
    leave them to the decoder, which reports the reference's status), an arena that is far too small, and the
    huff0 / tile kernels feeding the decode kernel, with room in the literal arena and without."""
    frames, caps = [], []
    for name, z, orig in corpus_pairs(max_orig=1000):
        frames.append(z)
        caps.append(len(orig) + 16)
    b = synth.generate("mix", 24, first_index=4242, nthreads=2)
    keep = [i for i in range(b.n) if b.regen[i] < 30000][:2]
    frames += [b.frame(i) for i in keep]
    caps += [int(b.regen[i]) + 8 for i in keep]
    for idx, (name, z, orig) in enumerate(corpus_pairs(max_orig=350)):
        rng = np.random.default_rng(100 + idx)
        a = bytearray(z)
        a[int(rng.integers(0, len(a)))] ^= 1 << int(rng.integers(0, 8))
        frames.append(bytes(a))
        caps.append(len(orig) * 2 + 4096)
        frames.append(z[: len(z) // 2])
        caps.append(len(orig) * 2 + 4096)
    _run_and_compare(frames, caps, chain_bytes=8 << 20, lit_bytes=4 << 20)
    assert "frames have their literals done" in emu_runner.run.last_stderr
    _run_and_compare(frames[::9], caps[::9], chain_bytes=8 << 20)
    _run_and_compare(frames[:10], caps[:10], chain_bytes=4096, lit_bytes=6000)


def test_emu_scan_lists_over_several_waves():
    """cz_scan_kernel counts per class in LDS and shares the class ranges out per WAVE (scan_wave): a batch of more than two
    waves of frames — tiny Raw frames with small corpus frames (several blocks, Huffman literals, sequences) spread over the
    waves — through the whole pre-pass pipeline."""
    small = [(z, len(orig) + 16) for name, z, orig in corpus_pairs(max_orig=400)][:10]
    rng = np.random.default_rng(77)
    frames, caps = [], []
    for i in range(134):
        if i % 13 == 5:
            z, cap = small[(i // 13) % len(small)]
            frames.append(z); caps.append(cap)
        else:
            n = int(rng.integers(1, 40))
            frames.append(raw_frame_with_checksum(rng.integers(0, 256, n, dtype=np.uint8).tobytes())); caps.append(n + 8)
    _run_and_compare(frames, caps, chain_bytes=8 << 20, lit_bytes=4 << 20, exec_kernel=True)


def test_emu_execute_frames_kernel():
    """cz_execute_frames_kernel (czstd_kernels.hip compiled without its decoders: the frames the pre-pass finished) under
    ASan/UBSan, followed by cz_decode_frames_kernel on the frames it left: corpus frames, synthetic frames, malformed
    frames (it must hand them over), with the content checksum verified on the device."""
    frames, caps = [], []
    for name, z, orig in corpus_pairs(max_orig=1200):
        frames.append(z)
        caps.append(len(orig) + 16)
    b = synth.generate("mix", 24, first_index=4242, nthreads=2)
    keep = [i for i in range(b.n) if b.regen[i] < 30000][:2]
    frames += [b.frame(i) for i in keep]
    caps += [int(b.regen[i]) + 8 for i in keep]
    for idx, (name, z, orig) in enumerate(corpus_pairs(max_orig=450)):
        rng = np.random.default_rng(300 + idx)
        a = bytearray(z)
        a[int(rng.integers(0, len(a)))] ^= 1 << int(rng.integers(0, 8))
        frames.append(bytes(a))
        caps.append(len(orig) * 2 + 4096)
    _run_and_compare(frames, caps, chain_bytes=8 << 20, lit_bytes=4 << 20, exec_kernel=True)
    err = emu_runner.run.last_stderr
    assert "frames finished by cz_execute_frames_kernel" in err
    done = int(err.split("EMU_EXEC: ")[1].split()[0])
    assert 0 < done < len(frames), (done, len(frames))                  # some frames took it, the malformed ones did not

def test_emu_execute_frames_kernel_chunk_buffer_long_runs():
    """The chunk buffer of cz_execute_frames_kernel at 4 waves per SIMD (3 KiB, runs up to 128 bytes: literals fetched by the wave
    into the staging area, far matches four 16-byte pieces a turn, near matches 8 bytes a step): corpus files of 12-40 KB,
    whose chunks run to a few KB, under ASan/UBSan."""
    pairs = [p for p in corpus_pairs(max_orig=40000) if len(p[2]) >= 12000][:4]
    assert len(pairs) >= 2
    _run_and_compare([z for _, z, _ in pairs], [len(o) + 16 for _, _, o in pairs], chain_bytes=8 << 20, lit_bytes=4 << 20, exec_kernel=True)
    done = int(emu_runner.run.last_stderr.split("EMU_EXEC: ")[1].split()[0])
    assert done == len(pairs), (done, len(pairs))


def test_emu_wexec_kernel():
    """cz_wexec_kernel (czstd_wexec.hip: several waves per frame, LDS window, look-back over chunk entries, bitmap of final bytes)
    under ASan/UBSan with 4 waves per workgroup, ahead of cz_execute_frames_kernel and cz_decode_frames_kernel: corpus frames,
    corpus-like frames (several blocks, sources in earlier blocks, long matches and literal runs, repeat offsets of every kind), a
    frame whose second block regenerates more than the window holds, and damaged frames, which it hands on."""
    frames, caps = [], []
    for name, z, orig in corpus_pairs(max_orig=2500):
        frames.append(z)
        caps.append(len(orig) + 16)
    b = synth.generate("mix", 30, first_index=0, nthreads=2)
    keep = [i for i in range(b.n) if b.regen[i] < 30000][:6] + [13]     # 13: 200 KB, a block above 128 KiB
    frames += [b.frame(i) for i in keep]
    caps += [int(b.regen[i]) + 8 for i in keep]
    for idx, (name, z, orig) in enumerate(corpus_pairs(max_orig=700)):
        rng = np.random.default_rng(900 + idx)
        a = bytearray(z)
        a[int(rng.integers(0, len(a)))] ^= 1 << int(rng.integers(0, 8))
        frames.append(bytes(a))
        caps.append(len(orig) * 2 + 4096)
    _run_and_compare(frames, caps, chain_bytes=16 << 20, lit_bytes=8 << 20, exec_kernel=True, wexec_waves=4, verify=False)
    err = emu_runner.run.last_stderr
    listed, done = int(err.split("EMU_WEXEC: ")[1].split()[0]), int(err.split("frames listed, ")[1].split()[0])
    assert listed > 20 and 10 < done <= listed, (listed, done)


def test_emu_wexec_kernel_large_frames_of_a_near_offset_batch_and_chunks_longer_than_the_window():
    """The execute stage as it is arranged on a near-offset batch (decided from the code tables, as on the device): the batch's large
    frames (the emulator build puts the mark at 600 sequences) on cz_wexec_kernel, all others on cz_execute_frames_kernel.  Three of
    the large frames have a chunk of 64 sequences that regenerates more than the 128 KiB window (wx_slow_chunk)."""
    frames, caps = [], []
    for first in (8265, 9021, 1750):
        b = synth.generate("mix", 1, first_index=first, nthreads=1)
        frames.append(b.frame(0))
        caps.append(int(b.regen[0]) + 8)
    b = synth.generate("mix", 30, first_index=0, nthreads=2)
    keep = [i for i in range(b.n) if b.regen[i] < 30000][:6]
    frames += [b.frame(i) for i in keep]
    caps += [int(b.regen[i]) + 8 for i in keep]
    _run_and_compare(frames, caps, chain_bytes=16 << 20, lit_bytes=8 << 20, exec_kernel=True, wexec_waves=16, verify=False, wexec_auto=True)
    err = emu_runner.run.last_stderr
    done_wx, not_handed_on = int(err.split("frames listed, ")[1].split()[0]), int(err.split("EMU_EXEC: ")[1].split()[0])
    assert done_wx == 3 and not_handed_on == len(frames), (done_wx, not_handed_on, len(frames))   # the large ones there, nothing left to cz_decode_frames_kernel


def test_emu_wexec_kernel_bounded_waits_poisoned_lookback_entry():
    """Every device-side wait of cz_wexec_kernel is bounded (WX_SPIN_LIMIT).  CZ_DEBUG_WX_POISON: chunk 2 of every block never
    publishes its look-back entry, so every chunk behind it polls until the bound, sets ctl.err, and the frame is handed to
    cz_decode_frames_kernel — which decodes it like any other frame.  Frames of at most two chunks are not affected."""
    import cairo_zstd_amd as cz
    pairs = [p for p in corpus_pairs(max_orig=30000) if len(p[2]) >= 6000][:3] + corpus_pairs(max_orig=1500)[:3]   # blocks of several chunks, and small ones
    frames, caps = [z for _, z, _ in pairs], [len(o) + 16 for _, _, o in pairs]
    _run_and_compare(frames, caps, chain_bytes=16 << 20, lit_bytes=8 << 20, exec_kernel=True, wexec_waves=4, verify=False)
    err = emu_runner.run.last_stderr
    listed, done = int(err.split("EMU_WEXEC: ")[1].split()[0]), int(err.split("frames listed, ")[1].split()[0])
    _run_and_compare(frames, caps, chain_bytes=16 << 20, lit_bytes=8 << 20, exec_kernel=True, wexec_waves=4, verify=False, debug_flags=cz.DEBUG_WX_POISON)
    err = emu_runner.run.last_stderr
    listed_p, done_p = int(err.split("EMU_WEXEC: ")[1].split()[0]), int(err.split("frames listed, ")[1].split()[0])
    assert listed_p == listed and done_p < done, (listed, done, listed_p, done_p)     # the frames with a third chunk were given up at the bound ...
    assert int(err.split("EMU_EXEC: ")[1].split()[0]) == len(frames) - (done - done_p)  # ... and went to cz_decode_frames_kernel, once each


def test_emu_frames_handed_back_by_huf_kernel_are_listed_once_beside_wexec_kernel():
    """a frame listed for cz_wexec_kernel whose literals cz_huf_kernel hands back (the D5 block — an uneven 4-stream
    split — with a sequence behind it) used to be put on the fall-back list by BOTH execute kernels.  Now whoever sets CZ_PRE_LISTED first lists it: the
    emulator fails the run when a frame is on the list twice or the list has more than n entries."""
    import json
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vectors")
    man = json.load(open(os.path.join(d, "manifest_r5.json")))
    name = "d5_uneven_split_with_sequences.zst"                          # the D5 block + one sequence: chain records, so both execute kernels have it
    z5s = open(os.path.join(d, name), "rb").read()
    frames, caps = [z5s] * 12, [man[name]["orig_len"] + 16] * 12
    for _, z, orig in corpus_pairs(max_orig=1500)[:4]:
        frames.append(z)
        caps.append(len(orig) + 16)
    for exec_first in ("0", "1"):                                       # either execute kernel may meet a frame first
        os.environ["EMU_EXEC_FIRST"] = exec_first
        try:
            _run_and_compare(frames, caps, chain_bytes=8 << 20, lit_bytes=4 << 20, exec_kernel=True, wexec_waves=4, verify=False)
        finally:
            del os.environ["EMU_EXEC_FIRST"]
        err = emu_runner.run.last_stderr
        assert int(err.split("EMU_LIT: ")[1].split()[0]) <= len(frames) - 12, err      # cz_huf_kernel handed the D5 frames back
        assert int(err.split("EMU_WEXEC: ")[1].split()[0]) >= 12, err                  # ... which were listed for cz_wexec_kernel
        assert int(err.split("EMU_EXEC: ")[1].split()[0]) <= len(frames) - 12, err     # and they are on the fall-back list (once: the emulator checks)


def test_emu_split_chain_prepass_and_early_execute_launches():
    """cz_context_set_early_execute(1) under ASan/UBSan: the chain pre-pass as two launches (the emulator build calls a block of 256
    sequences large), the large blocks of the batch's large frames published one by one; the early launch of cz_execute_frames_kernel on
    the frames without a large block, the early launch of cz_wexec_kernel on the large frames, each large block behind its chain's flag;
    then the later launches, which find those frames claimed.  EMU_SPLIT=2 runs cz_wexec_kernel's early launch BEFORE the large blocks'
    chains: every wait for a flag runs into its bound and the frames go to cz_decode_frames_kernel."""
    frames, caps = [], []
    for first in (8265, 1750):
        b = synth.generate("mix", 1, first_index=first, nthreads=1)
        frames.append(b.frame(0))
        caps.append(int(b.regen[0]) + 8)
    b = synth.generate("mix", 30, first_index=0, nthreads=2)
    keep = [i for i in range(b.n) if b.regen[i] < 30000][:6]
    frames += [b.frame(i) for i in keep]
    caps += [int(b.regen[i]) + 8 for i in keep]
    for _, z, orig in corpus_pairs(max_orig=1500)[:4]:
        frames.append(z)
        caps.append(len(orig) + 16)
    for split, want_wx in (("1", 2), ("2", 0)):
        os.environ["EMU_SPLIT"] = split
        try:
            _run_and_compare(frames, caps, chain_bytes=16 << 20, lit_bytes=8 << 20, exec_kernel=True, wexec_waves=16, verify=False, wexec_auto=True)
        finally:
            del os.environ["EMU_SPLIT"]
        err = emu_runner.run.last_stderr
        done_wx = int(err.split("frames listed, ")[1].split()[0])
        assert done_wx == want_wx, (split, done_wx, err)


def test_emu_d2_weight_log_10_unsupported():
    """DESIGN.md D2, device side: CZ_E_UNSUPPORTED for a Huffman-weight FSE accuracy log above 9."""
    import os
    from conftest import GOLDEN
    z = open(os.path.join(GOLDEN, "vectors", "d2_weight_log10.zst"), "rb").read()
    for chain in (0, 1 << 20):
        res = emu_runner.run([z], [64], chain_bytes=chain)
        assert int(res[0][0]["status"]) == status.CZ_E_UNSUPPORTED


def test_emu_dictionary_frames():
    """cz_dict_setup_kernel and the dictionary arm of the match copy under ASan/UBSan: the committed dictionary frames, every
    frame of the batch started from the dictionary (what cz_context_set_dictionary does), with and without the pre-pass."""
    import glob
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dict")
    pairs = [(open(z, "rb").read(), open(z[:-4] + ".orig", "rb").read()) for z in sorted(glob.glob(os.path.join(d, "frame_*.zst")))[:5]]
    for chain_bytes in (0, 8 << 20):
        got = emu_runner.run([z for z, _ in pairs], [len(o) + 16 for _, o in pairs], dict_path=os.path.join(d, "dict.bin"),
                             chain_bytes=chain_bytes, lit_bytes=(4 << 20) if chain_bytes else 0)
        for i, ((r, out), (_, orig)) in enumerate(zip(got, pairs)):
            assert int(r["status"]) == 0, (i, int(r["status"]))
            assert out == orig, i


def test_emu_prepass_kernels_and_divergence_vectors():
    """cz_huf_kernel (workgroups of several waves) and cz_tile_kernel under ASan/UBSan: Raw / RLE frames, literals-only frames
    (the scan writes their result records), the D1 vector (direct weights, unequal nibbles), the D5 vector (uneven 4-stream
    split: cz_huf_kernel hands it back), corpus frames, and lists that are too short."""
    import json
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vectors")
    man = json.load(open(os.path.join(d, "manifest_r3.json")))
    frames, caps = [], []
    for name in ("d1_unequal_direct_weights.zst", "d5_uneven_4stream_split.zst"):
        frames.append(open(os.path.join(d, name), "rb").read())
        caps.append(man[name]["orig_len"] + 16)
    b = synth.generate("raw_rle", 2, first_index=5, nthreads=2)
    frames += [b.frame(i) for i in range(2)]
    caps += [int(r) + 8 for r in b.regen]
    for name, z, orig in corpus_pairs(max_orig=9000)[-8:]:              # multi-block frames with Huffman (also Treeless) literals
        frames.append(z)
        caps.append(len(orig) + 16)
    _run_and_compare(frames, caps, chain_bytes=8 << 20, lit_bytes=4 << 20, exec_kernel=True)
    err = emu_runner.run.last_stderr
    assert "frames have their literals done" in err
    assert int(err.split("EMU_LIT: ")[1].split()[0]) == len(frames) - 1     # all but the D5 frame
    os.environ["EMU_HUF1"] = "1"                                            # the same batch with every literals section on cz_huf1_kernel
    try:
        _run_and_compare(frames, caps, chain_bytes=8 << 20, lit_bytes=4 << 20, exec_kernel=True)
        assert int(emu_runner.run.last_stderr.split("EMU_LIT: ")[1].split()[0]) == len(frames)   # its stream decoder redoes an uneven split itself: D5 stays
    finally:
        del os.environ["EMU_HUF1"]
    # frames whose TWO literals sections both go back (the D5 block twice): each is listed for the decode kernel once
    z5 = frames[1]
    two = z5[:5] + (2 * 288 - 256).to_bytes(2, "little") + bytes([z5[7] & 0xFE]) + z5[8:] + z5[7:]
    _run_and_compare([two] * 5, [2 * 288 + 16] * 5, chain_bytes=8 << 20, lit_bytes=4 << 20, exec_kernel=True)
    os.environ["EMU_SEGS"] = "3"                                            # lists far too short: most frames are not pre-passed at all
    try:
        _run_and_compare(frames[:6], caps[:6], chain_bytes=8 << 20, lit_bytes=4 << 20, exec_kernel=True)
    finally:
        del os.environ["EMU_SEGS"]
