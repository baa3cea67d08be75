"""Pins the CPU oracle (oracle/zstd_oracle.c) to the reference's own known answers.

Vectors (data only) come from the reference's tests:
  * src/tests/bit_reader.cairo:14-16,56-58   16-byte constant + chunk schedule
  * src/tests/utils.cairo:134-150            14 XXH64 known answers (seed 0)
  * src/decoding/sequence_section_decoder.cairo:707-737   5 predefined-LL table entries
  * data/decode_corpus                        (original, .zst) pairs -> tests/golden/decode_corpus
"""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

import oracle
from conftest import GOLDEN, REFERENCE_CORPUS
from cairo_zstd_amd import status

CONST16 = bytes.fromhex("C141080000ECC8964279D4BCF72CD548")  # ba.append_word(0xC141..., 16): big-endian bytes


def _schedule():
    """x += 3; num_bits = x % 16, clipped to 128 total (bit_reader.cairo:23-39)."""
    widths, read, x = [], 0, 0
    while read < 128:
        x = (x + 3) & 0xFF
        n = x % 16
        if read > 128 - n:
            n = 128 - read
        widths.append(n)
        read += n
    return widths


def test_bitreader_reversed_kat():
    widths = _schedule()
    w = np.array(widths, dtype=np.uint8)
    src = np.frombuffer(CONST16, dtype=np.uint8)
    out = np.zeros(len(widths), dtype=np.uint64)
    rem = C.c_int64()
    assert oracle.lib().czo_kat_reverse_reads(src.ctypes.data, 16, w.ctypes.data, len(widths),
                                              out.ctypes.data_as(oracle.u64p), C.byref(rem)) == 0
    acc, read = 0, 0
    for n, v in zip(widths, out.tolist()):
        read += n
        acc |= int(v) << (128 - read)
    assert acc == 0x48D52CF7BCD4794296C8EC00000841C1  # num_rev, bit_reader.cairo:16
    assert rem.value == 0


def test_bitreader_forward_kat():
    widths = _schedule()
    w = np.array(widths, dtype=np.uint8)
    src = np.frombuffer(CONST16, dtype=np.uint8)
    out = np.zeros(len(widths), dtype=np.uint64)
    assert oracle.lib().czo_kat_forward_reads(src.ctypes.data, 16, w.ctypes.data, len(widths),
                                              out.ctypes.data_as(oracle.u64p)) == 0
    acc, read = 0, 0
    for n, v in zip(widths, out.tolist()):
        acc |= int(v) << read
        read += n
    assert acc == 0x48D52CF7BCD4794296C8EC00000841C1  # bit_reader.cairo:58


def test_bitreader_reversed_past_start_zero_fill():
    # bit_reader_reverse.cairo:147-159: reads straddling / past the start are zero-extended
    src = np.frombuffer(bytes([0b10110001]), dtype=np.uint8)
    w = np.array([5, 6, 7], dtype=np.uint8)
    out = np.zeros(3, dtype=np.uint64)
    rem = C.c_int64()
    oracle.lib().czo_kat_reverse_reads(src.ctypes.data, 1, w.ctypes.data, 3, out.ctypes.data_as(oracle.u64p), C.byref(rem))
    assert out.tolist() == [0b10110, 0b001000, 0]
    assert rem.value == 8 - 18


XXH_KATS = [  # src/tests/utils.cairo:134-150
    (0xEF46DB3751D8E999, b""), (0xD24EC4F1A98C6E5B, b"a"), (0x65F708CA92D04A61, b"ab"),
    (0x44BC2CF5AD770999, b"abc"), (0xDE0327B0D25D92CC, b"abcd"), (0x07E3670C0C8DC7EB, b"abcde"),
    (0xFA8AFD82C423144D, b"abcdef"), (0x1860940E2902822D, b"abcdefg"), (0x3AD351775B4634B7, b"abcdefgh"),
    (0x27F1A34FDBB95E13, b"abcdefghi"), (0xD6287A1DE5498BB2, b"abcdefghij"),
    (0xBF2CD639B4143B80, b"abcdefghijklmnopqrstuvwxyz012345"),
    (0x64F23ECF1609B766, b"abcdefghijklmnopqrstuvwxyz0123456789"),
    (0xC5A8B11443765630,
     b"Lorem ipsum dolor sit amet, consectetur adipiscing elit, sed do eiusmod tempor incididunt ut labore et dolore "
     b"magna aliqua. Ut enim ad minim veniam, quis nostrud exercitation ullamco laboris nisi ut aliquip ex ea commodo "
     b"consequat. Duis aute irure dolor in reprehenderit in voluptate velit esse cillum dolore eu fugiat nulla "
     b"pariatur. Excepteur sint occaecat cupidatat non proident, sunt in culpa qui officia deserunt mollit anim id "
     b"est laborum."),
]


@pytest.mark.parametrize("want,data", XXH_KATS)
def test_xxh64_kats(want, data):
    assert oracle.xxh64(data) == want


def _fse_table(which, acc_log=0, probs=None):
    n = 1 << 12
    sym = np.zeros(n, dtype=np.uint8)
    nb = np.zeros(n, dtype=np.uint8)
    bl = np.zeros(n, dtype=np.uint32)
    size = C.c_uint32()
    p = np.array(probs if probs is not None else [0], dtype=np.int32)
    st = oracle.lib().czo_kat_fse_table(which, acc_log, p.ctypes.data, len(p), sym.ctypes.data, nb.ctypes.data,
                                        bl.ctypes.data, C.byref(size))
    assert st == 0
    s = size.value
    return sym[:s], nb[:s], bl[:s]


def test_predefined_ll_table_entries():
    # sequence_section_decoder.cairo:707-737
    sym, nb, bl = _fse_table(0)
    assert len(sym) == 64
    for idx, (s, n, b) in {0: (0, 4, 0), 19: (27, 6, 0), 39: (25, 4, 16), 60: (35, 6, 0), 59: (24, 5, 32)}.items():
        assert (sym[idx], nb[idx], bl[idx]) == (s, n, b), idx


def test_predefined_tables_are_valid_fse():
    # every state of a symbol with count c must cover [0, size) exactly once via (baseline, 2^nbits)
    for which, size in ((0, 64), (1, 32), (2, 64)):
        sym, nb, bl = _fse_table(which)
        assert len(sym) == size
        cover = {}
        for s, n, b in zip(sym.tolist(), nb.tolist(), bl.tolist()):
            cover.setdefault(s, []).append((b, 1 << n))
        for s, spans in cover.items():
            spans.sort()
            pos = 0
            for b, w in spans:
                assert b == pos
                pos += w
            assert pos == size


def test_corpus_golden_subset(golden_corpus):
    """_test_decode (src/tests/decoding.cairo:4-21) on every committed pair."""
    assert len(golden_corpus) == 69
    for name, z, orig in golden_corpus:
        st, out, info = oracle.decode_frame(z, cap=len(orig) + 64)
        assert st == 0, name
        assert out == orig, name
        assert info["consumed"] == len(z)
        assert info["has_checksum"] and info["checksum"] == (oracle.xxh64(orig) & 0xFFFFFFFF), name


def test_corpus_manifest_matches_fixtures(golden_corpus):
    man = json.load(open(os.path.join(GOLDEN, "decode_corpus_manifest.json")))
    assert len(man) == 100
    for name, z, orig in golden_corpus:
        m = man[name]
        assert m["orig_sha256"] == hashlib.sha256(orig).hexdigest()
        assert m["zst_sha256"] == hashlib.sha256(z).hexdigest()
        assert int(m["xxh64"], 16) == oracle.xxh64(orig)


@pytest.mark.skipif(not os.path.isdir(REFERENCE_CORPUS), reason="full reference corpus only exists in the build container")
def test_corpus_full_reference_tree():
    man = json.load(open(os.path.join(GOLDEN, "decode_corpus_manifest.json")))
    for name, m in man.items():
        z = open(os.path.join(REFERENCE_CORPUS, name + ".zst"), "rb").read()
        st, out, info = oracle.decode_frame(z, cap=m["orig_len"] + 64)
        assert st == 0, name
        assert hashlib.sha256(out).hexdigest() == m["orig_sha256"], name
        assert info["checksum"] == int(m["xxh64"], 16) & 0xFFFFFFFF


def test_frame_decoder_object_matches_one_shot(golden_corpus):
    """FrameDecoder state machine: decode_blocks(UptoBlocks 1) + collect == one-shot; the
    calculated checksum only becomes valid after draining (decode_buffer.cairo:162,181)."""
    for name, z, orig in golden_corpus[:25]:
        fd = oracle.FrameDecoder()
        st, hl, _ = fd.new(z)
        assert st == 0
        pos, out = hl, b""
        while not fd.is_finished():
            st, used, fin = fd.decode_blocks(z[pos:], oracle.FrameDecoder.UPTO_BLOCKS, 1)
            assert st == 0, name
            pos += used
            got = fd.collect(cap=len(orig) + 64)
            if got:
                out += got
        rest = fd.collect(cap=len(orig) + 64)
        if rest:
            out += rest
        assert out == orig, name
        assert pos == len(z) and fd.bytes_read_from_source() == len(z)
        assert fd.get_checksum_from_data() == fd.get_calculated_checksum()


def test_decode_from_to_streaming(golden_corpus):
    """decode_from_to (frame_decoder.cairo:245-326) fed in 1000-byte slices."""
    for name, z, orig in golden_corpus[30:45]:
        fd = oracle.FrameDecoder()
        st, hl, _ = fd.new(z)
        assert st == 0
        pos, out = hl, b""
        guard = 0
        while not fd.is_finished() and guard < 10000:
            guard += 1
            chunk = z[pos:pos + 200000]
            st, r, got = fd.decode_from_to(chunk, cap=len(orig) + 64)
            assert st == 0, name
            pos += r
            out += got
        st, r, got = fd.decode_from_to(b"", cap=len(orig) + 64)
        out += got
        assert out == orig, name


def test_oracle_large_corpus_frames_vs_manifest():
    """The 31 large corpus frames (compressed side committed under tests/golden/decode_corpus_large): the oracle's
    output matches the sha256 / XXH64 the manifest recorded from the reference's originals."""
    import hashlib
    import json
    from conftest import GOLDEN
    m = json.load(open(os.path.join(GOLDEN, "decode_corpus_manifest.json")))
    n = 0
    for name in sorted(m):
        e = m[name]
        if e["committed"]:
            continue
        z = open(os.path.join(GOLDEN, "decode_corpus_large", name + ".zst"), "rb").read()
        assert hashlib.sha256(z).hexdigest() == e["zst_sha256"]
        st, out, info = oracle.decode_frame(z, cap=e["orig_len"] + 16)
        assert st == 0 and len(out) == e["orig_len"], name
        assert hashlib.sha256(out).hexdigest() == e["orig_sha256"] and f"{oracle.xxh64(out):016x}" == e["xxh64"], name
        assert info["has_checksum"] and info["checksum"] == int(e["xxh64"], 16) & 0xFFFFFFFF, name
        n += 1
    assert n == 31


def test_oracle_accepts_huffman_weight_fse_log_10():
    """DESIGN.md D2, reference side: max_log 100 for the FSE table of the Huffman weights (src/huff0/huff0_decoder.cairo:176)."""
    from conftest import GOLDEN
    d = os.path.join(GOLDEN, "vectors")
    z, want = open(os.path.join(d, "d2_weight_log10.zst"), "rb").read(), open(os.path.join(d, "d2_weight_log10"), "rb").read()
    st, out, _ = oracle.decode_frame(z, cap=64)
    assert st == 0 and out == want


# ---------------------------------------------------------------- dictionaries
DICT_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dict")


def _dict_cases():
    import glob
    raw = open(os.path.join(DICT_DIR, "dict.bin"), "rb").read()
    frames = []
    for z in sorted(glob.glob(os.path.join(DICT_DIR, "frame_*.zst"))):
        frames.append((os.path.basename(z), open(z, "rb").read(), open(z[:-4] + ".orig", "rb").read()))
    return raw, frames


def test_oracle_dictionary_decode_dict_fields():
    """DictionaryTrait::decode_dict (src/decoding/dictionary.cairo:35-91) on the committed trained dictionary
    (scripts/gen_dict_vectors.py): magic, id, four tables, three repeat offsets, content."""
    raw, _ = _dict_cases()
    d = oracle.Dictionary(raw)
    assert d.status == 0
    assert d.info["id"] == int.from_bytes(raw[4:8], "little")
    assert d.info["content_off"] + d.info["content_len"] == len(raw) and d.info["content_len"] > 0
    assert (d.info["hist0"], d.info["hist1"], d.info["hist2"]) == tuple(
        int.from_bytes(raw[d.info["content_off"] - 12 + 4 * k: d.info["content_off"] - 8 + 4 * k], "little") for k in range(3))
    assert 1 <= d.info["huf_max_bits"] <= 11 and d.info["ll_log"] <= 9 and d.info["ml_log"] <= 9 and d.info["of_log"] <= 8
    bad = bytearray(raw); bad[0] ^= 1
    e = oracle.Dictionary(bytes(bad))
    assert e.status == status.CZ_E_DICT_BAD_MAGIC and e.detail == int.from_bytes(bad[0:4], "little")
    assert oracle.Dictionary(raw[:7]).status == status.CZ_E_DICT_TRUNCATED
    assert oracle.Dictionary(raw[: d.info["content_off"] - 5]).status == status.CZ_E_DICT_TRUNCATED
    assert oracle.Dictionary(raw[:40]).status != 0                       # inside the tables: a table error or a truncation


def test_oracle_decodes_frames_compressed_with_a_dictionary():
    """init_from_dict (src/decoding/scratch.cairo:60-65) + the dictionary arm of DecodeBuffer::repeat
    (src/decoding/decode_buffer.cairo:65-93): frames made by libzstd's ZSTD_compress_usingDict decode to their originals;
    without the dictionary the same frames fail the way the reference does."""
    raw, frames = _dict_cases()
    d = oracle.Dictionary(raw)
    assert d.status == 0
    for name, z, orig in frames:
        st, out = oracle.decode_frame_with_dict(z, d, cap=len(orig) + 64)
        assert st == 0 and out == orig, name
        st2 = oracle.decode_frame(z, cap=len(orig) + 64)[0]
        assert st2 != 0, f"{name} decodes without its dictionary"


# ---------------------------------------------------------------- round-3 vectors (scripts/gen_r3_vectors.py, gen_dict_vectors.py)
def _r3_vectors():
    import json
    d = os.path.join(GOLDEN, "vectors")
    man = json.load(open(os.path.join(d, "manifest_r3.json")))
    return [(n, open(os.path.join(d, n), "rb").read(), man[n]) for n in sorted(man)]


def test_oracle_r3_vectors_vs_manifest():
    """Real encoder output (libzstd levels 1 / 3 / 19, a frame with OF in Repeat mode) and the two hand-built divergence vectors:
    the oracle reproduces what libzstd decoded when the files were made (length, sha256, XXH64 in the manifest)."""
    import hashlib
    for name, z, e in _r3_vectors():
        st, out, info = oracle.decode_frame(z, cap=e["orig_len"] + 64)
        assert st == 0 and len(out) == e["orig_len"] and info["consumed"] == len(z), name
        assert hashlib.sha256(out).hexdigest() == e["orig_sha256"] and f"{oracle.xxh64(out):016x}" == e["xxh64"], name


def test_d1_direct_weight_nibble_order_is_observable():
    """with direct weights of unequal nibbles the zstd order (even index = high nibble; what libzstd decoded to
    .orig) and the reference's literal test at src/huff0/huff0_decoder.cairo:302 (low nibble for indices 0 and 1 only) give
    different tables — the oracle's switch reproduces the literal behaviour."""
    d = os.path.join(GOLDEN, "vectors")
    z = open(os.path.join(d, "d1_unequal_direct_weights.zst"), "rb").read()
    orig = open(os.path.join(d, "d1_unequal_direct_weights.orig"), "rb").read()
    st, out, _ = oracle.decode_frame(z, cap=1024)
    assert st == 0 and out == orig
    oracle.lib().czo_set_d1_reference_nibbles(1)
    try:
        st_ref, out_ref, _ = oracle.decode_frame(z, cap=1024)
    finally:
        oracle.lib().czo_set_d1_reference_nibbles(0)
    assert st_ref != 0 or out_ref != orig
    st2, out2, _ = oracle.decode_frame(z, cap=1024)                      # the switch is off again
    assert st2 == 0 and out2 == orig


def test_d5_uneven_four_stream_split_is_accepted():
    """four huff0 streams of 144 / 48 / 48 / 48 symbols (not ceil(288 / 4) each): the reference concatenates what
    every stream yields and checks the total only (src/decoding/literals_section_decoder.cairo:112-115, :172-178)."""
    d = os.path.join(GOLDEN, "vectors")
    z = open(os.path.join(d, "d5_uneven_4stream_split.zst"), "rb").read()
    st, out, _ = oracle.decode_frame(z, cap=1024)
    assert st == 0 and out == open(os.path.join(d, "d5_uneven_4stream_split.orig"), "rb").read()


def test_oracle_dictionary_history_fixtures():
    """dict_hist.bin = dict.bin with repeat offsets (21, 7, 96): the hist_* frames begin with repeat-offset codes, so they decode
    to their originals from the dictionary's history (dictionary.cairo:81-85, scratch.cairo:60-65) and NOT from (1, 4, 8)."""
    import glob
    d = os.path.join(GOLDEN, "dict")
    dh, d0 = oracle.Dictionary(open(os.path.join(d, "dict_hist.bin"), "rb").read()), oracle.Dictionary(open(os.path.join(d, "dict.bin"), "rb").read())
    assert (dh.info["hist0"], dh.info["hist1"], dh.info["hist2"]) == (21, 7, 96)
    files = sorted(glob.glob(os.path.join(d, "hist_*.zst")))
    assert len(files) >= 2
    for zf in files:
        z, orig = open(zf, "rb").read(), open(zf[:-4] + ".orig", "rb").read()
        st, out = oracle.decode_frame_with_dict(z, dh, cap=len(orig) + 64)
        assert st == 0 and out == orig, zf
        st0, out0 = oracle.decode_frame_with_dict(z, d0, cap=len(orig) + 64)
        assert st0 != 0 or out0 != orig, zf
