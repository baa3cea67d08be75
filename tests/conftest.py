import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
REFERENCE_CORPUS = "/root/reference/data/decode_corpus"  # only present in the build container


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def corpus_pairs(max_orig: int | None = None):
    """(name, zst_bytes, original_bytes) for every committed golden corpus pair."""
    d = os.path.join(GOLDEN, "decode_corpus")
    out = []
    for n in sorted(os.listdir(d)):
        if n.endswith(".zst"):
            continue
        orig = open(os.path.join(d, n), "rb").read()
        if max_orig is not None and len(orig) > max_orig:
            continue
        out.append((n, open(os.path.join(d, n + ".zst"), "rb").read(), orig))
    return out


@pytest.fixture(scope="session")
def golden_corpus():
    return corpus_pairs()


def raw_frame_with_checksum(data: bytes, corrupt: bool = False) -> bytes:
    """One zstd frame of Raw blocks with the content-checksum flag set (frame.cairo:401 bit 2)."""
    import oracle
    out = bytearray(b"\x28\xb5\x2f\xfd" + bytes([0x04, 0x38]))          # no FCS, window 2^17
    chunks = [data[i:i + 65536] for i in range(0, len(data), 65536)] or [b""]
    for k, c in enumerate(chunks):
        v = (1 if k == len(chunks) - 1 else 0) | (len(c) << 3)
        out += bytes([v & 255, (v >> 8) & 255, (v >> 16) & 255]) + c
    ck = (oracle.xxh64(data) & 0xFFFFFFFF) ^ (0x10 if corrupt else 0)
    return bytes(out + ck.to_bytes(4, "little"))


def add_checksum(frame: bytes, original: bytes) -> bytes:
    """Sets the content-checksum flag of a frame that has none and appends XXH64(original) low 32."""
    import oracle
    assert not frame[4] & 4
    return frame[:4] + bytes([frame[4] | 4]) + frame[5:] + (oracle.xxh64(original) & 0xFFFFFFFF).to_bytes(4, "little")
