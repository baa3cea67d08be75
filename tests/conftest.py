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
    config.addinivalue_line("markers", "xdist_group(name): tests of one group stay on one worker (pytest-xdist)")
    _spread_cpu_tests(config)


def _spread_cpu_tests(config):
    """On a box WITHOUT a GPU the suite is dominated by the emulator tests (tests/emu: the kernel source with lanes as threads under
    ASan — each run is one mostly serial program), so the tests are spread over a few worker processes when pytest-xdist is there
    and the command line did not ask for anything itself.  Never on a GPU box: GPU tests run in ONE process.  CZ_TESTS_SERIAL=1 turns
    it off."""
    if os.path.exists("/dev/kfd") or hasattr(config, "workerinput") or os.environ.get("CZ_TESTS_SERIAL") == "1":
        return
    if not config.pluginmanager.hasplugin("xdist") or getattr(config.option, "numprocesses", None) or getattr(config.option, "dist", "no") != "no":
        return
    if getattr(config.option, "collectonly", False) or getattr(config.option, "usepdb", False):
        return
    workers = min(4, os.cpu_count() or 1)
    if workers < 2:
        return
    import subprocess
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "emu"), "emu_decode"], stdout=subprocess.DEVNULL)   # once, before the workers ask for it
    config.option.numprocesses = workers
    config.option.dist = "loadgroup"
    config.option.tx = ["popen"] * workers


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
