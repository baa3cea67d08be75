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
