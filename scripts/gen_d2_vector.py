#!/usr/bin/env python3
"""Generates tests/golden/vectors/d2_weight_log10.zst: a frame that is VALID for the reference and for the
oracle but whose Huffman-weight FSE description has accuracy log 10 (the zstd format allows 6; the reference
passes max_log 100, src/huff0/huff0_decoder.cairo:176).  The device caps that log at 9 and answers
CZ_E_UNSUPPORTED (DESIGN.md divergence D2); tests/test_oracle_kats.py and tests/test_gpu_parity.py pin both sides.

Construction (checked against the oracle below): weights = fifteen 1s, coded with an FSE table of log 10 in which
symbol 0 has probability 1 and symbol 1 probability 1023 — every state >= 2 of that table decodes symbol 1 with 0
bits and goes to state - 2, state 1 decodes symbol 1 and reads 1 bit — so two start states (100 and 13) and no
further bits give exactly 15 weights (src/huff0/huff0_decoder.cairo:227-274).  16 symbols of 4 bits follow."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402


class FwdBits:                                       # LSB-first writer (bit_reader.cairo)
    def __init__(self):
        self.v, self.n = 0, 0

    def put(self, val, bits):
        self.v |= val << self.n
        self.n += bits

    def bytes(self):
        return self.v.to_bytes((self.n + 7) // 8, "little")


def reversed_stream(fields):
    """fields = [(value, bits), ...] in READ order; returns the bytes of a backward bitstream with its padding marker."""
    v = 1
    for val, bits in fields:
        v = (v << bits) | val
    return v.to_bytes((v.bit_length() + 7) // 8, "little")


def main():
    # FSE description: accuracy log 10, probabilities (1, 1023)   (fse_decoder.cairo:258-368)
    fb = FwdBits()
    fb.put(10 - 5, 4)
    fb.put(2, 10)            # symbol 0: prob 1 -> value 2 < low 1022: short form, 10 bits
    fb.put(1024 + 1023, 11)  # symbol 1: prob 1023 -> value 1024 > mask 1023: value + low(1023), 11 bits
    fse = fb.bytes()
    weights = reversed_stream([(100, 10), (13, 10)])
    tree = bytes([len(fse) + len(weights)]) + fse + weights
    lits = bytes([3, 0, 15, 7, 8, 1, 12, 5, 5, 9])
    stream = reversed_stream([(b, 4) for b in lits])
    comp = len(tree) + len(stream)
    regen = len(lits)
    lsh = bytes([2 | (0 << 2) | ((regen & 0xF) << 4), ((regen >> 4) & 0x3F) | ((comp & 3) << 6), comp >> 2])
    block = lsh + tree + stream + b"\x00"
    bh = 1 | (2 << 1) | (len(block) << 3)
    frame = bytes.fromhex("28b52ffd") + bytes([0x20, regen]) + bh.to_bytes(3, "little") + block
    st, out, info = oracle.decode_frame(frame, cap=64)
    assert st == 0 and out == lits, (st, out)
    path = os.path.join(ROOT, "tests", "golden", "vectors", "d2_weight_log10.zst")
    open(path, "wb").write(frame)
    open(path[:-4], "wb").write(lits)
    print("wrote", path, len(frame), "bytes; oracle decodes it to", lits.hex())


if __name__ == "__main__":
    main()
