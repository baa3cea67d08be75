#!/usr/bin/env python3
"""Generates the round-5 golden vector under tests/golden/vectors/ (data only; needs the oracle, not libzstd):

  d5_uneven_split_with_sequences.zst / .orig
        the block of d5_uneven_4stream_split.zst (four huff0 streams of 144 / 48 / 48 / 48 symbols: divergence D5) followed by a
        sequences section of ONE sequence with the predefined tables (sequence_section_decoder.cairo:405-647), so that the frame
        has chain records and is listed for both execute kernels — and cz_huf_kernel still hands it back because of the uneven
        split.  Pins "a handed-back frame is put on the fall-back list once" (ADVICE r4).  The sequence is found by search: the
        first (LL, OF, ML) initial states the oracle accepts with a match longer than 3 bytes.
  manifest_r5.json     length, sha256 and XXH64 of what it decodes to (the oracle's answer: libzstd rejects the split).
"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "vectors")


def main():
    z5 = open(os.path.join(OUT, "d5_uneven_4stream_split.zst"), "rb").read()
    lits = open(os.path.join(OUT, "d5_uneven_4stream_split.orig"), "rb").read()
    assert z5[7] & 7 == 5 and z5[-1] == 0, "one last Compressed block that ends with `0 sequences`"   # block_decoder.cairo:237-278
    lit_section = z5[10:-1]
    found = None
    for ll_s in range(64):
        for of_s in range(32):
            for ml_s in range(64):
                # backward bitstream, read order: LL state (6 bits), OF state (5), ML state (6), then the extra bits of the one
                # sequence (OF first; two of them here, the codes searched for have no others): sequence_section_decoder.cairo:223-297
                v = (1 << 19) | (ll_s << 13) | (of_s << 8) | (ml_s << 2) | 1
                content = lit_section + bytes([1, 0]) + v.to_bytes(3, "little")
                size = len(content)
                fr = z5[:5] + (len(lits) + 64 - 256).to_bytes(2, "little") + bytes([1 | (2 << 1) | ((size & 31) << 3), (size >> 5) & 255, (size >> 13) & 255]) + content
                st, out, info = oracle.decode_frame(fr, cap=1024)
                if st == 0 and len(out) > len(lits) + 3:             # a match of more than the minimum: the output shows where it went
                    ll = next((i for i in range(len(lits)) if out[i] != lits[i]), len(lits))
                    found = (ll_s, of_s, ml_s, ll, len(out) - len(lits), content)
                    break
            if found:
                break
        if found:
            break
    assert found, "no acceptable sequence found"
    ll_s, of_s, ml_s, ll, ml, content = found
    size = len(content)
    st0, out0, _ = oracle.decode_frame(z5[:5] + (len(lits) + 64 - 256).to_bytes(2, "little") + bytes([1 | (2 << 1) | ((size & 31) << 3), (size >> 5) & 255, (size >> 13) & 255]) + content, cap=1024)
    fr = z5[:5] + (len(out0) - 256).to_bytes(2, "little") + bytes([1 | (2 << 1) | ((size & 31) << 3), (size >> 5) & 255, (size >> 13) & 255]) + content
    st, out, info = oracle.decode_frame(fr, cap=1024)
    assert st == 0 and out == out0 and info["content_size"] == len(out), (st, info)
    name = "d5_uneven_split_with_sequences.zst"
    open(os.path.join(OUT, name), "wb").write(fr)
    open(os.path.join(OUT, name[:-4] + ".orig"), "wb").write(out)
    man = {name: {"orig_len": len(out), "orig_sha256": hashlib.sha256(out).hexdigest(), "xxh64": "%016x" % oracle.xxh64(out),
                  "note": f"the D5 block (4 huff0 streams of 144/48/48/48 symbols) + one sequence with the predefined tables (initial states LL {ll_s} OF {of_s} ML {ml_s}: the output first differs from the literals at byte {ll}, match length {ml}); the oracle's answer"}}
    json.dump(man, open(os.path.join(OUT, "manifest_r5.json"), "w"), indent=1, sort_keys=True)
    print(name, len(fr), "bytes ->", len(out), "|", man[name]["note"])


if __name__ == "__main__":
    main()
