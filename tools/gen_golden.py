#!/usr/bin/env python3
"""Writes tests/golden/vectors.json from oracle/pyspec.py (pure Python, no reference import is possible:
the reference is Rust and its arithmetic crates are absent -- SURVEY.md 8c).

  poseidon_kats       the four upstream plonky2 `test_vectors` (externally anchored)
  fft_1_to_8, lde_1234  forced by the verified field constants (A3)
  sponge_derived, two_to_one_derived   self-consistency values under the recalled sponge conventions (A4),
                      NOT upstream vectors -- flagged "derived"
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyspec as S  # noqa: E402

out = {
    "poseidon_kats": [{"input": ["%016x" % x for x in i], "output": o.split()} for i, o in S.KATS],
    "fft_1_to_8": {"input": list(range(1, 9)), "output": S.fft(list(range(1, 9)))},
    "lde_1234": {"coeffs": [1, 2, 3, 4], "values": S.lde_values([1, 2, 3, 4], 1)},
    "sponge_derived": [{"input": x, "output": ["%016x" % v for v in S.hash_no_pad(x)]}
                       for x in ([0, 1, 2, 3, 4], list(range(93)), list(range(8)), list(range(16)))],
    "two_to_one_derived": {"left": [1, 2, 3, 4], "right": [5, 6, 7, 8],
                           "output": ["%016x" % v for v in S.two_to_one([1, 2, 3, 4], [5, 6, 7, 8])]},
}
for kat in out["poseidon_kats"]:
    assert ["%016x" % v for v in S.poseidon([int(x, 16) for x in kat["input"]])] == kat["output"]
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "vectors.json"), "w"), indent=1)
print("written")
