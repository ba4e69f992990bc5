"""Opt-in parity run beyond the default suite: north_star's 2^23-row unit proof (and 2^22) against the oracle's prover, word for word.
The oracle needs minutes of CPU time and about 6 GB of host memory per 2^20 rows, so the test only runs with ZK_TEST_BIG=1
(`ZK_TEST_BIG=1 python -m pytest tests/test_gpu_big.py -m gpu -s`); its last result is profiles/r02z_parity_big_2p22_2p23.jsonl."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.skipif(os.environ.get("ZK_TEST_BIG") != "1", reason="minutes of oracle time: set ZK_TEST_BIG=1")
@pytest.mark.parametrize("log_n", [22, 23])
def test_big_unit_proof_identical_to_the_oracles(log_n):
    here = os.path.dirname(os.path.abspath(__file__))
    out = subprocess.run([sys.executable, os.path.join(here, "parity_big.py"), str(log_n)], capture_output=True, text=True, timeout=3000)
    res = json.loads(out.stdout.strip().splitlines()[-1])
    if "skipped" in res:
        pytest.skip(res["skipped"])
    assert out.returncode == 0 and res["identical_word_for_word"], res
