"""Parity at the sizes the oracle cannot be re-run at inside the suite: BASELINE configs[2] (2^20 rows x 93 + 40 columns, bench.py's
seeds), north_star's 2^23-row unit (and 2^22), the seven-table unit of configs[3].  The ORACLE's proofs of exactly these inputs were
computed once (tools/gen_golden_big.py on a many-core box: 32 s / 132 s / 258 s of oracle time) and their sha256 + length are committed in
tests/golden/proof_digests_big.json; here the GPU proves the same inputs and its proof words must hash to the same digest -- bit-exact
parity in seconds, no oracle on the GPU box.  (The generator run itself compared GPU and oracle word for word: "gpu_identical".)"""
import hashlib
import json
import os

import numpy as np
import pytest

import eth_tx_proof_amd as E
from eth_tx_proof_amd import air, lib as zl

pytestmark = pytest.mark.gpu
GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "proof_digests_big.json")))


def _sha(p):
    return hashlib.sha256(np.ascontiguousarray(p, dtype=np.uint64).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def ctx():
    c = zl.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("ent", GOLD["unit_proofs"], ids=lambda e: "2p%d_seed%x" % (e["log_n"], e["seed"]))
def test_big_unit_proof_hashes_to_the_oracles_digest(ctx, ent):
    cfg = E.standard_fast_config()
    prog = air.cpu_like_air(ent["n_cols"], ent["n_helpers"])
    d, pub = zl.synth_trace_dev(ctx, ent["n_cols"], ent["log_n"], ent["seed"])
    try:
        proof = zl.prove_tables(ctx, cfg, [(prog, ent["log_n"], d, pub)])
    finally:
        ctx.dev_free(d)
    assert proof.size == ent["words"]
    assert _sha(proof) == ent["sha256"], "GPU proof of the 2^%d-row unit differs from the oracle's (committed digest)" % ent["log_n"]


@pytest.mark.parametrize("ent", GOLD["cfg4_units"], ids=lambda e: "cfg4")
def test_seven_table_unit_hashes_to_the_oracles_digest(ctx, ent):
    cfg = E.standard_fast_config()
    items, ds = [], []
    try:
        for t, (w, lg) in enumerate(zip(ent["widths"], ent["log_sizes"])):
            prog = air.cpu_like_air(w, min(19, w // 4))
            d, pub = zl.synth_trace_dev(ctx, w, lg, ent["seed_base"] + 1000 * t)
            ds.append(d)
            items.append((prog, lg, d, pub))
        proof = zl.prove_tables(ctx, cfg, items)
    finally:
        for d in ds:
            ctx.dev_free(d)
    assert proof.size == ent["words"] and _sha(proof) == ent["sha256"]
