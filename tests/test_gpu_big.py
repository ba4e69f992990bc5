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


TALL = GOLD.get("tall_unit_proofs", [])
TALL_SHAPES = [(24, 93, 19), (25, 21, 5), (26, 21, 5), (25, 93, 19)]     # the last one: the top of the reference's default Cpu range, 215 GB on the card


@pytest.mark.parametrize("log_n,n_cols,n_helpers", TALL_SHAPES, ids=lambda v: str(v))
def test_tall_unit_proof_is_accepted_by_the_oracles_verifier_and_hashes_to_its_digest(ctx, log_n, n_cols, n_helpers):
    """Tables above 2^23 rows: the reference's default ranges are Cpu 12..25, Arithmetic 16..23, Memory 17..28
    (/root/reference/common/src/prover_state/circuit.rs:76-86).  A Cpu-shaped 2^24 x 93 (+40 aux) unit and Memory-shaped 2^25 / 2^26 x 21
    (+12 aux) units are proved on one card.  The oracle's PROVER needs ~6 GB of host memory and minutes per 2^20 x 133 rows, so here the
    oracle's VERIFIER (which needs no trace: caps, openings, the constraint identity at zeta, every FRI query) accepts the proof words;
    where the oracle's proof of the same input was computed once (tools/gen_golden_big.py --tall), its committed sha256 must match too."""
    import orc
    cfg = E.standard_fast_config()
    prog = air.cpu_like_air(n_cols, n_helpers)
    seed = 0x7A110000 + 1000 * log_n + n_cols
    d, pub = zl.synth_trace_dev(ctx, n_cols, log_n, seed)
    try:
        proof = zl.prove_tables(ctx, cfg, [(prog, log_n, d, pub)])
    finally:
        ctx.dev_free(d)
        freed = ctx.trim()                       # tens of GB of cached blocks and 2^26-entry NTT tables: the tests after this one create their own contexts
    assert freed > (8 * n_cols << log_n), freed
    assert proof.size == orc.proof_size_words(cfg, [(prog, log_n, None, pub)])
    assert orc.verify(cfg, [(prog, log_n, None, pub)], proof) == 0, "the oracle's verifier rejects the GPU proof of the 2^%d x %d unit" % (log_n, n_cols)
    bad = proof.copy(); bad[proof.size // 2] ^= np.uint64(1)
    assert orc.verify(cfg, [(prog, log_n, None, pub)], bad) != 0
    for ent in TALL:
        if (ent["log_n"], ent["n_cols"], ent["n_helpers"], ent["seed"]) == (log_n, n_cols, n_helpers, seed):
            assert proof.size == ent["words"] and _sha(proof) == ent["sha256"], "GPU proof differs from the oracle's (committed digest)"


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


# ---- the Plonk half at its real sizes (round 4): BASELINE configs[3] end to end
@pytest.mark.parametrize("ent", GOLD.get("plonk_proofs", []), ids=lambda e: "plonk_2p%d" % e["degree_bits"])
def test_recursion_shaped_plonk_proof_hashes_to_the_oracles_digest(ctx, ent):
    """Recursion-shaped circuits (gates.py's gate set, 135 wires, standard_recursion_config) at the sizes of a real transaction's shrink
    chains and root: 2^14, 2^15, 2^16 rows -- 8n-coset transforms of 2^17 .. 2^19 points, the `_split` combine up to 2^16 coefficients
    and past it.  The oracle's proofs were computed once (tools/gen_golden_big.py --plonk 14,15,16); the GPU's words must hash alike."""
    from eth_tx_proof_amd import plonk
    from eth_tx_proof_amd._abi import standard_recursion_config
    rcfg = standard_recursion_config()
    circ, wires = plonk.recursion_circuit(degree_bits=ent["degree_bits"], seed=ent["seed"])
    pub = np.array(ent["public_inputs"], dtype=np.uint64)
    w = plonk.bind_public_inputs(circ, wires, plonk.SyntheticRecursion.public_inputs_hash(ctx, pub))
    st = zl.PlonkState(ctx, rcfg, circ)
    try:
        assert [int(x) for x in st.digest()] == ent["circuit_digest"]
        proof = zl.prove_plonk(st, w, pub)
        assert proof.size == ent["words"]
        assert _sha(proof) == ent["sha256"], "GPU Plonk proof of 2^%d rows differs from the oracle's (committed digest)" % ent["degree_bits"]
        other = zl.Context(0)                      # a worker's context proves with the same prover data
        try:
            assert _sha(zl.prove_plonk(st, w, pub, ctx=other)) == ent["sha256"]
        finally:
            other.close()
    finally:
        st.free()


@pytest.mark.parametrize("overlap", [False, True], ids=["upstream_order", "chains_overlapped"])
@pytest.mark.parametrize("ent", GOLD.get("prove_root", []), ids=lambda e: "cfg4")
def test_whole_prove_root_of_configs3_hashes_to_the_oracles_digests(ent, overlap):
    """TxProof.execute at BASELINE configs[3]'s REAL sizes: seven tables (2^9 .. 2^17 rows), shrink chains [[14,12],[12],[13],[16,12],
    [14,12],[14,12],[13]] and the 2^15 root over plonk.SyntheticRecursion() at size_shift 0 -- the transaction bench.py --workload
    prove_root times.  Every proof of every chain, the AllProof and the root proof hash to the ORACLE's digests
    (orc.prove_root, tools/gen_golden_big.py --root), in upstream's order in time and with the chains overlapped."""
    from eth_tx_proof_amd import ops, plonk
    old = ops.P_STATE
    ops.P_STATE = None
    try:
        rec = plonk.SyntheticRecursion()
        st = ops.set_prover_state_from_config(0, recursion=rec, chain_overlap=overlap)
        tabs = []
        for t, (w, lg) in enumerate(zip(ent["widths"], ent["log_sizes"])):
            d, pub = zl.synth_trace_dev(st.ctx(), w, lg, ent["seed_base"] + 1000 * t)
            tabs.append(ops.TableInput(air.cpu_like_air(w, min(19, w // 4)), lg, d, pub))
        assert [[b for _, _, b in rec.shrink_chain(t, tb.air, tb.log_n)] for t, tb in enumerate(tabs)] == ent["chains"]
        tm = {"collect": []}
        root, _, all_proof = ops.prove_root(ops.GenerationInputs(tabs), tm)
        assert all_proof.size == ent["all_proof"]["words"] and _sha(all_proof) == ent["all_proof"]["sha256"]
        got = {(ti, si): p for ti, si, p in tm["collect"]}
        assert len(got) == len(ent["chain_proofs"])
        for e in ent["chain_proofs"]:
            p = got[(e["table"], e["step"])]
            assert p.size == e["words"] and _sha(p) == e["sha256"], "chain proof of table %d step %d (2^%d rows) differs from the oracle's" % (e["table"], e["step"], e["degree_bits"])
        assert root.size == ent["root"]["words"] and _sha(root) == ent["root"]["sha256"], "root proof differs from the oracle's"
        # TxProof.execute is this call
        assert _sha(ops.TxProof().execute(ops.GenerationInputs(tabs)).proof) == ent["root"]["sha256"]
    finally:
        if ops.P_STATE is not None:
            ops.P_STATE.drop_ctx()
        ops.P_STATE = old


# ---- BASELINE configs[4] at full size on one card: a 16-transaction block, map -> fold -> block (round 4)
_BLOCK_GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "proof_digests_block.json")


@pytest.mark.skipif(not os.path.exists(_BLOCK_GOLD), reason="tests/golden/proof_digests_block.json not generated (tools/gen_golden_block.py)")
def test_sixteen_transaction_block_hashes_to_the_oracles_digests():
    """The caller of the operators at full size (/root/reference/leader/src/prover.rs:20-43, BASELINE configs[4] on one GPU): 16 transactions --
    each the seven tables of configs[3], shrink chains and root proof -- mapped over four workers, the leader's balanced pairwise fold
    (15 aggregation proofs of 2^14 rows), the block proof.  Every root proof, every proof of every fold level and the block proof hash to the
    ORACLE's digests (tools/gen_golden_block.py: four distinct transactions, each used four times)."""
    from eth_tx_proof_amd import leader, ops, plonk
    gold = json.load(open(_BLOCK_GOLD))
    old = ops.P_STATE
    ops.P_STATE = None
    try:
        rec = plonk.SyntheticRecursion()
        st = ops.set_prover_state_from_config(0, recursion=rec)
        state_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "eth-tx-proof_amd", "prover_state")
        c0 = st.ctx()
        units = []
        for u in range(gold["units"]):
            tabs = []
            for t, (w, lg) in enumerate(zip(gold["widths"], gold["log_sizes"])):
                d, pub = zl.synth_trace_dev(c0, w, lg, gold["block_seed"] + 100000 * (u % gold["distinct"]) + 1000 * t)
                tabs.append(ops.TableInput(air.cpu_like_air(w, min(19, w // 4)), lg, d, pub))
            units.append(ops.GenerationInputs(tabs, txn_number_before=u, block_number=1))
        c0.synchronize()
        if os.path.isdir(state_dir):
            zl.prover_state_cache_dir(state_dir)
            try:
                st.prepare(units[0].tables)
            finally:
                zl.prover_state_cache_dir(None)
        rt = leader.Runtime(num_workers=4)
        txs = rt.map(ops.TxProof(), units)
        for u, tx in enumerate(txs):
            g = gold["roots"][u % gold["distinct"]]
            assert tx.proof.size == g["words"] and _sha(tx.proof) == g["sha256"], "root proof of transaction %d differs from the oracle's" % u
        made = []

        class Collect(ops.AggProof):
            def combine(self, a, b):
                r = super().combine(a, b)
                made.append(r.proof)
                return r
        agg = leader.Runtime.fold(Collect(), txs)
        want = [e for lvl in gold["fold_levels"] for e in lvl]
        assert len(made) == len(want) == gold["units"] - 1
        for k, (p, g) in enumerate(zip(made, want)):
            assert p.size == g["words"] and _sha(p) == g["sha256"], "aggregation proof %d differs from the oracle's" % k
        block = ops.BlockProof().execute(agg)
        assert block.b_height == 1 and block.intern.size == gold["block"]["words"] and _sha(block.intern) == gold["block"]["sha256"]
        # the leader's own entry point does the same three stages
        again = leader.ProverInput(units).prove(leader.Runtime(num_workers=4))
        assert _sha(again.intern) == gold["block"]["sha256"]
    finally:
        if ops.P_STATE is not None:
            ops.P_STATE.drop_ctx()
        ops.P_STATE = old
