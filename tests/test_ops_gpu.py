"""The operator surface reads like the reference's: set the prover state once, TxProof.execute(input) -> proof."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_txproof_execute(orc):
    import eth_tx_proof_amd as E
    from eth_tx_proof_amd import air, ops
    with pytest.raises(RuntimeError, match="Prover state is not initialized"):
        if ops.P_STATE is None:
            ops.p_state()
        else:
            raise RuntimeError("Prover state is not initialized")
    st = ops.set_prover_state_from_config(0, stark_only=True)  # the STARK half alone, on explicit request
    assert ops.set_prover_state_from_config(0) is st          # second init keeps the first state
    prog = air.cpu_like_air(9, 2)
    tr, pub = orc.synth_trace(9, 7, 5)
    out = ops.TxProof().execute(ops.GenerationInputs([ops.TableInput(prog, 7, tr, pub)], txn_number_before=3, block_number=9))
    assert out.kind == "Txn"
    old = ops.P_STATE
    ops.P_STATE = ops.ProverState(0)                          # neither recursion circuits nor stark_only: prove_root cannot run
    try:
        with pytest.raises(ops.FatalError, match="no recursion circuits"):
            ops.TxProof().execute(ops.GenerationInputs([ops.TableInput(prog, 7, tr, pub)]))
    finally:
        ops.P_STATE = old
    rc, ref = orc.prove(st.config, [(prog, 7, tr, pub)])
    assert rc == 0 and (out.proof == ref).all()
    assert orc.verify(st.config, [(prog, 7, None, pub)], out.proof) == 0
    with pytest.raises(ops.FatalError):
        ops.AggProof().combine(out, out)
    with pytest.raises(NotImplementedError):
        ops.AggProof().empty()
    bad = ops.GenerationInputs([ops.TableInput(prog, 7, tr[:, :64], pub)])
    with pytest.raises((ops.FatalError, AssertionError)):
        ops.TxProof().execute(bad)


def test_cpp_operator_mirror_runs(zk, tmp_path):
    import subprocess
    from test_abi_cpu import _build_ops_smoke
    rc = subprocess.run([_build_ops_smoke(tmp_path)], capture_output=True, text=True)
    assert rc.returncode == 0, rc.stdout + rc.stderr
    assert "proof words" in rc.stdout


def _small_tables(orc, seed0, shapes=((165, 7), (21, 8), (9, 6))):
    from eth_tx_proof_amd import air, ops
    tabs, items = [], []
    for t, (w, lg) in enumerate(shapes):
        prog = air.cpu_like_air(w, min(19, w // 4))
        tr, pub = orc.synth_trace(w, lg, seed0 + t)
        tabs.append(ops.TableInput(prog, lg, tr, pub))
        items.append((prog, lg, tr, pub))
    return tabs, items


def test_txproof_is_the_whole_prove_root(orc):
    """TxProof.execute = generate_txn_proof = AllRecursiveCircuits::prove_root (/root/reference/ops/src/lib.rs:52): the STARK proofs of
    all tables, per table the initial wrapper and the shrinking wrappers (until degree_bits <= threshold), then the root proof -- every
    proof on the GPU, and the root proof, the shrunk proofs' chain and the STARK proofs equal the oracle's restatement of the same flow
    (orc.prove_root) word for word.  Circuits: plonk.SyntheticRecursion stand-ins (recursion-shaped gate set, reduced sizes)."""
    import eth_tx_proof_amd as E
    from eth_tx_proof_amd import ops, plonk
    old = ops.P_STATE
    ops.P_STATE = None
    try:
        rec = plonk.SyntheticRecursion(size_shift=-6, threshold=6, seed=2)
        st = ops.set_prover_state_from_config(0, recursion=rec)
        tabs, items = _small_tables(orc, 40)
        chains = [[b for _, _, b in rec.shrink_chain(t, it[0], it[1])] for t, it in enumerate(items)]
        assert chains == [[7, 6], [6], [6]]                   # a chain with a shrinking wrapper and two without
        timing = {}
        root, pv, all_proof = ops.prove_root(ops.GenerationInputs(tabs), timing)
        ref_root, ref_all, ref_shrunk = orc.prove_root(st.config, rec.cfg, items, rec)
        assert (all_proof == ref_all).all()
        assert root.size == ref_root.size and (root == ref_root).all()
        assert set(timing) == {"stark_s", "shrink_s", "root_s"}
        out = ops.TxProof().execute(ops.GenerationInputs(tabs))
        assert out.kind == "Txn" and (out.proof == ref_root).all() and len(out.public_values) == 3
        st.chain_streams = 3                                  # the three (independent) shrink chains side by side: the same proofs
        root3, _, all3 = ops.prove_root(ops.GenerationInputs(tabs))
        assert (root3 == ref_root).all() and (all3 == ref_all).all()
        # ... that was with every chain started under the later tables' STARK proofs (the rule for <= 2 calls in flight); upstream's order
        # in time -- prove(), then the chains side by side -- gives the same words
        st.chain_overlap = False
        root3b, _, all3b = ops.prove_root(ops.GenerationInputs(tabs))
        assert (root3b == ref_root).all() and (all3b == ref_all).all()
        st.chain_overlap = "auto"
        # zk_prove_tables_notify itself: table t's words arrive in order, equal to the slices of the finished AllProof; a listener's
        # exception surfaces after the call
        from eth_tx_proof_amd import lib as zl
        got = []
        pr = zl.prove_tables(st.ctx(), st.config, items, on_table=lambda t, w: got.append((t, w)))
        assert [t for t, _ in got] == [0, 1, 2] and (pr == ref_all).all()
        assert all((w == ref).all() for (_, w), ref in zip(got, ops.split_all_proof(st.config, tabs, pr)))
        with pytest.raises(ZeroDivisionError):
            zl.prove_tables(st.ctx(), st.config, items, on_table=lambda t, w: 1 // 0)
        st.chain_streams = 1
        # a size no circuits were built for is fatal, as upstream's "Missing preprocessed circuits"
        st.recursion = plonk.SyntheticRecursion(size_shift=-6, threshold=6, seed=2, degree_bits_ranges=[(8, 9)] * 3)
        with pytest.raises(ops.FatalError, match="Missing preprocessed circuits for table 0 with size 7"):
            ops.TxProof().execute(ops.GenerationInputs(tabs))
    finally:
        ops.P_STATE = old


def test_prover_input_map_fold_block(orc):
    """The caller of the operators, /root/reference/leader/src/prover.rs:20-43: map(&TxProof) over five units on three worker
    threads (each unit = prove_root: STARK proofs, shrink chains, root proof), fold(&AggProof) pairwise, BlockProof on the result --
    every proof on the GPU.  The recursion circuits are the stand-ins of plonk.SyntheticRecursion (the real ones are an input,
    SURVEY.md A10), so what is checked is the flow and the proofs: the block proof equals the oracle's for the whole job bit for bit
    and the oracle's verifier accepts."""
    from eth_tx_proof_amd import leader, ops, plonk
    old = ops.P_STATE
    ops.P_STATE = None
    try:
        rec = plonk.SyntheticRecursion(size_shift=-7, threshold=5, seed=3)
        st = ops.set_prover_state_from_config(0, recursion=rec)
        units, unit_items = [], []
        for u in range(5):
            tabs, items = _small_tables(orc, 100 + 10 * u, shapes=((9, 7), (12, 6)))
            units.append(ops.GenerationInputs(tabs, txn_number_before=u, block_number=1))
            unit_items.append(items)
        rt = leader.Runtime(num_workers=3)
        block = leader.ProverInput(units).prove(rt)
        assert rt.requeued == []
        ref, pubs, _ = orc.prove_block(st.config, rec.cfg, unit_items, rec)
        assert block.b_height == 1 and block.intern.size == ref.size and (block.intern == ref).all()
        assert orc.verify_plonk(orc._plonk_state_for(rec.cfg, rec.circuit("block")[0]), pubs, block.intern) == 0
        # one unit folds to a Txn proof, which the reference refuses (prover.rs:41)
        with pytest.raises(RuntimeError, match="not GeneratedAggProof"):
            leader.ProverInput(units[:1]).prove(rt)
    finally:
        ops.P_STATE = old


def test_wire_formats_on_gpu_made_proofs(orc):
    """SURVEY.md 8f-3 on proofs the GPU made: the AllProof (StarkProof per table), a Plonk proof of the chain, the root proof as
    GeneratedTxnProof and the block proof as the leader prints it (/root/reference/leader/src/main.rs:60) -> serde-shaped JSON -> words:
    identical words back, the oracle's verifiers accept what was re-read, and reject it after an edit of the JSON."""
    import json
    from eth_tx_proof_amd import leader, ops, plonk, wire
    old = ops.P_STATE
    ops.P_STATE = None
    try:
        rec = plonk.SyntheticRecursion(size_shift=-7, threshold=5, seed=6)
        st = ops.set_prover_state_from_config(0, recursion=rec)
        units, unit_items = [], []
        for u in range(2):
            tabs, items = _small_tables(orc, 500 + 10 * u, shapes=((9, 7), (12, 6)))
            units.append(ops.GenerationInputs(tabs, txn_number_before=u, block_number=1))
            unit_items.append(items)
        root, pv, all_proof = ops.prove_root(units[0])
        # ---- StarkProofs
        back = wire.from_json(wire.to_json(all_proof, st.config), st.config)
        assert (back == all_proof).all()
        vitems = [(p, lg, None, pub) for p, lg, _, pub in unit_items[0]]
        assert orc.verify(st.config, vitems, back) == 0
        d = wire.to_dict(all_proof, st.config)
        d["tables"][1]["proof"]["opening_proof"]["final_poly"]["coeffs"][0][0] ^= 1
        assert orc.verify(st.config, vitems, wire.from_dict(d, st.config)) != 0
        # ---- the root proof as GeneratedTxnProof { p_vals, intern }
        ref_root, _, shrunk = orc.prove_root(st.config, rec.cfg, unit_items[0], rec)
        root_pubs = np.concatenate([np.array([7, 6], dtype=np.uint64)] + [orc.words_digest(sp) for sp in shrunk])
        txn = ops.AggregatableProof("Txn", root, pv)
        j = json.loads(wire.txn_proof_to_json(txn, rec.cfg, root_pubs))
        assert len(j["p_vals"]) == 2 and j["intern"]["public_inputs"] == [int(x) for x in root_pubs]
        words, pis = wire.plonk_from_dict({"common": wire.common_of(j["intern"], rec.cfg), "proof_with_pis": j["intern"]})
        root_state = orc._plonk_state_for(rec.cfg, rec.circuit("root")[0])
        assert (words == root).all() and orc.verify_plonk(root_state, pis, words) == 0
        j["intern"]["proof"]["openings"]["plonk_zs"][0][0] ^= 1
        assert orc.verify_plonk(root_state, pis, wire.plonk_from_dict({"common": wire.common_of(j["intern"], rec.cfg), "proof_with_pis": j["intern"]})[0]) != 0
        # ---- the block proof
        block = leader.ProverInput(units).prove(leader.Runtime(num_workers=2))
        ref, blk_pubs, _ = orc.prove_block(st.config, rec.cfg, unit_items, rec)
        assert (block.intern == ref).all()
        text = wire.block_proof_to_json(block, rec.cfg, blk_pubs)
        h, words, pis = wire.block_proof_from_json(text)
        blk_state = orc._plonk_state_for(rec.cfg, rec.circuit("block")[0])
        assert h == 1 and (words == block.intern).all() and orc.verify_plonk(blk_state, pis, words) == 0
        jb = json.loads(text)
        jb["intern"]["public_inputs"][1] ^= 1
        _, w2, p2 = wire.block_proof_from_json(json.dumps(jb))
        assert orc.verify_plonk(blk_state, p2, w2) != 0
    finally:
        ops.P_STATE = old


def test_failed_unit_is_requeued_once(orc, monkeypatch):
    """A unit that fails with a device error goes back on the queue once ('here' column); a second failure, or
    any other error, is fatal as in the reference (every prover error -> FatalError, /root/reference/ops/src/lib.rs:52)."""
    from eth_tx_proof_amd import air, leader, lib as zl, ops
    old = ops.P_STATE
    ops.P_STATE = None
    try:
        ops.set_prover_state_from_config(0, stark_only=True)
        prog = air.cpu_like_air(9, 2)
        units = []
        for u in range(4):
            tr, pub = orc.synth_trace(9, 6, 200 + u)
            units.append(ops.GenerationInputs([ops.TableInput(prog, 6, tr, pub)], txn_number_before=u))
        want = [ops.TxProof().execute(u).proof for u in units]
        real = zl.prove_tables
        fails = {"left": 1}

        def flaky(ctx, cfg, items, out=None):
            if fails["left"] > 0 and items[0][3][0] == units[2].tables[0].public_inputs[0]:
                fails["left"] -= 1
                raise zl.ZkError(-3, "zk_prove_tables", "injected device error")
            return real(ctx, cfg, items, out=out)
        monkeypatch.setattr(zl, "prove_tables", flaky)
        rt = leader.Runtime(num_workers=2)
        got = rt.map(ops.TxProof(), units)
        assert [g.kind for g in got] == ["Txn"] * 4 and all((g.proof == w).all() for g, w in zip(got, want))
        assert len(rt.requeued) == 1 and rt.requeued[0][0] == 2
        fails["left"] = 2                                       # fails again after the re-queue: fatal
        with pytest.raises(ops.FatalError, match="injected device error"):
            leader.Runtime(num_workers=2).map(ops.TxProof(), units)

        def invalid(ctx, cfg, items, out=None):
            raise zl.ZkError(-1, "zk_prove_tables", "bad argument")
        monkeypatch.setattr(zl, "prove_tables", invalid)
        rt = leader.Runtime(num_workers=2)
        with pytest.raises(ops.FatalError, match="bad argument"):
            rt.map(ops.TxProof(), units)
        assert rt.requeued == []
    finally:
        ops.P_STATE = old
