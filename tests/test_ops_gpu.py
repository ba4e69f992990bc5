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
    st = ops.set_prover_state_from_config(0)
    assert ops.set_prover_state_from_config(0) is st          # second init keeps the first state
    prog = air.cpu_like_air(9, 2)
    tr, pub = orc.synth_trace(9, 7, 5)
    out = ops.TxProof().execute(ops.GenerationInputs([ops.TableInput(prog, 7, tr, pub)], txn_number_before=3, block_number=9))
    assert out.kind == "Txn"
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


def test_prover_input_map_fold_block(orc):
    """The caller of the operators, /root/reference/leader/src/prover.rs:20-43: map(&TxProof) over five units on three worker
    threads, fold(&AggProof) pairwise, BlockProof on the result -- every proof on the GPU (STARK prover for the units, Plonk prover
    for the fold and the block).  The recursion circuits are the synthetic stand-ins of plonk.SyntheticRecursion (the real ones are
    an input), so what is checked is the flow and the proofs: each stage equals the oracle's proof of the same
    statement bit for bit and the oracle's verifiers accept."""
    from eth_tx_proof_amd import air, leader, ops, plonk
    old = ops.P_STATE
    ops.P_STATE = None
    try:
        rec = plonk.SyntheticRecursion(degree_bits=7, seed=3)
        st = ops.set_prover_state_from_config(0, recursion=rec)
        prog = air.cpu_like_air(9, 2)
        units, traces = [], []
        for u in range(5):
            tr, pub = orc.synth_trace(9, 7, 100 + u)
            traces.append((tr, pub))
            units.append(ops.GenerationInputs([ops.TableInput(prog, 7, tr, pub)], txn_number_before=u, block_number=1))
        rt = leader.Runtime(num_workers=3)
        block = leader.ProverInput(units).prove(rt)
        assert rt.requeued == []
        # ---- the same job through the oracle
        proofs = []
        for tr, pub in traces:
            rc, p = orc.prove(st.config, [(prog, 7, tr, pub)])
            assert rc == 0
            proofs.append(p)
        dig = lambda w: orc.hash_rows(np.ascontiguousarray(w)[None, :])[0]
        agg_state = orc.PlonkState(rec.cfg, rec.circuits["agg"][0])
        blk_state = orc.PlonkState(rec.cfg, rec.circuits["block"][0])

        def combine(a, b):
            pubs = np.concatenate([dig(a), dig(b)])
            rc, p = orc.prove_plonk(agg_state, plonk.bind_public_inputs(*rec.circuits["agg"], dig(pubs)), pubs)
            assert rc == 0 and orc.verify_plonk(agg_state, pubs, p) == 0
            return p
        level = proofs
        while len(level) > 1:       # the same balanced tree as Runtime.fold
            nxt = [combine(level[j], level[j + 1]) for j in range(0, len(level) - 1, 2)]
            if len(level) & 1:
                nxt.append(level[-1])
            level = nxt
        pubs = np.concatenate([np.array([1], dtype=np.uint64), dig(level[0]), np.zeros(4, dtype=np.uint64)])
        rc, ref = orc.prove_plonk(blk_state, plonk.bind_public_inputs(*rec.circuits["block"], dig(pubs)), pubs)
        assert rc == 0
        assert block.b_height == 1 and (block.intern == ref).all()
        assert orc.verify_plonk(blk_state, pubs, block.intern) == 0
        # one unit folds to a Txn proof, which the reference refuses (prover.rs:41)
        with pytest.raises(RuntimeError, match="not GeneratedAggProof"):
            leader.ProverInput(units[:1]).prove(rt)
    finally:
        ops.P_STATE = old


def test_failed_unit_is_requeued_once(orc, monkeypatch):
    """A unit that fails with a device error goes back on the queue once ('here' column); a second failure, or
    any other error, is fatal as in the reference (every prover error -> FatalError, /root/reference/ops/src/lib.rs:52)."""
    from eth_tx_proof_amd import air, leader, lib as zl, ops
    old = ops.P_STATE
    ops.P_STATE = None
    try:
        ops.set_prover_state_from_config(0)
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
