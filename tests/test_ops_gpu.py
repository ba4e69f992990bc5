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
