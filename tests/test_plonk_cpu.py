"""Oracle side of the Plonk-shaped proof (oracle/plonk.c) on the CPU: the prover's output is accepted by the verifier written
beside it (vanishing-polynomial identity at zeta, FRI over the four oracles, every Merkle path), and every kind of wrong input --
a flipped proof word, a witness that breaks a gate where its selector is on, a witness that breaks a copy constraint, other
public inputs -- is rejected.  PARITY UNPINNED: no upstream vector exists for this path."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def setup(orc):
    import eth_tx_proof_amd as E
    from eth_tx_proof_amd import plonk
    cfg = E.standard_recursion_config()
    circ, wires = plonk.synthetic_circuit(degree_bits=6, seed=11)
    st = orc.PlonkState(cfg, circ)
    return cfg, circ, wires, st


def test_oracle_plonk_proof_verifies_and_is_deterministic(orc, setup):
    cfg, circ, wires, st = setup
    pub = np.array([7, 8, 9], dtype=np.uint64)
    rc, proof = orc.prove_plonk(st, wires, pub)
    assert rc == 0 and proof.size == orc.plonk_proof_size_words(cfg, circ)
    assert orc.verify_plonk(st, pub, proof) == 0
    rc2, proof2 = orc.prove_plonk(st, wires, pub)
    assert rc2 == 0 and (proof == proof2).all()
    assert (proof[5:] < np.uint64(0xFFFFFFFF00000001)).all()


def test_oracle_plonk_rejects_wrong_inputs(orc, setup):
    cfg, circ, wires, st = setup
    pub = np.array([7, 8, 9], dtype=np.uint64)
    rc, proof = orc.prove_plonk(st, wires, pub)
    rng = np.random.default_rng(5)
    for pos in rng.integers(5, proof.size, size=12):
        bad = proof.copy(); bad[pos] ^= np.uint64(1)
        assert orc.verify_plonk(st, pub, bad) != 0, pos
    assert orc.verify_plonk(st, np.array([7, 8, 10], dtype=np.uint64), proof) != 0
    assert orc.verify_plonk(st, pub, proof[:-1]) != 0
    on = int(np.nonzero(circ.constants[2] == 1)[0][2]); off = int(np.nonzero(circ.constants[2] == 0)[0][2])
    for row, expect_ok in ((on, False), (off, True)):       # a broken gate counts only where its selector is 1
        w = wires.copy(); w[90, row] ^= np.uint64(1)
        rc, p = orc.prove_plonk(st, w, pub)
        assert rc == 0 and (orc.verify_plonk(st, pub, p) == 0) == expect_ok
    w = wires.copy(); w[8, 5] ^= np.uint64(1)                 # routed wire 8 is tied to wire 9: copy constraint broken
    rc, p = orc.prove_plonk(st, w, pub)
    assert rc == 0 and orc.verify_plonk(st, pub, p) != 0


def test_oracle_plonk_refuses_malformed_circuits(orc):
    import eth_tx_proof_amd as E
    from eth_tx_proof_amd import plonk
    cfg = E.standard_recursion_config()
    circ, _ = plonk.synthetic_circuit(degree_bits=5, seed=1)
    circ.c.quotient_degree_factor = 4                          # must equal 2^rate_bits
    with pytest.raises(ValueError):
        orc.PlonkState(cfg, circ)
    assert orc.plonk_proof_size_words(cfg, circ) == 0
