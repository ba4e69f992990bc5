"""zk_prove_plonk on the GPU against oracle/plonk.c: same circuit, same witness, same public inputs -> the same proof words, bit
for bit, and the oracle's verifier accepts them (the recursion half of generate_txn_proof / generate_agg_proof /
generate_block_proof, /root/reference/ops/src/lib.rs:52,72,95, on a synthetic circuit with the recursion config's shape)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cfg():
    import eth_tx_proof_amd as E
    return E.standard_recursion_config()


@pytest.mark.parametrize("degree_bits,seed", [(6, 1), (9, 2), (12, 3), (13, 4), (14, 5)])
def test_plonk_proof_bit_exact_vs_oracle(ctx, zk, orc, degree_bits, seed):
    from eth_tx_proof_amd import plonk
    cfg = _cfg()
    circ, wires = plonk.synthetic_circuit(degree_bits=degree_bits, seed=seed)
    pub = np.array([degree_bits, seed, 99], dtype=np.uint64)
    ost = orc.PlonkState(cfg, circ)
    st = zk.PlonkState(ctx, cfg, circ)
    try:
        assert (st.digest() == ost.digest()).all()
        proof = zk.prove_plonk(st, wires, pub)
        rc, ref = orc.prove_plonk(ost, wires, pub)
        assert rc == 0 and proof.size == ref.size == zk.plonk_proof_size_words(cfg, circ)
        if not (proof == ref).all():
            bad = np.nonzero(proof != ref)[0]
            pytest.fail("first differing word %d of %d (%d differ)" % (bad[0], proof.size, bad.size))
        assert orc.verify_plonk(ost, pub, proof) == 0
        assert (zk.prove_plonk(st, wires, pub) == proof).all()
    finally:
        st.free()


def test_plonk_2_14_rows_recursion_shape_verifies_and_rejects(ctx, zk, orc):
    """2^14 rows x 135 wires (about the size of the reference's root circuit): device-resident witness, verifier accepts, a flipped
    word and a witness that breaks a copy constraint are rejected."""
    from eth_tx_proof_amd import plonk
    cfg = _cfg()
    circ, wires = plonk.synthetic_circuit(degree_bits=14, seed=14)
    pub = np.arange(4, dtype=np.uint64)
    ost = orc.PlonkState(cfg, circ)
    st = zk.PlonkState(ctx, cfg, circ)
    try:
        d = ctx.dev_alloc(wires.nbytes)
        ctx.upload(d, wires)
        proof = zk.prove_plonk(st, d, pub).copy()
        assert orc.verify_plonk(ost, pub, proof) == 0
        assert (zk.prove_plonk(st, wires, pub) == proof).all()          # host-resident witness: same proof
        bad = proof.copy(); bad[proof.size // 2] ^= np.uint64(1)
        assert orc.verify_plonk(ost, pub, bad) != 0
        w = wires.copy(); w[8, 77] ^= np.uint64(1)
        assert orc.verify_plonk(ost, pub, zk.prove_plonk(st, w, pub)) != 0
        ctx.dev_free(d)
    finally:
        st.free()


def test_plonk_error_paths(ctx, zk):
    from eth_tx_proof_amd import plonk
    import eth_tx_proof_amd as E
    cfg = _cfg()
    circ, wires = plonk.synthetic_circuit(degree_bits=5, seed=1)
    circ.c.quotient_degree_factor = 4
    with pytest.raises(zk.ZkError) as e:
        zk.PlonkState(ctx, cfg, circ)
    assert e.value.code == -1
    circ.c.quotient_degree_factor = 8
    st = zk.PlonkState(ctx, cfg, circ)
    other = E.standard_fast_config()
    st.cfg = other                                   # proving with another config than the one of the setup is refused
    with pytest.raises(zk.ZkError):
        zk.prove_plonk(st, wires, np.zeros(1, dtype=np.uint64))
    st.cfg = cfg
    st.free()


@pytest.mark.parametrize("k,num_wires,num_routed,n_muladd,n_sbox", [(2, 16, 8, 2, 1), (3, 16, 8, 2, 1), (4, 24, 16, 4, 2), (5, 40, 16, 4, 2), (8, 64, 24, 8, 8),
                                                                     (7, 135, 80, 16, 8), (10, 135, 80, 30, 20), (6, 170, 128, 20, 10), (9, 9, 8, 1, 0)])
def test_plonk_circuit_shape_sweep_bit_exact(ctx, zk, orc, k, num_wires, num_routed, n_muladd, n_sbox):
    """Circuit shapes away from the recursion config's: the smallest domains (2^2, 2^3 rows: the 8n coset is the cap), one partial-product
    chunk (8 routed wires) up to the sixteen the kernels hold (128), more wires than the recursion config, a circuit with a single gate kind.  Same proof
    words as the oracle, verifier accepts."""
    from eth_tx_proof_amd import plonk
    cfg = _cfg()
    circ, wires = plonk.synthetic_circuit(degree_bits=k, num_wires=num_wires, num_routed=num_routed, n_muladd=n_muladd, n_sbox=n_sbox, seed=100 + k)
    pub = np.array([k, num_wires, num_routed], dtype=np.uint64)
    ost = orc.PlonkState(cfg, circ)
    st = zk.PlonkState(ctx, cfg, circ)
    try:
        assert (st.digest() == ost.digest()).all()
        proof = zk.prove_plonk(st, wires, pub)
        rc, ref = orc.prove_plonk(ost, wires, pub)
        assert rc == 0 and proof.size == ref.size
        if not (proof == ref).all():
            bad = np.nonzero(proof != ref)[0]
            pytest.fail("first differing word %d of %d (%d differ)" % (bad[0], proof.size, bad.size))
        assert orc.verify_plonk(ost, pub, proof) == 0
    finally:
        st.free()


def test_public_input_gate_binds_the_proof_to_its_public_inputs(ctx, zk, orc):
    """A circuit with upstream's PublicInputGate (PUBLIC operands = the words of public_inputs_hash): same proof as the oracle, accepted;
    the same proof under other public inputs, and a witness that does not carry the hash, are rejected."""
    from eth_tx_proof_amd import plonk
    cfg = _cfg()
    circ, wires = plonk.synthetic_circuit(degree_bits=8, seed=21, public_input_gate=True)
    pub = np.array([11, 22, 33, 44, 55], dtype=np.uint64)
    pih = ctx.hash_rows(pub[None, :])[0]
    assert (pih == orc.hash_rows(pub[None, :])[0]).all()
    w = plonk.bind_public_inputs(circ, wires, pih)
    ost = orc.PlonkState(cfg, circ)
    st = zk.PlonkState(ctx, cfg, circ)
    try:
        proof = zk.prove_plonk(st, w, pub)
        rc, ref = orc.prove_plonk(ost, w, pub)
        assert rc == 0 and (proof == ref).all()
        assert orc.verify_plonk(ost, pub, proof) == 0
        other = pub.copy(); other[0] += np.uint64(1)
        assert orc.verify_plonk(ost, other, proof) != 0
        assert orc.verify_plonk(ost, pub, zk.prove_plonk(st, wires, pub)) != 0          # witness without the hash in the gate's wires
    finally:
        st.free()


@pytest.mark.parametrize("degree_bits,seed", [(5, 21), (8, 11), (11, 22), (13, 12)])
def test_recursion_shaped_circuit_bit_exact_vs_oracle(ctx, zk, orc, degree_bits, seed):
    """The gate load of the real recursion circuits (gates.py: PoseidonGate's 123 degree-7 constraints + arithmetic / reducing / base-sum
    gates behind selector filters, gate terms adding up per index, 29 program registers): GPU proof == oracle proof, word for word; the
    2^8 and 2^13 cases also against the digests committed in tests/golden/proof_digests.json."""
    import hashlib, json, os
    from eth_tx_proof_amd import plonk
    cfg = _cfg()
    circ, wires = plonk.recursion_circuit(degree_bits=degree_bits, seed=seed)
    pub = np.array([degree_bits, seed, 7], dtype=np.uint64)
    wires = plonk.bind_public_inputs(circ, wires, orc.hash_no_pad(pub))
    ost = orc.PlonkState(cfg, circ)
    st = zk.PlonkState(ctx, cfg, circ)
    try:
        assert (st.digest() == ost.digest()).all()
        proof = zk.prove_plonk(st, wires, pub)
        rc, ref = orc.prove_plonk(ost, wires, pub)
        assert rc == 0 and proof.size == ref.size
        if not (proof == ref).all():
            bad = np.nonzero(proof != ref)[0]
            pytest.fail("first differing word %d of %d (%d differ)" % (bad[0], proof.size, bad.size))
        assert orc.verify_plonk(ost, pub, proof) == 0
        gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "proof_digests.json")))["recursion_plonk_proofs"]
        for g in gold:
            if g["degree_bits"] == degree_bits and g["seed"] == seed:
                assert hashlib.sha256(proof.tobytes()).hexdigest() == g["sha256"]
        # another context of the same device proves with the same prover data (P_STATE is shared by all workers)
        other = zk.Context(0)
        try:
            assert (zk.prove_plonk(st, wires, pub, ctx=other) == proof).all()
        finally:
            other.close()
    finally:
        st.free()


def test_wide_routing_and_domain_separator(ctx, zk, orc):
    """More routed wires than 16 chunks of the quotient degree factor (the former limit: 128) and a non-empty domain separator."""
    from eth_tx_proof_amd import plonk
    cfg = _cfg()
    circ, wires = plonk.synthetic_circuit(degree_bits=7, num_wires=200, num_routed=152, seed=31)
    circ = plonk.Circuit(circ.degree_bits, circ.num_wires, circ.num_routed, circ.constants, circ.sigmas, circ.gates, domain_separator=[3, 1, 4, 1, 5, 9, 2, 6, 5])
    pub = np.array([1], dtype=np.uint64)
    ost = orc.PlonkState(cfg, circ)
    st = zk.PlonkState(ctx, cfg, circ)
    try:
        assert (st.digest() == ost.digest()).all()
        proof = zk.prove_plonk(st, wires, pub)
        rc, ref = orc.prove_plonk(ost, wires, pub)
        assert rc == 0 and (proof == ref).all() and orc.verify_plonk(ost, pub, proof) == 0
    finally:
        st.free()


def test_gate_program_compiled_at_setup_equals_interpreter_and_oracle(ctx, zk, orc, tmp_path):
    """A gate program that was NOT built with the library (136 wires: another operand mapping, so another kernel) is compiled by hiprtc
    inside zk_plonk_setup, chunked (5 816 instructions), kept in the prover-state directory, and gives the oracle's proof -- as does
    the interpreter kernel (ZK_PLONK_JIT=0, in a child process) on the same circuit."""
    import os, subprocess, sys, hashlib
    from eth_tx_proof_amd import plonk
    cfg = _cfg()
    circ, wires = plonk.recursion_circuit(degree_bits=6, seed=77, num_wires=136)
    pub = np.array([4, 5], dtype=np.uint64)
    wires = plonk.bind_public_inputs(circ, wires, orc.hash_no_pad(pub))
    ost = orc.PlonkState(cfg, circ)
    rc, ref = orc.prove_plonk(ost, wires, pub)
    assert rc == 0
    zk.prover_state_cache_dir(str(tmp_path))
    try:
        before = zk.prover_state_stats()
        st = zk.PlonkState(ctx, cfg, circ)
        after = zk.prover_state_stats()
        assert after[0] == before[0] + 1                            # one kernel compiled
        assert any(f.endswith(".co") for f in os.listdir(tmp_path))  # and persisted
        proof = zk.prove_plonk(st, wires, pub)
        st.free()
        assert (proof == ref).all()
    finally:
        zk.prover_state_cache_dir(None)
    code = ("import sys; sys.path.insert(0, %r); import numpy as np, hashlib, eth_tx_proof_amd as E; from eth_tx_proof_amd import lib as zl, plonk;"
            "c = zl.Context(0); circ, w = plonk.recursion_circuit(degree_bits=6, seed=77, num_wires=136);"
            "w = plonk.bind_public_inputs(circ, w, %r); st = zl.PlonkState(c, E.standard_recursion_config(), circ);"
            "print(hashlib.sha256(zl.prove_plonk(st, w, np.array([4, 5], dtype=np.uint64)).tobytes()).hexdigest())"
            % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), [int(x) for x in orc.hash_no_pad(pub)]))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, ZK_PLONK_JIT="0"), timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.strip().splitlines()[-1] == hashlib.sha256(ref.tobytes()).hexdigest()


def test_round3_launch_shapes_are_switches_not_results(ctx, zk):
    """The round-3 launch shapes -- an LDE's cosets in one launch per pass, 2^11 / 2^12 transforms as one whole-row pass, permutation rows
    spread over the workgroup -- must not show in any result: a child process with the switches OFF (they are read once per process)
    commits the same columns and proves the same circuit; caps and proof words must equal this process's."""
    import hashlib, json, os, subprocess, sys
    from eth_tx_proof_amd import plonk
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = r"""
import hashlib, json, sys
import numpy as np
sys.path.insert(0, %r)
import eth_tx_proof_amd as E
from eth_tx_proof_amd import lib as zl, plonk
ctx = zl.Context(0)
out = {}
rng = np.random.default_rng(5)
for k, c in ((5, 3), (11, 9), (12, 20), (13, 7)):
    vals = rng.integers(0, 0xFFFFFFFF00000001, size=(c, 1 << k), dtype=np.uint64)
    b = ctx.batch_from_values(vals, rate_bits=3, cap_height=4)
    out["cap_%%d" %% k] = hashlib.sha256(np.ascontiguousarray(b.cap).tobytes()).hexdigest()
    b.free()
cfg = E.standard_recursion_config()
for k in (6, 12):
    circ, wires = plonk.synthetic_circuit(degree_bits=k, seed=k)
    st = zl.PlonkState(ctx, cfg, circ)
    out["plonk_%%d" %% k] = hashlib.sha256(zl.prove_plonk(st, wires, np.arange(3, dtype=np.uint64)).tobytes()).hexdigest()
    st.free()
print("RESULT " + json.dumps(out))
""" % root

    def run(env_extra):
        r = subprocess.run([sys.executable, "-c", script], env=dict(os.environ, **env_extra), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        return json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    on = run({})
    off = run({"ZK_NTT_COSET_MERGE": "0", "ZK_NTT_ROW_W": "10", "ZK_PLONK_PERM_SPLIT": "0"})
    assert on == off
    # and this process (defaults) agrees with both on one of them
    rng = np.random.default_rng(5)
    vals = rng.integers(0, 0xFFFFFFFF00000001, size=(3, 1 << 5), dtype=np.uint64)
    b = ctx.batch_from_values(vals, rate_bits=3, cap_height=4)
    assert hashlib.sha256(np.ascontiguousarray(b.cap).tobytes()).hexdigest() == on["cap_5"]
    b.free()
