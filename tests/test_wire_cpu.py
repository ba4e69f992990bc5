"""Host logic on CPU: proof words <-> upstream-shaped serde JSON are exact inverses, on proofs made by the oracle."""
import json

import numpy as np
import pytest


@pytest.mark.parametrize("n_cols,n_helpers,log_n", [(6, 2, 6), (7, 0, 5), (9, 3, 10)])
def test_wire_round_trip(orc, n_cols, n_helpers, log_n):
    import eth_tx_proof_amd as E
    from eth_tx_proof_amd import air, wire
    cfg = E.standard_fast_config()
    prog = air.cpu_like_air(n_cols, n_helpers)
    tr, pub = orc.synth_trace(n_cols, log_n, 5)
    rc, proof = orc.prove(cfg, [(prog, log_n, tr, pub)])
    assert rc == 0
    d = wire.to_dict(proof, cfg)
    p = d["tables"][0]["proof"]
    assert len(p["trace_cap"]) == 16 and len(p["opening_proof"]["query_round_proofs"]) == 84
    assert (p["auxiliary_polys_cap"] is None) == (n_helpers == 0)
    assert len(p["openings"]["local_values"]) == n_cols and len(p["openings"]["quotient_polys"]) == 4
    text = wire.to_json(proof, cfg)
    back = wire.from_json(text, cfg)
    assert back.dtype == np.uint64 and (back == proof).all()
    assert orc.verify(cfg, [(prog, log_n, None, pub)], back) == 0
    # a proof re-serialised after editing one opening no longer verifies
    d["tables"][0]["proof"]["openings"]["local_values"][0][0] ^= 1
    assert orc.verify(cfg, [(prog, log_n, None, pub)], wire.from_dict(d, cfg)) != 0


def test_wire_multi_table_and_errors(orc):
    import eth_tx_proof_amd as E
    from eth_tx_proof_amd import air, wire
    cfg = E.standard_fast_config()
    items = []
    for t, (w, lg, h) in enumerate(((9, 6, 2), (4, 5, 0))):
        prog = air.cpu_like_air(w, h)
        tr, pub = orc.synth_trace(w, lg, t)
        items.append((prog, lg, tr, pub))
    rc, proof = orc.prove(cfg, items)
    assert rc == 0
    d = wire.to_dict(proof, cfg)
    assert [t["degree_bits"] for t in d["tables"]] == [6, 5]
    assert (wire.from_dict(json.loads(json.dumps(d)), cfg) == proof).all()
    with pytest.raises(ValueError):
        wire.to_dict(proof[:-1], cfg)
    with pytest.raises(ValueError):
        wire.to_dict(np.concatenate([proof, proof[:1]]), cfg)
    bad = proof.copy(); bad[0] ^= np.uint64(1)
    with pytest.raises(ValueError):
        wire.to_dict(bad, cfg)


def test_plonk_wire_round_trip(orc):
    """ProofWithPublicInputs / GeneratedBlockProof shape for the Plonk half: exact inverse on the oracle's proof of a recursion-shaped
    circuit, the re-read proof verifies, an edited opening or public input does not."""
    import eth_tx_proof_amd as E
    from eth_tx_proof_amd import ops, plonk, wire
    cfg = E.standard_recursion_config()
    circ, wires = plonk.recursion_circuit(degree_bits=6, seed=13)
    pubs = np.array([3, 1, 4, 1, 5], dtype=np.uint64)
    w = plonk.bind_public_inputs(circ, wires, orc.hash_no_pad(pubs))
    st = orc.PlonkState(cfg, circ)
    rc, proof = orc.prove_plonk(st, w, pubs)
    assert rc == 0
    d = wire.plonk_to_dict(proof, cfg, pubs)
    p = d["proof_with_pis"]["proof"]
    assert len(p["wires_cap"]) == 16 and len(p["openings"]["wires"]) == 135 and len(p["openings"]["plonk_sigmas"]) == 80
    assert len(p["openings"]["partial_products"]) == 2 * 9 and len(p["openings"]["quotient_polys"]) == 16
    assert len(p["opening_proof"]["query_round_proofs"]) == 28 and len(p["opening_proof"]["query_round_proofs"][0]["initial_trees_proof"]["evals_proofs"]) == 4
    back, pis = wire.plonk_from_dict(json.loads(json.dumps(d)))
    assert (back == proof).all() and (pis == pubs).all()
    assert orc.verify_plonk(st, pis, back) == 0
    blk = ops.GeneratedBlockProof(7, proof)
    h, words, pis2 = wire.block_proof_from_json(wire.block_proof_to_json(blk, cfg, pubs))
    assert h == 7 and (words == proof).all() and (pis2 == pubs).all()
    # upstream's keys only: GeneratedBlockProof { b_height, intern }, GeneratedTxnProof { p_vals, intern }
    assert sorted(json.loads(wire.block_proof_to_json(blk, cfg, pubs))) == ["b_height", "intern"]
    txt = wire.txn_proof_to_json(ops.AggregatableProof("Txn", proof, [pubs]), cfg, pubs)
    assert sorted(json.loads(txt)) == ["intern", "p_vals"]
    pv, words, pis2 = wire.txn_proof_from_json(txt, cfg)
    assert (words == proof).all() and (pis2 == pubs).all() and pv == [[int(x) for x in pubs]]
    d["proof_with_pis"]["proof"]["openings"]["wires"][5][1] ^= 1
    assert orc.verify_plonk(st, pubs, wire.plonk_from_dict(d)[0]) != 0
    d["proof_with_pis"]["proof"]["openings"]["wires"][5][1] ^= 1
    d["proof_with_pis"]["public_inputs"][0] ^= 1
    bw, bp = wire.plonk_from_dict(d)
    assert orc.verify_plonk(st, bp, bw) != 0
    with pytest.raises(ValueError):
        wire.plonk_to_dict(proof[:-3], cfg)
