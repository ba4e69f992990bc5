#!/usr/bin/env python3
"""Writes tests/golden/proof_digests.json: sha256 of whole proofs the ORACLE produces for fixed seeds (STARK unit proofs, a proof with
a logUp lookup, a Plonk-shaped proof).  These are DERIVED values -- regression pins of this repo's own oracle, NOT upstream vectors
(the reference holds none): they make an accidental change of the oracle, of the proof layout or of a convention
visible in the CPU suite, and they are what the GPU proofs are compared with on the GPU box (tests compare GPU == oracle bit for bit)."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np  # noqa: E402
import eth_tx_proof_amd as E  # noqa: E402
from eth_tx_proof_amd import air, plonk  # noqa: E402
import orc  # noqa: E402


def digests():
    out = {"note": "derived: sha256 over the little-endian proof words produced by oracle/ (parity unpinned, regression pins only)"}
    cfg = E.standard_fast_config()
    stark = []
    for n_cols, n_helpers, log_n, seed in ((6, 2, 6, 101), (93, 19, 8, 102), (7, 0, 5, 103), (21, 4, 10, 104)):
        prog = air.cpu_like_air(n_cols, n_helpers)
        tr, pub = orc.synth_trace(n_cols, log_n, seed)
        rc, p = orc.prove(cfg, [(prog, log_n, tr, pub)])
        assert rc == 0 and orc.verify(cfg, [(prog, log_n, None, pub)], p) == 0
        stark.append({"n_cols": n_cols, "n_helpers": n_helpers, "log_n": log_n, "seed": seed, "words": int(p.size), "sha256": hashlib.sha256(p.tobytes()).hexdigest()})
    out["stark_unit_proofs"] = stark
    prog, tr, pub = air.lookup_air(5, (1, 3), log_n=7, seed=7)
    rc, p = orc.prove(cfg, [(prog, 7, tr, pub)])
    assert rc == 0 and orc.verify(cfg, [(prog, 7, None, pub)], p) == 0
    out["lookup_proof"] = {"n_looked": 5, "filtered": [1, 3], "log_n": 7, "seed": 7, "words": int(p.size), "sha256": hashlib.sha256(p.tobytes()).hexdigest()}
    rcfg = E.standard_recursion_config()
    circ, wires = plonk.synthetic_circuit(degree_bits=7, seed=5)
    st = orc.PlonkState(rcfg, circ)
    pubs = np.array([1, 2, 3], dtype=np.uint64)
    rc, p = orc.prove_plonk(st, wires, pubs)
    assert rc == 0 and orc.verify_plonk(st, pubs, p) == 0
    out["plonk_proof"] = {"degree_bits": 7, "seed": 5, "public_inputs": [1, 2, 3], "words": int(p.size), "sha256": hashlib.sha256(p.tobytes()).hexdigest(),
                          "circuit_digest": ["%016x" % int(x) for x in st.digest()]}
    # circuits of the recursion shape (gates.py: PoseidonGate + arithmetic / reducing / base-sum gates behind selector filters)
    rec = []
    for k, seed in ((8, 11), (13, 12)):
        circ, wires = plonk.recursion_circuit(degree_bits=k, seed=seed)
        st = orc.PlonkState(rcfg, circ)
        pubs = np.array([k, seed, 7], dtype=np.uint64)
        w = plonk.bind_public_inputs(circ, wires, orc.hash_no_pad(pubs))
        rc, p = orc.prove_plonk(st, w, pubs)
        assert rc == 0 and orc.verify_plonk(st, pubs, p) == 0
        rec.append({"degree_bits": k, "seed": seed, "public_inputs": [k, seed, 7], "words": int(p.size), "sha256": hashlib.sha256(p.tobytes()).hexdigest(),
                    "circuit_digest": ["%016x" % int(x) for x in st.digest()]})
    out["recursion_plonk_proofs"] = rec
    return out


if __name__ == "__main__":
    json.dump(digests(), open(os.path.join(ROOT, "tests", "golden", "proof_digests.json"), "w"), indent=1)
    print("written")
