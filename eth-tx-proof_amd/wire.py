"""Wire format of proofs: proof words <-> the serde-JSON shape of upstream's proof structs.

The reference writes proofs with `serde_json::to_vec(&proof)` (/root/reference/leader/src/main.rs:60); the structs live in
starky 0.4.0 / plonky2 0.2.2 (Cargo.lock:4529-4531, :3441-3443), source absent, so field names and nesting below are
RECALLED [U] and cannot be checked against a reference run here.  What is pinned: `to_dict`/`from_dict` are exact inverses
on the words the prover emits (tests/test_wire_cpu.py), so a future diff against a real reference proof needs only this
module to change.

  StarkProof { trace_cap, auxiliary_polys_cap, quotient_polys_cap, openings, opening_proof }
  MerkleCap = [HashOut], HashOut = {"elements": [u64; 4]}, extension element = [u64; 2]
  StarkOpeningSet { local_values, next_values, auxiliary_polys, auxiliary_polys_next, ctl_zs_first, quotient_polys }
  FriProof { commit_phase_merkle_caps, query_round_proofs: [{initial_trees_proof: {evals_proofs: [[leaf, {siblings}]]},
             steps: [{evals, merkle_proof: {siblings}}]}], final_poly: {coeffs}, pow_witness }
upstream serialises the raw (possibly non-canonical) u64 of a field element; the prover always emits canonical words.
"""
import json

import numpy as np

MAGIC = 0x31465250474C4B5A


def _arities(cfg, degree_bits):
    out, d = [], degree_bits
    while d > cfg.fri_final_poly_bits and d + cfg.rate_bits - cfg.fri_arity_bits >= cfg.cap_height:
        out.append(cfg.fri_arity_bits)
        d -= cfg.fri_arity_bits
    return out


class _R:
    def __init__(self, w):
        self.w, self.p = [int(x) for x in w], 0

    def take(self, n):
        if self.p + n > len(self.w):
            raise ValueError("proof truncated")
        out = self.w[self.p:self.p + n]
        self.p += n
        return out

    def one(self):
        return self.take(1)[0]


def _hashes(words):
    return [{"elements": words[i:i + 4]} for i in range(0, len(words), 4)]


def _exts(words):
    return [words[i:i + 2] for i in range(0, len(words), 2)]


def to_dict(proof_words, cfg):
    """-> {"tables": [{"degree_bits", "proof": StarkProof-shaped dict}]}"""
    r = _R(proof_words)
    if r.one() != MAGIC:
        raise ValueError("bad magic")
    nt = r.one()
    capw = 4 << cfg.cap_height
    nch = cfg.num_challenges
    tables = []
    for _ in range(nt):
        log_n, C, A = r.one(), r.one(), r.one()
        Q = nch << cfg.rate_bits
        lgL = log_n + cfg.rate_bits
        trace_cap = _hashes(r.take(capw))
        aux_cap = _hashes(r.take(capw)) if A else None
        quot_cap = _hashes(r.take(capw))
        local, nxt = _exts(r.take(2 * C)), _exts(r.take(2 * C))
        aux, auxn = _exts(r.take(2 * A)), _exts(r.take(2 * A))
        zs = r.take(nch) if A else None
        quot = _exts(r.take(2 * Q))
        ar = _arities(cfg, log_n)
        if r.one() != len(ar):
            raise ValueError("unexpected number of FRI layers")
        caps = [_hashes(r.take(capw)) for _ in ar]
        flen = r.one()
        final = _exts(r.take(2 * flen))
        pow_witness = r.one()
        npath = lgL - cfg.cap_height
        rounds = []
        for _q in range(cfg.num_query_rounds):
            evals_proofs = []
            for ncols in ([C, A, Q] if A else [C, Q]):
                leaf = r.take(ncols)
                evals_proofs.append([leaf, {"siblings": _hashes(r.take(4 * npath))}])
            steps, lg = [], lgL
            for ab in ar:
                ev = _exts(r.take(2 << ab))
                lg -= ab
                steps.append({"evals": ev, "merkle_proof": {"siblings": _hashes(r.take(4 * (lg - cfg.cap_height)))}})
            rounds.append({"initial_trees_proof": {"evals_proofs": evals_proofs}, "steps": steps})
        tables.append({"degree_bits": log_n, "proof": {
            "trace_cap": trace_cap, "auxiliary_polys_cap": aux_cap, "quotient_polys_cap": quot_cap,
            "openings": {"local_values": local, "next_values": nxt, "auxiliary_polys": aux if A else None,
                         "auxiliary_polys_next": auxn if A else None, "ctl_zs_first": zs, "quotient_polys": quot},
            "opening_proof": {"commit_phase_merkle_caps": caps, "query_round_proofs": rounds,
                              "final_poly": {"coeffs": final}, "pow_witness": pow_witness}}})
    if r.p != len(r.w):
        raise ValueError("trailing words")
    return {"tables": tables}


def from_dict(d, cfg):
    """exact inverse of to_dict"""
    w = [MAGIC, len(d["tables"])]

    def flat_h(hs):
        return [x for h in hs for x in h["elements"]]

    def flat_e(es):
        return [x for e in es for x in e]
    for t in d["tables"]:
        p = t["proof"]
        o = p["openings"]
        A = len(o["auxiliary_polys"]) if o["auxiliary_polys"] is not None else 0
        w += [t["degree_bits"], len(o["local_values"]), A]
        w += flat_h(p["trace_cap"])
        if A:
            w += flat_h(p["auxiliary_polys_cap"])
        w += flat_h(p["quotient_polys_cap"])
        w += flat_e(o["local_values"]) + flat_e(o["next_values"])
        if A:
            w += flat_e(o["auxiliary_polys"]) + flat_e(o["auxiliary_polys_next"]) + list(o["ctl_zs_first"])
        w += flat_e(o["quotient_polys"])
        f = p["opening_proof"]
        w += [len(f["commit_phase_merkle_caps"])]
        for c in f["commit_phase_merkle_caps"]:
            w += flat_h(c)
        w += [len(f["final_poly"]["coeffs"])] + flat_e(f["final_poly"]["coeffs"]) + [f["pow_witness"]]
        for q in f["query_round_proofs"]:
            for leaf, mp in q["initial_trees_proof"]["evals_proofs"]:
                w += list(leaf) + flat_h(mp["siblings"])
            for s in q["steps"]:
                w += flat_e(s["evals"]) + flat_h(s["merkle_proof"]["siblings"])
    return np.array(w, dtype=np.uint64)


def to_json(proof_words, cfg):
    return json.dumps(to_dict(proof_words, cfg), separators=(",", ":"))


def from_json(text, cfg):
    return from_dict(json.loads(text), cfg)


# ------------------------------------------------------------------------------------------------ Plonk proofs (the recursion half)
# plonky2 0.2.2 plonk::proof (names RECALLED, as above):
#   ProofWithPublicInputs { proof: Proof, public_inputs: [u64] }
#   Proof { wires_cap, plonk_zs_partial_products_cap, quotient_polys_cap, openings: OpeningSet, opening_proof: FriProof }
#   OpeningSet { constants, plonk_sigmas, wires, plonk_zs, plonk_zs_next, partial_products, quotient_polys, lookup_zs, lookup_zs_next }
# and the reference's own wrappers (proof_gen::proof_types, what `serde_json::to_vec(&proof)` at /root/reference/leader/src/main.rs:60 writes):
#   GeneratedTxnProof / GeneratedAggProof { p_vals: PublicValues, intern: ProofWithPublicInputs },  GeneratedBlockProof { b_height, intern }
# The proof words also carry what upstream keeps in CommonCircuitData (degree_bits, wire / constant counts).  plonk_to_dict returns them
# under "common" beside the upstream-shaped dict; the JSON of the reference's wrappers below has upstream's keys ONLY (no "common":
# ADVICE r3) -- reading it back derives the four numbers from the proof itself (`common_of`: opening counts, Merkle path length).
PLONK_MAGIC = 0x314B4C504C474B5A


def plonk_to_dict(proof_words, cfg, public_inputs=()):
    """-> {"common": {..}, "proof_with_pis": ProofWithPublicInputs-shaped dict}; cfg: the recursion config the proof was made with"""
    r = _R(proof_words)
    if r.one() != PLONK_MAGIC:
        raise ValueError("bad magic")
    k, W, R, NC = r.one(), r.one(), r.one(), r.one()
    nch, qdf, capw = cfg.num_challenges, 1 << cfg.rate_bits, 4 << cfg.cap_height
    if R == 0 or R % qdf:
        raise ValueError("routed wires not a multiple of the quotient degree factor")
    npp = R // qdf - 1
    lgL = k + cfg.rate_bits
    wires_cap, zs_cap, quot_cap = _hashes(r.take(capw)), _hashes(r.take(capw)), _hashes(r.take(capw))
    o = {"constants": _exts(r.take(2 * NC)), "plonk_sigmas": _exts(r.take(2 * R)), "wires": _exts(r.take(2 * W)),
         "plonk_zs": _exts(r.take(2 * nch)), "plonk_zs_next": _exts(r.take(2 * nch)), "partial_products": _exts(r.take(2 * nch * npp)),
         "quotient_polys": _exts(r.take(2 * nch * qdf)), "lookup_zs": [], "lookup_zs_next": []}
    ar = _arities(cfg, k)
    if r.one() != len(ar):
        raise ValueError("unexpected number of FRI layers")
    caps = [_hashes(r.take(capw)) for _ in ar]
    flen = r.one()
    final = _exts(r.take(2 * flen))
    pow_witness = r.one()
    npath = lgL - cfg.cap_height
    rounds = []
    for _q in range(cfg.num_query_rounds):
        evals_proofs = []
        for ncols in (NC + R, W, nch * (1 + npp), nch * qdf):
            leaf = r.take(ncols)
            evals_proofs.append([leaf, {"siblings": _hashes(r.take(4 * npath))}])
        steps, lg = [], lgL
        for ab in ar:
            ev = _exts(r.take(2 << ab))
            lg -= ab
            steps.append({"evals": ev, "merkle_proof": {"siblings": _hashes(r.take(4 * (lg - cfg.cap_height)))}})
        rounds.append({"initial_trees_proof": {"evals_proofs": evals_proofs}, "steps": steps})
    if r.p != len(r.w):
        raise ValueError("trailing words")
    return {"common": {"degree_bits": k, "num_wires": W, "num_routed_wires": R, "num_constants": NC},
            "proof_with_pis": {"proof": {"wires_cap": wires_cap, "plonk_zs_partial_products_cap": zs_cap, "quotient_polys_cap": quot_cap, "openings": o,
                                         "opening_proof": {"commit_phase_merkle_caps": caps, "query_round_proofs": rounds,
                                                           "final_poly": {"coeffs": final}, "pow_witness": pow_witness}},
                               "public_inputs": [int(x) for x in public_inputs]}}


def plonk_from_dict(d):
    """exact inverse of plonk_to_dict: -> (proof words, public inputs)"""
    c, pw = d["common"], d["proof_with_pis"]
    p, o = pw["proof"], pw["proof"]["openings"]
    fh = lambda hs: [x for h in hs for x in h["elements"]]
    fe = lambda es: [x for e in es for x in e]
    w = [PLONK_MAGIC, c["degree_bits"], c["num_wires"], c["num_routed_wires"], c["num_constants"]]
    w += fh(p["wires_cap"]) + fh(p["plonk_zs_partial_products_cap"]) + fh(p["quotient_polys_cap"])
    for key in ("constants", "plonk_sigmas", "wires", "plonk_zs", "plonk_zs_next", "partial_products", "quotient_polys"):
        w += fe(o[key])
    f = p["opening_proof"]
    w += [len(f["commit_phase_merkle_caps"])]
    for cap in f["commit_phase_merkle_caps"]:
        w += fh(cap)
    w += [len(f["final_poly"]["coeffs"])] + fe(f["final_poly"]["coeffs"]) + [f["pow_witness"]]
    for q in f["query_round_proofs"]:
        for leaf, mp in q["initial_trees_proof"]["evals_proofs"]:
            w += list(leaf) + fh(mp["siblings"])
        for s in q["steps"]:
            w += fe(s["evals"]) + fh(s["merkle_proof"]["siblings"])
    return np.array(w, dtype=np.uint64), np.array(pw["public_inputs"], dtype=np.uint64)


def common_of(proof_with_pis, cfg=None):
    """CommonCircuitData's four numbers from a ProofWithPublicInputs-shaped dict: opening counts give the wire / routed / constant
    counts, the length of a query's initial Merkle paths gives degree_bits.  cfg: the recursion config (default: the standard one)."""
    if cfg is None:
        from ._abi import standard_recursion_config
        cfg = standard_recursion_config()
    o = proof_with_pis["proof"]["openings"]
    q0 = proof_with_pis["proof"]["opening_proof"]["query_round_proofs"][0]
    npath = len(q0["initial_trees_proof"]["evals_proofs"][0][1]["siblings"])
    return {"degree_bits": npath + cfg.cap_height - cfg.rate_bits, "num_wires": len(o["wires"]), "num_routed_wires": len(o["plonk_sigmas"]),
            "num_constants": len(o["constants"])}


def block_proof_to_json(block, cfg, public_inputs):
    """GeneratedBlockProof { b_height, intern } as the leader prints it (/root/reference/leader/src/main.rs:60): upstream's two keys"""
    d = plonk_to_dict(block.intern, cfg, public_inputs)
    return json.dumps({"b_height": int(block.b_height), "intern": d["proof_with_pis"]}, separators=(",", ":"))


def block_proof_from_json(text, cfg=None):
    """-> (b_height, proof words, public inputs)"""
    j = json.loads(text)
    words, pis = plonk_from_dict({"common": common_of(j["intern"], cfg), "proof_with_pis": j["intern"]})
    return int(j["b_height"]), words, pis


def txn_proof_to_json(agg_proof, cfg, public_inputs):
    """GeneratedTxnProof / GeneratedAggProof { p_vals, intern }: p_vals here = the tables' public inputs the unit carried"""
    d = plonk_to_dict(agg_proof.proof, cfg, public_inputs)
    return json.dumps({"p_vals": [[int(x) for x in pv] for pv in agg_proof.public_values], "intern": d["proof_with_pis"]},
                      separators=(",", ":"))


def txn_proof_from_json(text, cfg=None):
    """-> (p_vals, proof words, public inputs)"""
    j = json.loads(text)
    words, pis = plonk_from_dict({"common": common_of(j["intern"], cfg), "proof_with_pis": j["intern"]})
    return j["p_vals"], words, pis
