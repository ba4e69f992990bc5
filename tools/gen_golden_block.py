#!/usr/bin/env python3
"""BASELINE configs[4] through the ORACLE at full size: a 16-transaction block -- every transaction the seven tables of configs[3] at their real
sizes with shrink chains and root proof (orc.prove_root), the leader's balanced pairwise fold with the aggregation circuit, the block circuit
(/root/reference/leader/src/prover.rs:20-43) -- and the sha256 + length of every proof on the way, for tests/test_gpu_big.py.
Four DISTINCT transactions (trace seeds BLOCK_SEED + 100000 u + 1000 t) repeated four times make the 16: the oracle needs 4 minutes per distinct
transaction on 8 threads, the fold and the block proof differ for every pair anyway.  DERIVED values (regression pins of oracle/, parity unpinned).
usage: tools/gen_golden_block.py [--out tests/golden/proof_digests_block.json] [--threads N] [--units 16] [--distinct 4]"""
import argparse, hashlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np  # noqa: E402
import eth_tx_proof_amd as E  # noqa: E402
from eth_tx_proof_amd import air, plonk  # noqa: E402
from eth_tx_proof_amd._abi import standard_recursion_config  # noqa: E402
import orc  # noqa: E402

BLOCK_SEED = 0xB10C0000


def sha(p):
    return hashlib.sha256(np.ascontiguousarray(p, dtype=np.uint64).tobytes()).hexdigest()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "proof_digests_block.json"))
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--units", type=int, default=16)
    ap.add_argument("--distinct", type=int, default=4)
    args = ap.parse_args()
    orc.lib().orc_set_threads(args.threads or min(128, os.cpu_count() or 1))
    cfg, rcfg = E.standard_fast_config(), standard_recursion_config()
    rec = plonk.SyntheticRecursion()
    t0 = time.time()
    roots = []
    for u in range(args.distinct):
        items = []
        for t, (w, lg) in enumerate(zip(air.ZKEVM_TABLE_WIDTHS, air.ZKEVM_CFG4_LOG_SIZES)):
            tr, pub = orc.synth_trace(w, lg, BLOCK_SEED + 100000 * u + 1000 * t)
            items.append((air.cpu_like_air(w, min(19, w // 4)), lg, tr, pub))
        root, _, _ = orc.prove_root(cfg, rcfg, items, rec)
        roots.append(root)
        print("[gen_golden_block] transaction %d: root %d words, %.0f s so far" % (u, root.size, time.time() - t0), file=sys.stderr, flush=True)
        del items
    level = [roots[u % args.distinct] for u in range(args.units)]
    out = {"note": "derived: sha256 over the little-endian proof words produced by oracle/ (parity unpinned, regression pins only); made by tools/gen_golden_block.py",
           "block_seed": BLOCK_SEED, "units": args.units, "distinct": args.distinct, "widths": list(air.ZKEVM_TABLE_WIDTHS), "log_sizes": list(air.ZKEVM_CFG4_LOG_SIZES),
           "roots": [{"words": int(r.size), "sha256": sha(r)} for r in roots], "fold_levels": []}
    circ_a, wires_a = rec.circuit("agg")
    while len(level) > 1:
        nxt = []
        for j in range(0, len(level) - 1, 2):
            nxt.append(orc._prove_bound(rcfg, circ_a, wires_a, np.concatenate([orc.words_digest(level[j]), orc.words_digest(level[j + 1])])))
        if len(level) & 1:
            nxt.append(level[-1])
        out["fold_levels"].append([{"words": int(p.size), "sha256": sha(p)} for p in nxt])
        level = nxt
        print("[gen_golden_block] fold level of %d proofs, %.0f s so far" % (len(level), time.time() - t0), file=sys.stderr, flush=True)
    circ_b, wires_b = rec.circuit("block")
    pubs = np.concatenate([np.array([1], dtype=np.uint64), orc.words_digest(level[0]), np.zeros(4, dtype=np.uint64)])
    block = orc._prove_bound(rcfg, circ_b, wires_b, pubs)
    out["block"] = {"words": int(block.size), "sha256": sha(block), "public_inputs": [int(x) for x in pubs]}
    out["oracle_seconds"] = round(time.time() - t0, 1)
    json.dump(out, open(args.out, "w"), indent=1)
    print(json.dumps(out)[:400])


if __name__ == "__main__":
    main()
