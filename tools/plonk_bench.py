#!/usr/bin/env python3
"""Plonk proof of a recursion-shaped circuit (gates.py gate set; recursion config: 135 wires, 80 routed, rate_bits 3, 28 queries) on one
MI355X: ms per proof with the ctx profiler, beside the oracle (CPU port) on the same circuit.
usage: python tools/plonk_bench.py [degree_bits=13] [reps=5] [toy]     (toy: the round-2 synthetic_circuit with its 24 gate terms)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import eth_tx_proof_amd as E
from eth_tx_proof_amd import lib as zl, plonk
k = int(sys.argv[1]) if len(sys.argv) > 1 else 13
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
cfg = E.standard_recursion_config()
toy = len(sys.argv) > 3 and sys.argv[3] == "toy"
pub = np.arange(4, dtype=np.uint64)
ctx = zl.Context(0)
if toy:
    circ, wires = plonk.synthetic_circuit(degree_bits=k, seed=k)
else:
    circ, wires = plonk.recursion_circuit(degree_bits=k, seed=k)
    wires = plonk.bind_public_inputs(circ, wires, plonk.SyntheticRecursion.public_inputs_hash(ctx, pub))
st = zl.PlonkState(ctx, cfg, circ)
d = ctx.dev_alloc(wires.nbytes); ctx.upload(d, wires)
p = zl.prove_plonk(st, d, pub).copy()
t0 = time.time()
for _ in range(reps):
    zl.prove_plonk(st, d, pub)
print("profiler off: %.2f ms per proof" % ((time.time() - t0) / reps * 1e3))
ctx.profile(True); ctx.profile_reset()
t0 = time.time()
for _ in range(reps):
    zl.prove_plonk(st, d, pub)
dt = (time.time() - t0) / reps
print("plonk proof 2^%d rows x 135 wires, %s gates: %.2f ms on the GPU with the hipEvent profiler on (device-resident witness), %d words" % (k, "toy" if toy else "recursion-shaped", dt * 1e3, p.size))
for name, (l, ms) in sorted(ctx.profile_report().items(), key=lambda kv: -kv[1][1])[:16]:
    print("  %-28s %5d launches %8.3f ms/proof" % (name, l // reps, ms / reps))
if not os.environ.get("NO_ORACLE"):
    import orc
    ost = orc.PlonkState(cfg, circ)
    t0 = time.time(); rc, ref = orc.prove_plonk(ost, wires, pub); dtc = time.time() - t0
    print("oracle (CPU port, %d threads): %.2f s, same words: %s, verifier: %d" % (orc.lib().orc_get_threads(), dtc, bool((ref == p).all()), orc.verify_plonk(ost, pub, p)))
