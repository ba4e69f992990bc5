#!/usr/bin/env python3
"""BASELINE configs[3]: seven-table unit (log sizes 15/9/12/14/9/12/17, recalled zkEVM widths) on one MI355X.
Prints the latency of one unit alone with its kernel table (hipEvent profiler on), then -- like bench.py's headline -- the rate with
`--inflight` units being proved at once on their own contexts/streams (profiler off): ms per unit = wall time / units."""
import argparse, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import eth_tx_proof_amd as E
from eth_tx_proof_amd import air, lib as zl

ap = argparse.ArgumentParser()
ap.add_argument("--inflight", type=int, default=4)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--top", type=int, default=12)
ap.add_argument("--jit", action="store_true", help="compile the constraint kernels of all seven tables first (zk_prover_state_prepare; minutes for the wide ones)")
args = ap.parse_args()
cfg = E.standard_fast_config()


def unit(ctx):
    items = []
    for t, (w, lg) in enumerate(zip(air.ZKEVM_TABLE_WIDTHS, air.ZKEVM_CFG4_LOG_SIZES)):
        prog = air.cpu_like_air(w, min(19, w // 4))
        d, pub = zl.synth_trace_dev(ctx, w, lg, 0x5EED0000 + 1000 * t)
        items.append((prog, lg, d, pub))
    return items


ctxs = [zl.Context(0) for _ in range(max(1, args.inflight))]
units = [unit(c) for c in ctxs]
if args.jit:
    t0 = time.time()
    zl.prover_state_prepare(ctxs[0], cfg, units[0])
    print("prover state prepared (constraint kernels compiled / loaded) in %.1f s: %s" % (time.time() - t0, zl.prover_state_stats()), flush=True)
first = zl.prove_tables(ctxs[0], cfg, units[0])
for c, u in zip(ctxs[1:], units[1:]):
    assert np.array_equal(zl.prove_tables(c, cfg, u), first)        # same seeds: every context must produce the same proof
ctx = ctxs[0]
t0 = time.time()
for _ in range(args.reps):
    p = zl.prove_tables(ctx, cfg, units[0])
print("cfg4 unit alone, profiler off: %.2f ms" % ((time.time() - t0) / args.reps * 1e3))
ctx.profile(True); ctx.profile_reset()
t0 = time.time()
for _ in range(args.reps):
    p = zl.prove_tables(ctx, cfg, units[0])
dt = (time.time() - t0) / args.reps
ctx.profile(False)
print("cfg4 unit alone, hipEvent profiler on: %.2f ms, proof %d words" % (dt * 1e3, p.size))
for k, (l, ms) in sorted(ctx.profile_report().items(), key=lambda kv: -kv[1][1])[:args.top]:
    print("  %-28s %5d launches %8.3f ms/unit" % (k, l // args.reps, ms / args.reps))
if args.inflight > 1:
    def work(c, u):
        for _ in range(args.reps):
            zl.prove_tables(c, cfg, u)
    th = [threading.Thread(target=work, args=(c, u)) for c, u in zip(ctxs, units)]
    t0 = time.time()
    for t in th: t.start()
    for t in th: t.join()
    dt = time.time() - t0
    n = args.reps * len(ctxs)
    print("cfg4 with %d units in flight: %.2f ms/unit (%.1f units/s, %d units in %.3f s)" % (len(ctxs), dt / n * 1e3, n / dt, n, dt))
