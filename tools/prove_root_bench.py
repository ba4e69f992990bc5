#!/usr/bin/env python3
"""One transaction end to end on one MI355X: TxProof.execute = prove_root (/root/reference/ops/src/lib.rs:52) over BASELINE configs[3]
-- the seven tables at their real sizes (STARK proofs on one challenger), each table's initial wrapper and shrinking wrappers (until
degree_bits <= 13), the root proof -- with plonk.SyntheticRecursion's stand-in circuits at the estimated sizes of the real ones
(recursion-shaped gate load, gates.py).  Prints the STARK / shrink / root split of one unit alone, the kernel table of the Plonk half,
and the rate with `--inflight` units in flight (one worker thread + context each, like paladin --num-workers)."""
import argparse, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import eth_tx_proof_amd as E
from eth_tx_proof_amd import air, lib as zl, ops, plonk

ap = argparse.ArgumentParser()
ap.add_argument("--inflight", type=int, default=4)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--top", type=int, default=14)
ap.add_argument("--jit", action="store_true", help="compile the constraint kernels of the seven tables first (minutes for the wide ones)")
ap.add_argument("--size-shift", type=int, default=0)
ap.add_argument("--chain-streams", type=int, default=3, help="shrink chains of one transaction proved side by side by its worker")
args = ap.parse_args()

t0 = time.time()
rec = plonk.SyntheticRecursion(size_shift=args.size_shift)
st = ops.set_prover_state_from_config(0, recursion=rec, chain_streams=args.chain_streams)


def unit(ctx):
    tabs = []
    for t, (w, lg) in enumerate(zip(air.ZKEVM_TABLE_WIDTHS, air.ZKEVM_CFG4_LOG_SIZES)):
        prog = air.cpu_like_air(w, min(19, w // 4))
        d, pub = zl.synth_trace_dev(ctx, w, lg, 0x5EED0000 + 1000 * t)
        tabs.append(ops.TableInput(prog, lg, d, pub))
    return ops.GenerationInputs(tabs)


ctx = st.ctx()
u0 = unit(ctx)
chains = [[b for _, _, b in rec.shrink_chain(t, tb.air, tb.log_n)] for t, tb in enumerate(u0.tables)]
print("shrink chains (degree bits per circuit, initial wrapper first):", chains, "root", rec._bits(rec.ROOT_BITS), flush=True)
if args.jit:
    st.prepare(u0.tables)
    print("constraint kernels prepared: %s" % (zl.prover_state_stats(),), flush=True)
timing = {}
root, pv, allp = ops.prove_root(u0, timing)        # first call: builds every circuit's witness and prover data (start-up work)
print("start-up (circuits, witnesses, commitments) + first unit: %.1f s" % (time.time() - t0), flush=True)
tt = {"stark_s": 0.0, "shrink_s": 0.0, "root_s": 0.0}
t1 = time.time()
for _ in range(args.reps):
    timing = {}
    r2, _, _ = ops.prove_root(u0, timing)
    assert np.array_equal(r2, root)
    for k in tt:
        tt[k] += timing[k]
dt = (time.time() - t1) / args.reps
n_plonk = sum(len(c) for c in chains) + 1
print("one unit alone: %.1f ms = STARK %.1f + %d shrink-chain proofs %.1f + root %.1f   (root proof %d words, STARK proofs %d words)"
      % (dt * 1e3, tt["stark_s"] / args.reps * 1e3, n_plonk - 1, tt["shrink_s"] / args.reps * 1e3, tt["root_s"] / args.reps * 1e3, root.size, allp.size), flush=True)
ctx.profile(True); ctx.profile_reset()
ops.prove_root(u0)
rep = ctx.profile_report()
ctx.profile(False)
print("kernel table of one unit (hipEvent profiler on):")
for k, (l, ms) in sorted(rep.items(), key=lambda kv: -kv[1][1])[:args.top]:
    print("  %-30s %6d launches %9.3f ms" % (k, l, ms))
if args.inflight > 1:
    errs = []
    go, done = threading.Barrier(args.inflight + 1), threading.Barrier(args.inflight + 1)

    def worker(s):              # a worker's context is thread-local: create it, make its unit, prove once untimed, then the timed units
        try:
            u = unit(st.ctx())
            ops.prove_root(u)
            go.wait()
            for _ in range(args.reps):
                ops.prove_root(u)
            done.wait()
        except Exception as e:
            errs.append(e)
            go.abort(); done.abort()
    th = [threading.Thread(target=worker, args=(s,)) for s in range(args.inflight)]
    for t in th: t.start()
    go.wait()
    t2 = time.time()
    done.wait()
    dt = time.time() - t2
    for t in th: t.join()
    if errs: raise errs[0]
    n = args.reps * args.inflight
    print("%d units in flight: %.1f ms per transaction (%.2f tx/s; %d units in %.2f s)" % (args.inflight, dt / n * 1e3, n / dt, n, dt))
