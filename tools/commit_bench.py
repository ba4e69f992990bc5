#!/usr/bin/env python3
"""Micro-bench of one PolynomialBatch::from_values (BASELINE configs[1]: 2^20 NTT + Poseidon Merkle cap) with the ctx profiler.
usage: python tools/commit_bench.py [log_n] [n_cols] [reps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import eth_tx_proof_amd
from eth_tx_proof_amd import lib as zl
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
C = int(sys.argv[2]) if len(sys.argv) > 2 else 93
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
ctx = zl.Context(0)
d, _ = zl.synth_trace_dev(ctx, C, log_n, 1)
b = ctx.batch_from_values_dev(d, log_n, C); b.free()
ctx.profile(True); ctx.profile_reset()
t0 = time.time()
for _ in range(reps):
    b = ctx.batch_from_values_dev(d, log_n, C); b.free()
ctx.synchronize()
dt = (time.time() - t0) / reps
rep = ctx.profile_report()
n = 1 << log_n
print("commit %.2f ms  (tab=%s colfast=%s)" % (dt * 1e3, os.environ.get("ZK_NTT_TAB"), os.environ.get("ZK_NTT_COLFAST")))
for k, (l, ms) in sorted(rep.items(), key=lambda kv: -kv[1][1]):
    extra = ""
    if k.startswith("ntt_"):
        gb = (l / reps) * 0  # per-launch bytes differ; report ms only
    print("  %-28s %4d launches %8.3f ms/commit" % (k, l // reps, ms / reps))
ntt = sum(ms for k, (l, ms) in rep.items() if k.startswith("ntt_d")) / reps
print("  NTT total %.3f ms -> algorithmic %.0f GB/s (16N + 8N + 16N... = %d MB)" % (ntt, (16 * n * C + 24 * n * C) / ntt / 1e6, (40 * n * C) >> 20))
