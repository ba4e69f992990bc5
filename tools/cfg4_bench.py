#!/usr/bin/env python3
"""BASELINE configs[3]: seven-table unit (log sizes 15/9/12/14/9/12/17, recalled zkEVM widths) on one MI355X; ms per unit."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import eth_tx_proof_amd as E
from eth_tx_proof_amd import air, lib as zl
cfg = E.standard_fast_config()
ctx = zl.Context(0)
items = []
for t, (w, lg) in enumerate(zip(air.ZKEVM_TABLE_WIDTHS, air.ZKEVM_CFG4_LOG_SIZES)):
    prog = air.cpu_like_air(w, min(19, w // 4))
    d, pub = zl.synth_trace_dev(ctx, w, lg, 0x5EED0000 + 1000 * t)
    items.append((prog, lg, d, pub))
zl.prove_tables(ctx, cfg, items)
ctx.profile(True); ctx.profile_reset()
t0 = time.time(); reps = 5
for _ in range(reps):
    p = zl.prove_tables(ctx, cfg, items)
dt = (time.time() - t0) / reps
print("cfg4 unit: %.2f ms, proof %d words" % (dt * 1e3, p.size))
for k, (l, ms) in sorted(ctx.profile_report().items(), key=lambda kv: -kv[1][1])[:10]:
    print("  %-28s %5d launches %8.3f ms/unit" % (k, l // reps, ms / reps))
