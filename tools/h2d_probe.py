#!/usr/bin/env python3
"""What the host link gives: one pinned 2^20 x 93 trace (780 MB) uploaded with zk_trace_upload_async, alone on an idle GPU, as one
copy and cut into 2 / 4 parts on side streams (ZK_UPLOAD_PARTS).  usage: python tools/h2d_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import eth_tx_proof_amd  # noqa: F401
from eth_tx_proof_amd import lib as zl
ctx = zl.Context(0)
C, n = 93, 1 << 20
h = zl.pinned_empty(ctx, (C, n))
h[...] = np.arange(C * n, dtype=np.uint64).reshape(C, n)
for name, env in (("dma 1 part", {"ZK_UPLOAD_PARTS": "1"}), ("dma 2 parts", {"ZK_UPLOAD_PARTS": "2"}), ("dma 4 parts", {"ZK_UPLOAD_PARTS": "4"})):
    os.environ.update(env)
    best = 0
    for rep in range(4):
        ctx.synchronize()
        t0 = time.perf_counter()
        d = zl.trace_upload_async(ctx, h)
        ctx.dev_free(d)
        ctx.synchronize()
        dt = time.perf_counter() - t0
        best = max(best, 8 * C * n / dt / 1e9)
    print("%-12s %.1f GB/s" % (name, best), flush=True)
# split and unsplit uploads deliver the same bytes
os.environ["ZK_UPLOAD_PARTS"] = "4"
d = zl.trace_upload_async(ctx, h)
b1 = ctx.batch_from_values_dev(d, 20, C)
os.environ["ZK_UPLOAD_PARTS"] = "1"
d2 = zl.trace_upload_async(ctx, h)
b2 = ctx.batch_from_values_dev(d2, 20, C)
print("same commitment:", bool((b1.cap == b2.cap).all()))
