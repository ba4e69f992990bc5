#!/usr/bin/env python3
"""What one card holds: proves units of the given shapes, one after the other on one context, and reports the device
memory in use right after each proof -- the context caches every block it used, so that is the proof's peak reservation -- then trims.
A shape that does not fit must fail with ZK_ERR_OOM and leave the context usable.
usage: tools/capacity_probe.py log_n:cols:helpers [...]   (e.g. 24:93:19 26:21:5 27:21:5 25:93:19)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import eth_tx_proof_amd as E
from eth_tx_proof_amd import air, lib as zl
import orc

cfg = E.standard_fast_config()
ctx = zl.Context(0)
used = lambda: (torch.cuda.mem_get_info()[1] - torch.cuda.mem_get_info()[0]) / 1e9
base = used()
for spec in sys.argv[1:]:
    log_n, cols, helpers = [int(v) for v in spec.split(":")]
    prog = air.cpu_like_air(cols, helpers)
    t0 = time.time()
    try:
        d, pub = zl.synth_trace_dev(ctx, cols, log_n, 0x7A110000 + 1000 * log_n + cols)
    except Exception as e:
        print("2^%d x %d+%d: trace allocation failed: %s" % (log_n, cols, prog.n_aux_cols, str(e)[:120]), flush=True)
        continue
    try:
        t1 = time.time()
        proof = zl.prove_tables(ctx, cfg, [(prog, log_n, d, pub)])
        dt = time.time() - t1
        peak = used() - base
        ok = orc.verify(cfg, [(prog, log_n, None, pub)], proof) == 0
        print("2^%d x %d+%d: proved in %.2f s, %.1f GB reserved at the end of the proof (trace %.1f GB), oracle's verifier accepts: %s"
              % (log_n, cols, prog.n_aux_cols, dt, peak, 8 * cols * 2.0 ** log_n / 1e9, ok), flush=True)
    except Exception as e:
        print("2^%d x %d+%d: %s (%.1f GB reserved when it gave up)" % (log_n, cols, prog.n_aux_cols, str(e)[:160], used() - base), flush=True)
    finally:
        ctx.dev_free(d)
        print("   trim: %.1f GB back to the driver, %.1f GB still in use" % (ctx.trim() / 1e9, used() - base), flush=True)
ctx.close()
