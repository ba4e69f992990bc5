#!/usr/bin/env python3
"""Two/four contexts committing concurrently (threads): caps must equal the single-context result.  usage: conc_probe.py [log_n] [cols] [threads]"""
import os, sys, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import eth_tx_proof_amd
from eth_tx_proof_amd import lib as zl
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
C = int(sys.argv[2]) if len(sys.argv) > 2 else 21
T = int(sys.argv[3]) if len(sys.argv) > 3 else 4
ctxs = [zl.Context(0) for _ in range(T)]
data = [zl.synth_trace_dev(c, C, log_n, 7)[0] for c in ctxs]
b = ctxs[0].batch_from_values_dev(data[0], log_n, C); ref = b.cap.copy(); b.free()
print("reference cap word0 %016x" % int(ref.ravel()[0]), flush=True)
bad = []
def work(i):
    for r in range(6):
        b = ctxs[i].batch_from_values_dev(data[i], log_n, C)
        if not (b.cap == ref).all(): bad.append((i, r))
        b.free()
th = [threading.Thread(target=work, args=(i,)) for i in range(T)]
for t in th: t.start()
for t in th: t.join()
print("concurrent commits: %s" % ("OK" if not bad else "MISMATCH %s" % bad[:8]), flush=True)
