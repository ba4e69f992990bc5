#!/bin/bash
# How busy is the GPU while eight transactions are in flight?  rocprofv3 --kernel-trace over bench.py --workload prove_root; over the densest
# 1.5 s of the trace (the timed region): share of wall time with at least one kernel running, time-weighted number of kernels running at once,
# and the same counting only kernels that can fill the chip (>= 512 workgroups x 256 threads or >= 1024 waves).
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/trace8
rm -rf $OUT && mkdir -p $OUT
cd $R
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -o c -- python3 bench.py --workload prove_root --chain-overlap 0 --inflight 8 --steps 8 > $OUT/bench.json 2> $OUT/bench.err || { tail -3 $OUT/bench.err; exit 1; }
python3 - <<'PY' | tee $OUT/trace_concurrency.txt
import csv, glob, json, os
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/trace8"
d = json.load(open(root + "/bench.json")); print("tx/s", round(d["value"], 2), "timed region ms", round(d["ms_per_step"] * d["steps"], 1))
f = glob.glob(root + "/t/**/c_kernel_trace.csv", recursive=True)[0]
ev = []
for r in csv.DictReader(open(f)):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    wg = int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1); gx = int(r.get("Grid_Size_X", r.get("Grid_Size", 1)) or 1)
    ev.append((s, e, gx // 64, r["Kernel_Name"].split("(")[0][:40]))
ev.sort()
t_end = max(e for _, e, _, _ in ev)
win = d["ms_per_step"] * d["steps"] * 1e6
lo, hi = t_end - win, t_end              # the timed region is the end of the run
sel = [(max(s, lo), min(e, hi), w, k) for s, e, w, k in ev if e > lo and s < hi]
pts = []
for s, e, w, k in sel:
    pts.append((s, 1, w)); pts.append((e, -1, -w))
pts.sort()
busy = 0; conc_area = 0; waves_area = 0; cur = 0; curw = 0; last = lo
hist = {}
for t, dc, dw in pts:
    dt = t - last
    if cur > 0: busy += dt
    conc_area += cur * dt; waves_area += curw * dt
    b = min(cur, 12); hist[b] = hist.get(b, 0) + dt
    cur += dc; curw += dw; last = t
print("kernels in the window: %d; wall %.1f ms; some kernel running %.1f %% of it; kernels running at once (time-weighted) %.2f; waves launched and not yet finished (time-weighted) %.0f of 8192 slots"
      % (len(sel), win / 1e6, 100 * busy / win, conc_area / win, waves_area / win))
print("share of wall time by number of kernels running:", {k: round(100 * v / win, 1) for k, v in sorted(hist.items())})
tot = {}
for s, e, w, k in sel:
    a = tot.setdefault(k, [0, 0]); a[0] += 1; a[1] += e - s
print("kernel time in the window, top 12:")
for k, (n, t) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:12]:
    print("  %-42s %6d launches %8.1f ms  avg %7.1f us" % (k, n, t / 1e6, t / n / 1e3))
PY
rm -rf $OUT/t
