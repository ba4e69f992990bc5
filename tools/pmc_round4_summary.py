#!/usr/bin/env python3
"""Summary of tools/pmc_round4.sh: per kernel name and counter the number of dispatches and the mean per dispatch; from the pass that carries
GRBM_GUI_ACTIVE also the mean duration and the clock the chip ran the kernel at (GRBM_GUI_ACTIVE / 8 XCDs / duration, MI355X_MICROARCH.md
"DVFS give-back"); from the plain kernel trace the overlap of the timed four-stream region (sum of kernel durations / wall-clock span of the
densest window).  usage: tools/pmc_round4_summary.py gpurun_out/pmc4 > summary.json"""
import csv, glob, json, os, sys
root = sys.argv[1]
out = {"counters": {}, "derived": {}}


def short(name):
    n = name.split("(")[0]
    return n.replace("void ", "")[:90]


for d in sorted(glob.glob(os.path.join(root, "*/"))):
    tag = os.path.basename(d.rstrip("/"))
    cc = glob.glob(d + "**/c_counter_collection.csv", recursive=True)
    kt = glob.glob(d + "**/c_kernel_trace.csv", recursive=True)
    dur = {}
    if kt:
        for row in csv.DictReader(open(kt[0])):
            try:
                dur[row.get("Dispatch_Id") or row.get("Correlation_Id")] = (short(row["Kernel_Name"]), int(row["Start_Timestamp"]), int(row["End_Timestamp"]))
            except Exception:
                pass
    if cc:
        acc = {}
        gui = {}
        for row in csv.DictReader(open(cc[0])):
            k = (row["Counter_Name"], short(row["Kernel_Name"]))
            a = acc.setdefault(k, [0, 0.0])
            v = float(row["Counter_Value"])
            a[0] += 1; a[1] += v
            if row["Counter_Name"] == "GRBM_GUI_ACTIVE":
                did = row.get("Dispatch_Id")
                if did in dur:
                    g = gui.setdefault(short(row["Kernel_Name"]), [0, 0.0, 0.0])
                    g[0] += 1; g[1] += v; g[2] += dur[did][2] - dur[did][1]
        for (c, kern), (n, tot) in acc.items():
            out["counters"].setdefault(tag, {}).setdefault(c, {})[kern] = {"dispatches": n, "mean": tot / n}
        for kern, (n, g, ns) in gui.items():
            if ns > 0:
                out["derived"].setdefault("effective_clock_GHz", {})[kern] = {"dispatches": n, "mean_duration_us": ns / n / 1e3, "GHz": g / 8.0 / ns}
    elif dur and tag == "bench_trace":
        ev = sorted((s, e, k) for k, s, e in dur.values())
        if ev:
            # the four-stream timed region = the window of the trace with the most kernel time per wall time over 0.3 s: report the whole trace
            # and the best 0.3 s window
            t0, t1 = ev[0][0], max(e for _, e, _ in ev)
            busy = sum(e - s for s, e, _ in ev)
            out["derived"]["kernel_trace_whole_run"] = {"kernels": len(ev), "sum_of_durations_ms": busy / 1e6, "wall_ms": (t1 - t0) / 1e6}
            win = 300e6
            best = (0.0, 0)
            j = 0
            acc_busy = 0.0
            starts = [s for s, _, _ in ev]
            import bisect
            for i in range(0, len(ev), 50):
                lo = ev[i][0]
                hi = lo + win
                jj = bisect.bisect_left(starts, hi)
                b = sum(min(e, hi) - s for s, e, _ in ev[i:jj])
                if b > best[0]:
                    best = (b, lo)
            out["derived"]["densest_300ms_window"] = {"sum_of_durations_ms": best[0] / 1e6, "overlap_factor": best[0] / win}
json.dump(out, sys.stdout, indent=1, sort_keys=True)
