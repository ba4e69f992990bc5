#!/usr/bin/env python3
"""From the PMC passes of tools/profile_round.sh (one 93-column 2^20-row commit under rocprofv3 --pmc, one counter group per pass):
the HBM traffic and the instruction mix of the dominant kernel, in the file bench.py reads for `roofline.traffic` / `roofline.alu`.
usage: python tools/pmc_leaf_hash.py gpurun_out/prof > profiles/rNN_pmc_leaf_hash.json"""
import csv, glob, json, os, sys
root = sys.argv[1]
KERN = "hash_leaves_colmajor_mfma_kernel"
LOG_N, COLS = 20, 93
L = 2 << LOG_N
vals = {}
for d in sorted(glob.glob(os.path.join(root, "pmc_*"))):
    f = os.path.join(d, "c_counter_collection.csv")
    if not os.path.isfile(f):
        continue
    acc = {}
    for row in csv.DictReader(open(f)):
        if not row["Kernel_Name"].startswith(KERN):
            continue
        a = acc.setdefault(row["Counter_Name"], [0, 0.0])
        a[0] += 1; a[1] += float(row["Counter_Value"])
    for c, (n, tot) in acc.items():
        vals[c] = tot / n                      # mean per launch (warm-up and timed launch are the same shape)
perms = L * ((COLS + 7) // 8)
waves = perms // 64
alg = 8 * L * COLS + 32 * L
out = {"command": "tools/profile_round.sh: rocprofv3 --kernel-trace --pmc <one counter group per pass> -- python3 tools/commit_bench.py 20 93 1 "
                  "(warm-up + one timed 93-column 2^20-row commit: the leaf-hash launch covers 2^21 leaves x 12 permutations)",
       "note": "gfx950: FETCH_SIZE tallies the 128-B requests of a coalesced streaming read at 64 B -> doubled (MI355X_MICROARCH.md, HBM section); WRITE_SIZE as is; both in KB",
       "leaf_hash_93cols": {"kernel": KERN, "algorithmic_bytes": alg, "permutations": perms, "waves_of_64_permutations": waves}}
r = out["leaf_hash_93cols"]
if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
    r["FETCH_SIZE_KB"], r["WRITE_SIZE_KB"] = vals["FETCH_SIZE"], vals["WRITE_SIZE"]
    r["pmc_bytes"] = (2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0
if "SQ_INSTS_VALU" in vals:
    r["valu_instr_per_wave_permutation"] = vals["SQ_INSTS_VALU"] / waves
if "SQ_INSTS_SALU" in vals:
    r["salu_instr_per_wave_permutation"] = vals["SQ_INSTS_SALU"] / waves
if "SQ_INSTS_VALU_MFMA_I8" in vals:
    r["mfma_i8_instr_per_wave_permutation"] = vals["SQ_INSTS_VALU_MFMA_I8"] / waves
for c in ("SQ_WAVE_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY", "SQ_BUSY_CYCLES"):
    if c in vals:
        r[c] = vals[c]
missing = [k for k in ("pmc_bytes", "valu_instr_per_wave_permutation", "mfma_i8_instr_per_wave_permutation") if k not in r]
if missing:
    sys.exit("pmc_leaf_hash.py: passes missing for %s under %s" % (missing, root))
json.dump(out, sys.stdout, indent=1)
