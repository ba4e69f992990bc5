#!/usr/bin/env python3
"""bench.py -- unit STARK proofs/s on MI355X for BASELINE.json's metric.

Workload (config.workload): BASELINE.json configs[2], the configuration the metric is quoted on
("segment STARK proofs/sec (2^20 CPU_CIRCUIT rows)"): one full unit proof = STARK + FRI with
StarkConfig::standard_fast_config() over a synthetic, satisfiable, degree-3, Cpu-shaped table of 2^20 rows x 93 trace
columns (+40 logUp/CTL-shaped auxiliary columns).  A "step" = one batch of `--inflight` independent unit proofs on
this rank (one HIP stream + one host thread each, like paladin `--num-workers`).  Traces are generated in HBM before
the timed region (witness generation is out of scope).  `--log-n 23` runs north_star's target
configuration (one 2^23-row unit) the same way.

Multi-GPU: units are independent (reference: IndexedStream.map(&TxProof), leader/src/prover.rs:26-27), so ranks shard
units with no data-path collective ("scaling": "weak"); the only exchange is the hand-off of finished proofs to rank 0
(host buffers over a gloo group, each proof sent from a background thread while the next unit is proved: sharding.Handoff), inside the
timed region; RCCL carries the barrier and the max-over-ranks timing.  `--gpus N` with no WORLD_SIZE in the environment starts
the N ranks itself (child processes, before anything in this process touches the GPU); under torch.distributed.run the
environment's WORLD_SIZE must equal --gpus.

Timing: `value` comes from EXACTLY --steps steps with the per-kernel hipEvent profiler OFF; the same number of steps is
then repeated with the profiler ON (`value_profiled`), and the kernel table and `roofline` come from that second region.

Prints ONE JSON line on rank 0.
"""
import argparse
import glob
import json
import os
import re
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--log-n", type=int, default=20)
    ap.add_argument("--cols", type=int, default=93)
    ap.add_argument("--helpers", type=int, default=19)
    ap.add_argument("--inflight", type=int, default=0,
                    help="independent unit proofs in flight per GPU, one stream each (default 4; 2 at 2^22 rows and above)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pcie", action="store_true", help="skip the PCIe-inclusive (host-resident traces) measurement")
    ap.add_argument("--no-live-pmc", action="store_true", help="roofline.traffic from the newest committed PMC pass instead of a live rocprofv3 --pmc child run")
    ap.add_argument("--no-profile", action="store_true", help="skip the profiled repeat (no kernel table / roofline)")
    ap.add_argument("--chain-overlap", type=int, default=-1, help="--workload prove_root: 1 = a table's shrink chain starts under the STARK proofs of the later tables, 0 = after all of "
                    "them, -1 (default) = the prover state's own rule: overlap while at most two transactions are in flight")
    ap.add_argument("--chain-streams", type=int, default=3, help="--workload prove_root: shrink chains of ONE transaction proved side by side by its worker")
    ap.add_argument("--prepare", choices=["auto", "always", "never"], default="auto", help="prove_root: constraint kernels of the wide tables -- see run_prove_root")
    ap.add_argument("--no-log23", action="store_true", help="default workload: skip the 2^23-row record (\"log23\" in the line)")
    ap.add_argument("--no-prove-root", action="store_true", help="default workload: skip the short transaction record (\"prove_root\" in the line)")
    ap.add_argument("--workload", choices=["stark_unit", "prove_root"], default="stark_unit",
                    help="stark_unit (default, BASELINE's metric): one unit STARK proof per step and worker.  prove_root: one TRANSACTION end to end "
                         "per step and worker -- BASELINE configs[3]'s seven tables, their shrink chains and the root proof (TxProof.execute)")
    return ap.parse_args()


CPU_BASELINE_CACHES = [os.path.join(ROOT, "gpurun_out", ".cpu_baseline_cache.json"), "/tmp/zkgl_cpu_baseline_cache.json"]     # written by an N = 1 run, read by N > 1 runs
PROVER_STATE_DIR = os.path.join(ROOT, "eth-tx-proof_amd", "prover_state")      # compiled constraint kernels of the wide tables (git-ignored *.co, travels with the tree)


def golden_prove_root():
    try:
        ents = json.load(open(os.path.join(ROOT, "tests", "golden", "proof_digests_big.json"))).get("prove_root", [])
    except Exception:
        return None
    return ents[0] if ents else None


def run_prove_root(inflight, steps, warmup, chain_streams=3, chain_overlap=-1, prepare="auto"):
    """One TRANSACTION per step and worker: what one call of the reference's operator does (/root/reference/ops/src/lib.rs:52,
    generate_txn_proof = prove_root): the STARK proofs of the seven tables at configs[3]'s sizes, every table's initial wrapper and
    shrinking wrappers, the root proof -- all on the GPU, recursion circuits = plonk.SyntheticRecursion's stand-ins at the estimated sizes
    of the real ones.  Returns the record; the root proof and the AllProof of the first transaction are compared with the ORACLE's
    committed digests (tests/golden/proof_digests_big.json, tools/gen_golden_big.py --root).
    prepare: "always" = zk_prover_state_prepare (hiprtc compiles the constraint kernels of the 430 / 523 / 2 431-column tables, minutes, unless
    the prover-state directory holds them); "never" = those three tables on the bytecode interpreter; "auto" = prepare when the
    prover-state directory already holds compiled kernels (no compile inside a bench run)."""
    import glob as _glob
    import hashlib
    import numpy as np
    import torch
    import eth_tx_proof_amd as E  # noqa: F401
    from eth_tx_proof_amd import air, lib as zl, ops, plonk
    rec = plonk.SyntheticRecursion()
    old = ops.P_STATE
    ops.P_STATE = None
    st = ops.set_prover_state_from_config(0, recursion=rec, chain_streams=chain_streams, chain_overlap={0: False, 1: True}.get(chain_overlap, "auto"))
    try:
        def unit(ctx):
            tabs = []
            for t, (w, lg) in enumerate(zip(air.ZKEVM_TABLE_WIDTHS, air.ZKEVM_CFG4_LOG_SIZES)):
                d, pub = zl.synth_trace_dev(ctx, w, lg, 0x5EED0000 + 1000 * t)
                tabs.append(ops.TableInput(air.cpu_like_air(w, min(19, w // 4)), lg, d, pub))
            return ops.GenerationInputs(tabs)
        ctx0 = st.ctx()
        u0 = unit(ctx0)
        chains = [[b for _, _, b in rec.shrink_chain(t, tb.air, tb.log_n)] for t, tb in enumerate(u0.tables)]
        have_state = len(_glob.glob(os.path.join(PROVER_STATE_DIR, "prover_state_*.gfx950.co"))) >= 3
        do_prepare = prepare == "always" or (prepare == "auto" and have_state)
        c0, l0 = zl.prover_state_stats()
        t_prep = time.perf_counter()
        if do_prepare:
            os.makedirs(PROVER_STATE_DIR, exist_ok=True)
            zl.prover_state_cache_dir(PROVER_STATE_DIR)
            st.prepare(u0.tables)                  # constraint kernels of the wide tables (hiprtc or the prover-state directory): start-up work
        t_prep = time.perf_counter() - t_prep
        c1, l1 = zl.prover_state_stats()
        root0, _, all0 = ops.prove_root(u0)        # builds every circuit's prover data: start-up work too
        split = {"stark_s": 0.0, "shrink_s": 0.0, "root_s": 0.0}
        reps = 3
        ws0 = (rec.witness_standin_seconds, rec.witness_standin_rows)
        for _ in range(reps):
            tm = {}
            r, _, _ = ops.prove_root(u0, tm)
            assert np.array_equal(r, root0)
            for k in split:
                split[k] += tm[k] / reps
        ws_ms_per_tx, ws_rows_per_tx = (rec.witness_standin_seconds - ws0[0]) * 1e3 / reps, (rec.witness_standin_rows - ws0[1]) // reps
        # the same without the CPU stand-in for per-proof witness generation (round 4's number): plonk.SyntheticRecursion.witness_standin
        standin_default = rec.witness_standin
        alone_without = None
        if standin_default:
            rec.witness_standin = False
            alone_without = 0.0
            for _ in range(reps):
                tm = {}
                r, _, _ = ops.prove_root(u0, tm)
                assert np.array_equal(r, root0)
                alone_without += sum(tm[k] for k in split) / reps
            rec.witness_standin = True
        ctx0.profile(True); ctx0.profile_reset()
        ops.prove_root(u0)
        rep = ctx0.profile_report()
        ctx0.profile(False)
        def timed_region():
            """`inflight` worker threads, one context and one transaction each: warm-up, then `steps` transactions per worker between two barriers"""
            errs = []
            go, done = threading.Barrier(inflight + 1), threading.Barrier(inflight + 1)

            def worker(s):
                try:
                    u = unit(st.ctx())
                    for _ in range(max(1, warmup)):
                        ops.prove_root(u)
                    go.wait()
                    for _ in range(steps):
                        r = ops.prove_root(u)[0]
                    if not np.array_equal(r, root0):
                        raise RuntimeError("worker %d: another root proof than the first unit's" % s)
                    done.wait()
                except threading.BrokenBarrierError:
                    pass                               # another worker failed first
                except Exception as e:
                    import traceback
                    errs.append(e)
                    sys.stderr.write("[bench] transaction worker %d failed: %r\n%s" % (s, e, traceback.format_exc()))
                    sys.stderr.flush()
                    go.abort(); done.abort()
                finally:
                    st.drop_ctx()
            th = [threading.Thread(target=worker, args=(s,)) for s in range(inflight)]
            for t in th:
                t.start()
            try:
                go.wait()
            except threading.BrokenBarrierError:
                pass                                   # a worker failed during its warm-up: its error is raised below
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if os.environ.get("ZK_VA_PROBE"):          # debug: the runtime's view of the address windows the unexplained GPU faults named, with every context alive and busy
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import va_probe
                time.sleep(0.2)
                va_probe.probe("transactions in flight")
            try:
                done.wait()
            except threading.BrokenBarrierError:
                pass                                   # a worker failed: what it raised is in errs
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            for t in th:
                t.join()
            if errs:
                raise errs[0]
            return dt
        # Two timed regions each way, alternating (a region's tx/s scatters by about +-5 % on one box): `value` = with the CPU stand-in for
        # per-proof witness generation (the default), `value_without_witness_generation` = round 4's form
        dts, dts_without = [timed_region()], []
        if standin_default:
            for _ in range(2):
                rec.witness_standin = False
                dts_without.append(timed_region())
                rec.witness_standin = True
                if len(dts) < 2:
                    dts.append(timed_region())
        dt = sum(dts) / len(dts)
        dt_without = (sum(dts_without) / len(dts_without)) if dts_without else None
        n_tx = steps * inflight
        sha = lambda w: hashlib.sha256(np.ascontiguousarray(w, dtype=np.uint64).tobytes()).hexdigest()
        gold = golden_prove_root()
        check = {"root_sha256": sha(root0), "all_proof_sha256": sha(all0), "oracle_root_sha256": gold["root"]["sha256"] if gold else None,
                 "oracle_all_proof_sha256": gold["all_proof"]["sha256"] if gold else None,
                 "identical_to_oracle": (sha(root0) == gold["root"]["sha256"] and sha(all0) == gold["all_proof"]["sha256"]) if gold else None,
                 "note": "first transaction of the run; every timed transaction's root proof equals it word for word (checked per worker)"}
        kern = {k: {"launches_per_tx": l, "ms_per_tx_alone": round(ms, 3)} for k, (l, ms) in sorted(rep.items(), key=lambda kv: -kv[1][1])[:14] if not k.startswith("phase: ")}
        return {
            "metric": "transaction root proofs/sec (prove_root: 7 table STARKs + shrink chains + root proof)", "value": n_tx / dt, "unit": "tx/s", "n_gpus": 1,
            "value_without_witness_generation": (n_tx / dt_without) if dt_without else None,
            "timed_regions_tx_per_s": {"with_witness_generation": [round(n_tx / t, 2) for t in dts], "without": [round(n_tx / t, 2) for t in dts_without]},
            "steps": steps, "warmup": warmup, "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64 (Goldilocks field)", "data": "synthetic",
            "config": {"workload": "BASELINE configs[3] end to end: seven tables (log sizes %s, widths %s) on one challenger, shrink chains %s (degree bits), "
                                   "root 2^%d; recursion circuits: stand-ins of the recursion shape (plonk.SyntheticRecursion), standard_recursion_config"
                                   % (list(air.ZKEVM_CFG4_LOG_SIZES), list(air.ZKEVM_TABLE_WIDTHS), chains, rec._bits(rec.ROOT_BITS)),
                       "tx_per_step_per_gpu": inflight, "shrink_chains_side_by_side_per_tx": chain_streams,
                       "shrink_chains_start_under_the_later_tables_stark_proofs": {0: "never", 1: "always"}.get(chain_overlap, "auto: with at most 2 transactions in flight (so: for one_tx_alone_ms, not in the timed region unless --inflight <= 2)"),
                       "constraint_kernels_of_the_wide_tables": ("compiled (zk_prover_state_prepare: %d by hiprtc in %.1f s, %d from the prover-state directory)" % (c1 - c0, t_prep, l1 - l0)) if do_prepare
                       else "bytecode interpreter (no prover-state files in %s and no compile inside a bench run: python bench.py --workload prove_root --prepare always writes them)" % os.path.relpath(PROVER_STATE_DIR, ROOT),
                       "fiat_shamir_sponge": "device-resident (one-wave kernel) in the timed region; on a host thread (zk_ctx_set_host_transcript) while at most 2 transactions are in flight, "
                                             "i.e. for one_tx_alone_ms -- the same rule as the chain overlap; identical proof words either way",
                       "recursion_witnesses": ("CPU STAND-IN for per-proof witness generation inside every transaction (zk_host_plonk_witness_recompute: the cells of plonky2's PoseidonGenerator / "
                                               "ArithmeticBaseGenerator recomputed from the rows' inputs in the proving threads: %d gate rows = %.1f ms of CPU per transaction); `value_without_witness_generation` / "
                                               "`one_tx_alone_ms.total_without_witness_generation` = round 4's form (one precomputed witness per circuit).  The real circuits' generators are absent (SURVEY A10): a "
                                               "cost model, not upstream's code" % (ws_rows_per_tx, ws_ms_per_tx))
                                              if standin_default else "PRECOMPUTED: one satisfying witness per stand-in circuit (ZK_WITNESS_STANDIN=0); upstream's per-proof witness generators are NOT in this number",
                       "parallelism": "independent transactions, 1 GPU"},
            "one_tx_alone_ms": {"total": sum(split.values()) * 1e3, "total_without_witness_generation": alone_without * 1e3 if alone_without is not None else None, "stark_proofs": split["stark_s"] * 1e3,
                                "shrink_chains_%d_plonk_proofs" % sum(len(c) for c in chains): split["shrink_s"] * 1e3, "root_proof": split["root_s"] * 1e3},
            "timed_proof_check": check, "kernels_one_tx_alone": kern}
    finally:
        if ops.P_STATE is not None:
            ops.P_STATE.drop_ctx()
        ops.P_STATE = old


def log23_record(args, timeout_s=240):
    """north_star's target shape in the driver-run line: 2^23 rows x 93 (+40) columns, 2 units in flight x 4 steps, measured
    by a fresh CHILD process like the transaction record -- digest check against tests/golden/proof_digests_big.json, NTT and roofline
    figures for that size.  The headline `value` stays BASELINE's 2^20 metric."""
    cmd = [sys.executable, os.path.abspath(__file__), "--log-n", "23", "--inflight", "2", "--steps", "4", "--warmup", "1",
           "--no-cpu-baseline", "--no-pcie", "--no-live-pmc", "--no-prove-root"]
    pr, err = child_json_line(cmd, timeout_s)
    if err is not None:
        return err
    rec = {k: pr.get(k) for k in ("metric", "value", "unit", "steps", "ms_per_step", "ntt", "timed_proof_check")}
    rf = pr.get("roofline") or {}
    rec["roofline"] = {k: rf.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "avg_launch_ms", "algorithmic_bytes_per_launch")}
    rec["config"] = pr.get("config")
    rec["identical_to_oracle"] = (pr.get("timed_proof_check") or {}).get("identical_to_oracle")
    rec["note"] = "NOT the headline: north_star's 2^23-row unit, measured by a child process after the headline's own work (python bench.py --log-n 23 gives the long form)"
    return rec


def child_json_line(cmd, timeout_s):
    """runs a bench child process; returns (its last JSON line, None) or (None, an error record).  The child's stderr is passed on."""
    import tempfile
    with tempfile.TemporaryFile() as err:
        t0 = time.perf_counter()
        proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=err, env=dict(os.environ, ZK_BENCH_CHILD="1"))
        try:
            out, _ = proc.communicate(timeout=timeout_s)
            timed_out = False
        except subprocess.TimeoutExpired:
            proc.kill()
            try:
                out, _ = proc.communicate(timeout=15)
            except subprocess.TimeoutExpired:
                out = b""
            timed_out = True
        err.seek(0)
        etxt = err.read().decode(errors="replace")
    if etxt:
        sys.stderr.write(etxt if len(etxt) < 200000 else etxt[:100000] + "\n[...]\n" + etxt[-100000:])
        sys.stderr.flush()
    fault = "Memory access fault" in etxt or "[zk-fault]" in etxt
    if timed_out or proc.returncode != 0:
        return None, {"error": ("timed out after %d s" % timeout_s) if timed_out else "child process exited with code %s" % proc.returncode, "gpu_memory_fault": fault,
                      "wall_s": round(time.perf_counter() - t0, 1), "command": " ".join(cmd[1:]), "note": "the child's stderr is in this run's stderr"}
    try:
        return json.loads(out.decode().strip().splitlines()[-1]), None
    except Exception as e:
        return None, {"error": "no JSON line from the child process: %s" % str(e)[:200], "gpu_memory_fault": fault}


def prove_root_record(args, timeout_s=300):
    """The default line's transaction record, measured by a CHILD process (`bench.py --workload prove_root --inflight 8 --steps 4`) after this
    process has finished and released its own GPU work: the headline cannot be lost to anything that happens in the transaction workload
    (rounds 3 and 4 saw one "Memory access fault by GPU" in about twenty runs of it, DESIGN.md section 9, and a faulting process does not
    exit by itself on this stack).  The child's stderr is passed on (the library's flight recorder writes there)."""
    cmd = [sys.executable, os.path.abspath(__file__), "--workload", "prove_root", "--inflight", "8", "--steps", "4", "--warmup", "1", "--prepare", args.prepare]
    pr, err = child_json_line(cmd, timeout_s)
    if err is not None:
        return err
    rec = {k: pr.get(k) for k in ("metric", "value", "value_without_witness_generation", "timed_regions_tx_per_s", "unit", "steps", "ms_per_step", "one_tx_alone_ms", "timed_proof_check")}
    rec["config"] = pr["config"]
    rec["note"] = ("NOT the headline (`value` above is BASELINE's unit metric); measured by a child process after the headline's own work "
                   "(python bench.py --workload prove_root gives the long form)")
    return rec


def bench_prove_root(args):
    """--workload prove_root: the transaction line on its own (never the default: BASELINE's metric is the unit; the default line carries
    a short form of this record under "prove_root").  One GPU; a step = `--inflight` transactions, one worker thread + context each."""
    if int(os.environ.get("WORLD_SIZE", "1")) != 1 or args.gpus != 1:
        sys.exit("bench.py --workload prove_root: one GPU only (transactions shard over ranks exactly like the units of the default workload)")
    line = run_prove_root(args.inflight or 8, args.steps, args.warmup, args.chain_streams, args.chain_overlap, args.prepare)
    line.update({"roofline": None, "cpu_baseline": None,
                 "note": "not BASELINE's metric (that is the default workload); roofline / cpu_baseline are reported by the default workload"})
    print(json.dumps(line))
    if line["timed_proof_check"]["identical_to_oracle"] is False:
        sys.exit("bench.py: the transaction's proofs differ from the oracle's committed digests")


def fan_out(args):
    """--gpus N without a launcher: start the N ranks as children (this process has not touched the GPU and never will),
    relay rank 0's output, return the worst exit code."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # rank 0's output is drained by a thread while ALL children are watched: a rank that dies takes the others down with it (they
    # would otherwise wait in a collective for ever) and the failure is the exit code
    chunks = []
    rd = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    rd.start()
    failed = None
    while failed is None and any(p.poll() is None for p in procs):
        for r, p in enumerate(procs):
            if p.poll() not in (None, 0):
                failed = r
        time.sleep(0.2)
    if failed is not None:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        sys.stderr.write("bench.py: rank %d exited with code %d; the other ranks were stopped\n" % (failed, procs[failed].returncode))
    rcs = [p.wait() for p in procs]
    rd.join(timeout=10)
    sys.stdout.write(b"".join(c for c in chunks if c).decode())
    sys.stdout.flush()
    return max(abs(rc) for rc in rcs)


def newest_profile(pattern):
    """newest committed profile summary matching profiles/<pattern> (rounds are named r01.., r02..: lexicographic order)"""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    if not files:
        raise FileNotFoundError("profiles/%s: no committed profile summary (tools/profile_round.sh writes it)" % pattern)
    return files[-1]


def leaf_label_cols(name):
    m = re.fullmatch(r"poseidon_leaf_hash_c(\d+)", name)
    return int(m.group(1)) if m else None


def algorithmic_bytes_per_launch(kernel, log_n, rate_bits):
    """Compulsory HBM bytes of ONE launch of `kernel`; None if not modelled per launch."""
    L = (1 << log_n) << rate_bits
    c = leaf_label_cols(kernel)
    if c is not None:          # read every LDE value of the commit once, write one digest per leaf
        return 8 * L * c + 32 * L
    return None


def cpu_baseline(log_n_target, n_cols, n_helpers, budget_s=45.0, check_proof=None):
    """The oracle (CPU restatement, kind "port") timed on this host's cores on the same workload.  The thread count is chosen AT a
    size near the sample's (2^17 rows: OpenMP team start-up no longer decides), and the sample is the target size itself whenever one
    proof fits the budget -- then nothing is scaled.  check_proof: (proof words, pub) of a unit proved in the timed region: the
    oracle's verifier must accept it (returned as "timed_proof_verified")."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orc
    import eth_tx_proof_amd as E
    from eth_tx_proof_amd import air
    cfg = E.standard_fast_config()
    prog = air.cpu_like_air(n_cols, n_helpers)
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    verified = None
    if check_proof is not None:
        orc.lib().orc_set_threads(min(avail, 16))
        verified = orc.verify(cfg, [(prog, log_n_target, None, check_proof[1])], check_proof[0]) == 0
    log_c = min(log_n_target, 17)
    cal_tr, cal_pub = orc.synth_trace(n_cols, log_c, 1)
    best = None
    for th in sorted({avail, max(1, avail // 2), max(1, avail // 4), min(avail, 64)}, reverse=True):
        orc.lib().orc_set_threads(th)
        t0 = time.time()
        orc.prove(cfg, [(prog, log_c, cal_tr, cal_pub)])
        dt = time.time() - t0
        if best is None or dt < best[1]:
            best = (th, dt)
    del cal_tr
    cores = best[0]
    orc.lib().orc_set_threads(cores)
    nlogn = lambda lg: float(lg << lg)
    log_s = log_n_target
    while log_s > log_c and best[1] * nlogn(log_s) / nlogn(log_c) > budget_s:      # the largest size whose estimate fits the budget
        log_s -= 1
    tr, pub = orc.synth_trace(n_cols, log_s, 1)
    t0 = time.time()
    rc, proof = orc.prove(cfg, [(prog, log_s, tr, pub)])
    t_s = time.time() - t0
    assert rc == 0
    scale = nlogn(log_n_target) / nlogn(log_s)
    how = "nothing scaled" if log_s == log_n_target else "scaled by N log N (x%.2f) to 2^%d rows" % (scale, log_n_target)
    return {"value": 1.0 / (t_s * scale), "unit": "proofs/s", "cores": cores, "hardware_threads": avail, "kind": "port",
            "sample_log_n": log_s, "sample_seconds": round(t_s, 2), "timed_proof_verified": verified,
            "sample": "oracle/ C restatement (OpenMP, %d of %d hardware threads, thread count chosen at 2^%d rows), one unit proof at 2^%d rows "
                      "x %d cols (+%d aux) in %.2f s, %s; a naive scalar port, NOT the reference's AVX2/rayon prover: the GPU/CPU ratio "
                      "is not a speed-up over the reference" % (cores, avail, log_c, log_s, n_cols, 2 * (n_helpers + 1) if n_helpers else 0, t_s, how)}


def golden_digest(log_n, n_cols, n_helpers, seed):
    """sha256 of the ORACLE's proof of this bench unit, when committed (tests/golden/proof_digests_big.json: 2^20 / 2^22 / 2^23)"""
    f = os.path.join(ROOT, "tests", "golden", "proof_digests_big.json")
    if not os.path.exists(f):
        return None
    for e in json.load(open(f)).get("unit_proofs", []):
        if (e["log_n"], e["n_cols"], e["n_helpers"], e["seed"]) == (log_n, n_cols, n_helpers, seed):
            return e["sha256"]
    return None


def live_pmc_bytes(log_n, n_cols, timeout_s=90):
    """HBM bytes of ONE leaf-hash launch from the PMC counters, collected NOW: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: each in
    a run of its own with --kernel-trace only, as the pool requires) over a child process that commits one n_cols-column 2^log_n-row batch
    (tools/commit_bench.py: the very leaf-hash launch of the bench's unit).  FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for
    gfx950 (128-B requests tallied at 64 B); both counters are in KB.  Returns (bytes, detail) or raises: the caller falls back to the
    newest committed pass."""
    import csv, shutil, tempfile
    if not shutil.which("rocprofv3"):
        raise RuntimeError("rocprofv3 not on PATH")
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        raise RuntimeError("this process already runs under a profiler")
    kern = "hash_leaves_colmajor_mfma_kernel"
    vals = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="zk_pmc_", dir="/tmp")
        try:
            env = dict(os.environ, TMPDIR="/tmp")
            r = subprocess.run(["rocprofv3", "--kernel-trace", "--output-format", "csv", "--pmc", counter, "-d", d, "-o", "c", "--",
                                sys.executable, os.path.join(ROOT, "tools", "commit_bench.py"), str(log_n), str(n_cols), "1"],
                               cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout_s)
            files = glob.glob(os.path.join(d, "**", "c_counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                raise RuntimeError("rocprofv3 --pmc %s: rc %d%s" % (counter, r.returncode, "" if files else ", no counter file"))
            tot, cnt = 0.0, 0
            for row in csv.DictReader(open(files[0])):
                if row["Kernel_Name"].startswith(kern) and row["Counter_Name"] == counter:
                    tot += float(row["Counter_Value"]); cnt += 1
            if not cnt:
                raise RuntimeError("rocprofv3 --pmc %s: no %s dispatch in the counter file" % (counter, kern))
            vals[counter] = tot / cnt
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return (2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0, {"FETCH_SIZE_KB": vals["FETCH_SIZE"], "WRITE_SIZE_KB": vals["WRITE_SIZE"]}


def int_alu_peak():
    """measured wave-instructions/s of the chip for v_mad_u64_u32 (the instruction class the hashing kernels are made of),
    from the newest committed run of tools/ubench_int.hip"""
    f = newest_profile("r*_ubench_int_gfx950.txt")
    for line in open(f):
        if line.startswith("v_mad_u64_u32"):
            gops_lane = float(line.split()[2])
            return gops_lane * 1e9 / 64.0, os.path.basename(f)
    raise ValueError("%s: no v_mad_u64_u32 row" % f)


def main():
    if os.environ.get("ZK_BENCH_DUMP_ON_USR1"):          # soak runs: `timeout -s USR1 -k 20 N python bench.py ...` gets every thread's stack before the kill
        import faulthandler
        import signal
        faulthandler.register(signal.SIGUSR1, all_threads=True)
    args = parse_args()
    if args.workload == "prove_root":
        return bench_prove_root(args)
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        sys.exit(fan_out(args))
    if env_world is not None and int(env_world) != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%s (launch one rank per GPU, or drop the launcher and let --gpus fan out)"
                 % (args.gpus, env_world))
    if not args.inflight:
        args.inflight = 4 if args.log_n < 22 else 2
    if os.environ.get("ZK_BENCH_FANOUT_PROBE"):     # test hook (tests/test_bench_cpu.py): show the rank environment, touch nothing
        print(json.dumps({"rank": int(os.environ.get("RANK", "0")), "world": int(env_world or "1"),
                          "local_rank": int(os.environ.get("LOCAL_RANK", "0")), "master": os.environ.get("MASTER_ADDR")}))
        return

    import numpy as np
    import torch
    import torch.distributed as dist
    import eth_tx_proof_amd as E
    from eth_tx_proof_amd import air, lib as zl, sharding

    world = int(env_world or "1")
    # ZK_BENCH_FORCE_DIST=1: the process group, barrier, max-over-ranks reduction and the hand-off objects of the N > 1 path also at N = 1 --
    # the RCCL (nccl) branch then runs on ONE card (tests/test_sharding_gpu.py): the closest rehearsal of it that a single-GPU box allows
    use_dist = world > 1 or bool(os.environ.get("ZK_BENCH_FORCE_DIST"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # one rank per GPU over RCCL.  ZK_BENCH_BACKEND=gloo is a REHEARSAL switch: it lets several ranks share one card (RCCL refuses
    # that), so that the N > 1 code path -- sharded units, pipelined steps, the proof hand-off -- runs for real on a one-GPU box
    backend = os.environ.get("ZK_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local = local % max(1, torch.cuda.device_count())
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # finished proofs are host buffers: they are handed to rank 0 over a gloo group.  gloo prints its connection report on
        # stdout from C++ (so does the rehearsal's gloo default group): keep it away from the ONE JSON line this program owes its caller
        sys.stdout.flush()
        _fd1 = os.dup(1)
        os.dup2(2, 1)
        try:
            if backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
            else:
                dist.init_process_group(backend, rank=rank, world_size=world)
            host_pg = sharding.host_group()
        finally:
            os.dup2(_fd1, 1)
            os.close(_fd1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    cfg = E.standard_fast_config()
    prog = air.cpu_like_air(args.cols, args.helpers)
    C, A, Q = args.cols, prog.n_aux_cols, cfg.num_challenges << cfg.rate_bits
    n = 1 << args.log_n

    # one ctx (= one stream) per in-flight unit; traces resident in HBM (torch owns the buffers: plumbing only)
    ctxs, traces, pubs, outs = [], [], [], []
    for s in range(args.inflight):
        c = zl.Context(local)
        if os.environ.get("ZK_BENCH_CTX_TRACES"):        # debugging aid: traces from the context's allocator (fence-able)
            class _T:                                     # same interface as the torch tensor below
                def __init__(self, p): self.p = p
                def data_ptr(self): return self.p
            t = _T(c.dev_alloc(8 * C * n))
        else:
            t = torch.empty((C, n), dtype=torch.int64, device=dev)
        _, pub = zl.synth_trace_dev(c, C, args.log_n, 0x5EED0000 + 1000 * (rank * args.inflight + s), dptr=t.data_ptr())
        ctxs.append(c); traces.append(t); pubs.append(pub)
        outs.append(np.empty(zl.proof_size_words(cfg, [(prog, args.log_n, t.data_ptr(), pub)]), dtype=np.uint64))
    for c in ctxs:
        c.synchronize()
    # prover state (the reference builds it once at start-up, common/src/prover_state/mod.rs:59-95): constraint kernels of tables
    # that were not compiled with the library are compiled now (hiprtc), in this thread, outside the timed region
    zl.prover_state_prepare(ctxs[0], cfg, [(prog, args.log_n, traces[0].data_ptr(), pubs[0])])

    errors = []

    def work(s, k=1, ho=None):
        """Worker s proves k units back to back on its own context / stream (a paladin worker pulls its next unit as soon as one is done);
        with several ranks every finished proof goes straight to the background hand-off (sharding.Handoff), never to a list."""
        try:
            for j in range(k):
                zl.prove_tables(ctxs[s], cfg, [(prog, args.log_n, traces[s].data_ptr(), pubs[s])], out=outs[s])
                if ho is not None:
                    ho.put(rank + world * (j * args.inflight + s), outs[s])        # unit m of this rank = rank + world * m
        except Exception as e:   # surfaced after join
            errors.append(e)

    handoff = {"exposed_s": 0.0, "bytes": 0}
    own = {"proving_s": 0.0}       # this rank's own units alone: first launch to the last proof on the host (before the hand-off ends)

    def run_steps(k):
        """k steps = k x `inflight` unit proofs on this rank.  The steps are pipelined: each of the `inflight` workers proves its k units
        back to back, so the GPU does not drain at every step boundary -- the barrier + synchronize brackets are around the whole
        timed region, as the contract says, not around each step.  N > 1: the finished proofs travel to rank 0 (where
        leader.ProverInput.prove folds them) WHILE the next units are proved; what is left at the end -- the last proofs in flight and one
        status broadcast -- is `handoff.exposed_s`."""
        ho = sharding.Handoff(k * args.inflight * world, group=host_pg) if use_dist else None
        th = [threading.Thread(target=work, args=(s, k, ho)) for s in range(args.inflight)]
        t_own = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        own["proving_s"] = time.perf_counter() - t_own
        if ho is not None:
            got = ho.finish(ok=not errors)
            handoff["exposed_s"], handoff["bytes"] = ho.exposed_s, ho.bytes
            if rank == 0 and any(g is None for g in got):
                raise RuntimeError("hand-off: a proof is missing on rank 0")
        if errors:
            raise errors[0]

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(k):
        fence()
        t0 = time.perf_counter()
        run_steps(k)
        fence()
        dt = time.perf_counter() - t0
        if use_dist:
            tt = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else None)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt

    dbg = (lambda m: print("[bench] " + m, file=sys.stderr, flush=True)) if os.environ.get("ZK_BENCH_DEBUG") else (lambda m: None)
    if args.warmup > 0:
        run_steps(args.warmup)
        fence()
        dbg("warmup steps done")

    # ---- the headline: exactly --steps steps, profiler off
    dt = timed(args.steps)
    dbg("timed steps done")
    units = args.steps * args.inflight * world
    value = units / dt
    handoff_timed = dict(handoff)
    # every rank's own rate over the timed region (an imbalance between GPUs shows here before it shows in the job's aggregate)
    per_rank = None
    if use_dist:
        mine = {"rank": rank, "units": args.steps * args.inflight, "proving_s": own["proving_s"], "proofs_per_s": args.steps * args.inflight / own["proving_s"],
                "device": torch.cuda.get_device_name(dev) if torch.cuda.is_available() else None}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine, group=host_pg)
    timed_out0 = outs[0].copy()             # the proof worker 0 wrote INSIDE the timed region (the later regions overwrite outs[0])

    # ---- the same steps with the hipEvent profiler on every stream: kernel table and roofline
    solo, rep, dt_prof, phases = {}, {}, None, {}
    if not args.no_profile:
        # one unit ALONE on the GPU first: which kernel dominates is decided from these solo durations (when several
        # streams share the chip, a kernel's elapsed time also contains time spent waiting for CUs)
        ctxs[0].profile(True)
        ctxs[0].profile_reset()
        work(0)
        if errors:
            raise errors[0]
        solo_all = ctxs[0].profile_report()
        phases = {k[7:]: v for k, v in solo_all.items() if k.startswith("phase: ")}      # upstream's timed! scopes (ranges, not kernels)
        solo = {k: v for k, v in solo_all.items() if not k.startswith("phase: ")}
        dbg("solo profiled unit done")
        for c in ctxs:
            c.profile(True)
            c.profile_reset()
        dt_prof = timed(args.steps)
        for c in ctxs:
            for k, (l, ms) in c.profile_report().items():
                if k.startswith("phase: "):
                    continue
                a = rep.get(k, (0, 0.0))
                rep[k] = (a[0] + l, a[1] + ms)
            c.profile(False)
        dbg("profiled steps done")

    # ---- PCIe-inclusive: the same units from host-resident (pinned) traces, never `value`.  Every worker streams its units the way a
    # paladin worker would: while unit k is being proved, unit k+1's trace is already on its way up (zk_trace_upload_async on the
    # context's copy stream), so the PCIe transfer hides behind the proof instead of preceding it.
    pcie = None
    if not args.no_pcie and world == 1:
        hosts = []
        for s in range(args.inflight):
            h = zl.pinned_empty(ctxs[s], (C, n))
            h[...] = ctxs[s].download(traces[s].data_ptr(), (C, n))
            hosts.append(h)
        k = max(2, 2 * args.steps)
        # what the link gives on this box: one unit's trace, pinned -> device, alone on an idle GPU (the same upload call)
        fence()
        t0 = time.perf_counter()
        d0 = zl.trace_upload_async(ctxs[0], hosts[0])
        ctxs[0].dev_free(d0)                     # waits (on the device) for the upload; the stream sync below waits for that
        ctxs[0].synchronize()
        h2d_gbps = 8 * C * n / (time.perf_counter() - t0) / 1e9

        def stream_units(s):
            try:
                nxt = zl.trace_upload_async(ctxs[s], hosts[s])
                for i in range(k):
                    cur = nxt
                    if i + 1 < k:
                        nxt = zl.trace_upload_async(ctxs[s], hosts[s])
                    zl.prove_tables(ctxs[s], cfg, [(prog, args.log_n, cur, pubs[s])], out=outs[s])
                    ctxs[s].dev_free(cur)
            except Exception as e:
                errors.append(e)
        for warm in range(2):                    # first pass untimed (allocations, pinned chunks), second timed
            fence()
            t0 = time.perf_counter()
            th = [threading.Thread(target=stream_units, args=(s,)) for s in range(args.inflight)]
            for t in th:
                t.start()
            for t in th:
                t.join()
            fence()
            dt_h = time.perf_counter() - t0
            if errors:
                raise errors[0]
        pcie = {"value": k * args.inflight / dt_h, "unit": "proofs/s", "units": k * args.inflight,
                "h2d_GBps_one_trace_alone": round(h2d_gbps, 1), "link_bound_proofs_per_s": round(h2d_gbps * 1e9 / (8 * C * n), 1),
                "note": "traces in pinned host memory (%d MB per unit); each worker uploads unit k+1 on its copy stream while unit k is "
                        "proved (zk_trace_upload_async), first upload of every worker included in the time" % ((8 * C * n) >> 20)}
        del hosts
        dbg("pcie-inclusive units done")

    if rank == 0:
        line = {
            "metric": "segment STARK proofs/sec (2^%d CPU_CIRCUIT rows)" % args.log_n, "value": value, "unit": "proofs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64 (Goldilocks field)",
            "data": "synthetic",
            "config": {"workload": "%s: one unit STARK+FRI proof, 2^%d rows x %d trace cols + %d aux cols, "
                                   "standard_fast_config (rate_bits 1, cap 4, PoW 16, arity 16, 84 queries), synthetic degree-3 AIR"
                                   % ("BASELINE configs[2]" if args.log_n == 20 else "north_star target shape" if args.log_n == 23
                                      else "configs[2] shape at another size", args.log_n, C, A),
                       "units_per_step_per_gpu": args.inflight, "parallelism": "independent units, %d GPU(s)" % world},
            "value_profiled": (units / dt_prof) if dt_prof else None,
            "value_pcie_inclusive": pcie,
        }
        if use_dist:       # the only exchange of the job: finished proofs to rank 0, overlapped with proving (sharding.Handoff)
            line["per_rank"] = per_rank
            line["handoff"] = {"exposed_ms": handoff_timed["exposed_s"] * 1e3, "share_of_timed_region": handoff_timed["exposed_s"] / dt,
                               "MB_into_rank0": handoff_timed["bytes"] / 1e6,
                               "backend": ("RCCL default group (ZK_HANDOFF=rccl): ncclSend / ncclRecv of device tensors, rank 0 asks the ranks in turn" if os.environ.get("ZK_HANDOFF", "gloo").lower() == "rccl" and backend == "nccl"
                                           else "gloo host group, per-proof send from a background thread (ZK_HANDOFF=rccl: the same over RCCL)"),
                               "note": "exposed = what rank 0 still waited for after its own last unit (last proofs in flight + one status broadcast)"}
        if rep:
            proofs0 = args.steps * args.inflight     # unit proofs this rank produced in the profiled region
            tot_ms = sum(ms for _, ms in rep.values()) or 1.0
            dname = max(solo.items(), key=lambda kv: kv[1][1])[0]   # dominant kernel = largest total time when run alone
            dlaunch, dms = rep[dname]                                # its launches and hipEvent time over the profiled region
            solo_avg_ms = solo[dname][1] / solo[dname][0]
            per_launch = algorithmic_bytes_per_launch(dname, args.log_n, cfg.rate_bits)
            achieved = per_launch / (dms / dlaunch * 1e-3) / 1e9 if per_launch else None
            traffic, alu = None, None
            dcols = leaf_label_cols(dname)
            if dcols is not None:
                # HBM bytes and instruction mix from the newest committed rocprofv3 PMC passes over this kernel
                # (tools/profile_round.sh -> profiles/rNN_pmc_leaf_hash.json); loud failure when there is none
                pmf = newest_profile("r*_pmc_leaf_hash.json")
                pm = json.load(open(pmf))["leaf_hash_93cols"]
                traffic = per_launch * pm["pmc_bytes"] / pm["algorithmic_bytes"]
                traffic_src = {"source": "committed " + os.path.basename(pmf), "scaled_from_93_columns": dcols != 93}
                if not args.no_live_pmc and world == 1:
                    # ... measured in THIS run when the profiler is there: a child process under rocprofv3 --pmc (two passes, ~10 s each)
                    try:
                        traffic, det = live_pmc_bytes(args.log_n, dcols)
                        traffic_src = dict(det, source="live: rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE -- tools/commit_bench.py %d %d 1, "
                                                       "2 x FETCH_SIZE + WRITE_SIZE (KB), mean over the dispatches of the kernel" % (args.log_n, dcols))
                    except Exception as e:       # no profiler, refused, timed out: the committed pass stands, and the line says so
                        traffic_src["live_attempt_failed"] = str(e)[:200]
                perms_per_launch = (2 * n) * (-(-dcols // 8))
                vw = pm["valu_instr_per_wave_permutation"]
                peak_wi, ubf = int_alu_peak()
                pps = perms_per_launch / (dms / dlaunch * 1e-3)
                alu = {"permutations_per_s": pps,
                       "valu_instr_per_wave_of_64_permutations": vw,
                       "mfma_i8_instr_per_wave_of_64_permutations": pm["mfma_i8_instr_per_wave_permutation"],
                       "valu_issue_bound_permutations_per_s": peak_wi / vw * 64,
                       "frac_of_valu_issue_bound": pps / (peak_wi / vw * 64),
                       "sources": [os.path.basename(pmf), ubf],
                       "note": "the linear layers run on the i8 matrix cores, the s-boxes and limb recombination on the vector "
                               "ALU; bound = measured chip rate of v_mad_u64_u32 wave-instructions (tools/ubench_int.hip) / VALU "
                               "instructions per 64-permutation wave (SQ_INSTS_VALU)"}
            if alu is not None:
                # what the chip gave this kernel when it was last counted (tools/pmc_round4.sh, committed summary): the clock (GRBM_GUI_ACTIVE / 8 /
                # duration) and the share of SIMD cycles that issued an instruction (SQ_ACTIVE_INST_ANY x 4 quad-cycles / 1024 SIMDs / kernel cycles)
                try:
                    pf = newest_profile("r*_pmc_bench_regions_and_ntt_summary.json")
                    pj = json.load(open(pf))
                    kn = "hash_leaves_colmajor_mfma_kernel"
                    ghz = pj["derived"]["effective_clock_GHz"][kn]["GHz"]
                    act = [v[kn]["mean"] for t in pj["counters"].values() for c, v in t.items() if c == "SQ_ACTIVE_INST_ANY" and kn in v][0]
                    gui = [v[kn]["mean"] for t in pj["counters"].values() for c, v in t.items() if c == "GRBM_GUI_ACTIVE" and kn in v][0]
                    alu["clock_GHz_under_this_kernel"] = round(ghz, 3)
                    alu["simd_issue_busy_fraction"] = round(act * 4.0 / 1024.0 / (gui / 8.0), 3)
                    alu["frac_of_valu_issue_bound_at_that_clock"] = round(alu["frac_of_valu_issue_bound"] * 2.4 / ghz, 3)
                    alu["sources"].append(os.path.basename(pf))
                except Exception as e:
                    alu["clock_note"] = "no committed counter summary: %s" % str(e)[:80]
            line["roofline"] = {
                "bound": "hbm", "kernel": dname, "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": (achieved / HBM_PEAK_GBPS) if achieved else None, "traffic": traffic, "traffic_source": traffic_src if dcols is not None else None, "alu": alu,
                "avg_launch_ms": dms / dlaunch, "avg_launch_ms_alone": solo_avg_ms, "launches": dlaunch,
                "algorithmic_bytes_per_launch": per_launch, "share_of_kernel_time": dms / tot_ms,
                "note": "Poseidon hashing is integer-ALU-bound; the HBM fraction is reported as the contract asks, `alu` and "
                        "DESIGN.md give the ALU roofline"}
            # NTT GB/s (second half of BASELINE.json's metric): all NTT pass kernels, algorithmic 16 B per element per transform
            ntt_ms = sum(ms for k, (_, ms) in solo.items() if k.startswith("ntt_"))      # NTT kernels of one unit run alone
            L = n << cfg.rate_bits
            # every transform the `ntt_*` launches of one proof perform, each counted once as read + written (8 B per element):
            #   commits: iNTT of the trace and aux values, LDE (n -> L) of trace, aux and quotient-chunk coefficients, iNTT (size L) of the
            #   quotient values; FRI (prover_fri.hip): the alpha-combined polynomials onto the LDE coset (2 words x nb opening batches),
            #   the final values back to coefficients (size L, 2 words), one forward transform per commit-phase layer (2 words)
            nb = 3 if A else 2
            fri_layers, d = [], args.log_n
            while d > cfg.fri_final_poly_bits and d + cfg.rate_bits - cfg.fri_arity_bits >= cfg.cap_height:
                d -= cfg.fri_arity_bits
                fri_layers.append(1 << (d + cfg.rate_bits))
            ntt_bytes_commits = 16 * n * (C + A) + (8 * n + 8 * L) * (C + A + Q) + 16 * L * cfg.num_challenges
            ntt_bytes_fri = (8 * n + 8 * L) * 2 * nb + 16 * L * 2 + sum(16 * cur * 2 for cur in fri_layers)
            ntt_bytes = ntt_bytes_commits + ntt_bytes_fri
            line["ntt"] = {"GBps_algorithmic": (ntt_bytes / (ntt_ms * 1e-3) / 1e9) if ntt_ms else None, "ms_per_proof": ntt_ms,
                           "algorithmic_bytes_per_proof": {"commits": ntt_bytes_commits, "fri": ntt_bytes_fri},
                           "frac_of_hbm_peak": (ntt_bytes / (ntt_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS) if ntt_ms else None}
            line["kernels_one_unit_alone"] = {k: {"launches_per_proof": l, "ms_per_proof_alone": round(ms, 3)}
                                              for k, (l, ms) in sorted(solo.items(), key=lambda kv: -kv[1][1])[:14]}
            line["phases_one_unit_alone_ms"] = {k: round(ms, 3) for k, (l, ms) in sorted(phases.items(), key=lambda kv: -kv[1][1])}
            top = [k for k, _ in sorted(rep.items(), key=lambda kv: -kv[1][1])[:8]]
            top += [k for k in rep if leaf_label_cols(k) is not None and k not in top]       # every leaf-hash shape, always
            line["kernels_profiled_region"] = {k: {"launches": rep[k][0], "avg_ms": round(rep[k][1] / rep[k][0], 4)} for k in top}
            # rocprofv3 --stats averages per KERNEL NAME: hash_leaves_colmajor_mfma_kernel covers every leaf-hash shape of a proof
            # (93-column trace commit and 40-column auxiliary commit); this is the same mixed average from the hipEvent timings
            ll = [(l, ms) for k, (l, ms) in rep.items() if leaf_label_cols(k) is not None]
            line["roofline"]["avg_launch_ms_all_leaf_hash_shapes"] = sum(ms for _, ms in ll) / max(1, sum(l for l, _ in ll))
        # ---- a proof of the timed region, checked after the clock has stopped: its sha256 against the oracle's committed digest of the
        # same unit (bit-exact parity without oracle time), and the oracle's verifier in the cpu_baseline leg
        import hashlib
        seed0 = 0x5EED0000 + 1000 * (rank * args.inflight)
        gold = golden_digest(args.log_n, C, args.helpers, seed0)
        got = hashlib.sha256(timed_out0.tobytes()).hexdigest()
        line["timed_proof_check"] = {"unit": "worker 0, seed %#x" % seed0, "sha256": got, "oracle_sha256": gold,
                                     "identical_to_oracle": (got == gold) if gold else None}
        if gold and got != gold:
            print(json.dumps(line))
            sys.exit("bench.py: the proof of the timed region differs from the oracle's (sha256 %s != %s)" % (got, gold))
        # cpu_baseline is MEASURED at N = 1 only (the contract); that run leaves its record in a small cache file, and the N > 1 runs of the
        # same box (the driver runs N = 1, 2, 4, 8 back to back) carry it along, marked as cached
        cache_key = "%s/2^%d x %d+%d" % (os.uname().nodename, args.log_n, C, args.helpers)
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(args.log_n, C, args.helpers, check_proof=(timed_out0, pubs[0]))
            if line["cpu_baseline"]["timed_proof_verified"] is False:
                print(json.dumps(line))
                sys.exit("bench.py: the oracle's verifier rejects a proof of the timed region")
            for path in CPU_BASELINE_CACHES:
                try:
                    os.makedirs(os.path.dirname(path), exist_ok=True)
                    json.dump({"key": cache_key, "measured_unix": time.time(), "record": line["cpu_baseline"]}, open(path, "w"))
                except OSError:
                    pass
        elif not args.no_cpu_baseline:
            line["cpu_baseline"] = None
            for path in CPU_BASELINE_CACHES:
                try:
                    c = json.load(open(path))
                    if c.get("key") == cache_key:
                        line["cpu_baseline"] = dict(c["record"], cached_from="the N = 1 run of this box %.0f s earlier (%s)" % (time.time() - c["measured_unix"], os.path.basename(path)))
                        break
                except (OSError, ValueError, KeyError):
                    pass
        else:
            line["cpu_baseline"] = None
    for c in ctxs:
        c.close()
    ctxs = []
    if rank == 0:
        # ---- the operator's whole call: a short transaction record beside BASELINE's unit -- 8 transactions in flight x 4
        # steps (two steps are mostly ramp: 19 against 26 tx/s), one transaction alone split into its three stages, root proof and AllProof
        # against the oracle's committed digests
        if world == 1 and args.log_n == 20 and not args.no_prove_root:
            line["prove_root"] = prove_root_record(args)
        if world == 1 and args.log_n == 20 and not args.no_log23 and not os.environ.get("ZK_BENCH_CHILD"):
            line["log23"] = log23_record(args)
        print(json.dumps(line))
        if isinstance(line.get("log23"), dict) and line["log23"].get("identical_to_oracle") is False:
            sys.exit("bench.py: the 2^23-row unit's proof differs from the oracle's committed digest")
        if isinstance(line.get("prove_root"), dict) and (line["prove_root"].get("timed_proof_check") or {}).get("identical_to_oracle") is False:
            sys.exit("bench.py: the transaction's proofs differ from the oracle's committed digests")
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
