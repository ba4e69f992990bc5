"""bench.py's launch contract, checked without a GPU: `--gpus N` fans out N ranks itself (before anything touches the GPU),
a launcher's WORLD_SIZE must agree with --gpus, and the roofline inputs it reads from profiles/ exist."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    e.update(kw)
    return e


def test_gpus_flag_fans_out_ranks():
    out = subprocess.run([sys.executable, BENCH, "--gpus", "3"], capture_output=True, text=True, timeout=120,
                         env=_env(ZK_BENCH_FANOUT_PROBE="1"))
    assert out.returncode == 0, out.stderr[-2000:]
    got = json.loads(out.stdout.strip().splitlines()[-1])          # only rank 0's line is relayed
    assert got == {"rank": 0, "world": 3, "local_rank": 0, "master": "127.0.0.1"}


def test_launcher_world_size_must_match_gpus():
    out = subprocess.run([sys.executable, BENCH, "--gpus", "8"], capture_output=True, text=True, timeout=120,
                         env=_env(ZK_BENCH_FANOUT_PROBE="1", WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"))
    assert out.returncode != 0 and "WORLD_SIZE" in out.stderr
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2"], capture_output=True, text=True, timeout=120,
                         env=_env(ZK_BENCH_FANOUT_PROBE="1", WORLD_SIZE="2", RANK="1", LOCAL_RANK="1"))
    assert out.returncode == 0 and json.loads(out.stdout)["rank"] == 1


def test_a_failing_rank_fails_the_bench():
    # without a GPU every rank fails in zk_ctx_create (there is no CPU fallback): the parent must report it
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0", "--log-n", "8", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=300, env=_env())
    import torch
    if torch.cuda.is_available():
        return
    assert out.returncode != 0


def test_roofline_inputs_are_committed():
    sys.path.insert(0, ROOT)
    import bench
    pm = json.load(open(bench.newest_profile("r*_pmc_leaf_hash.json")))["leaf_hash_93cols"]
    assert pm["pmc_bytes"] > 0 and pm["valu_instr_per_wave_permutation"] > 0
    peak, src = bench.int_alu_peak()
    assert 1e11 < peak < 2e12, (peak, src)
    assert bench.algorithmic_bytes_per_launch("poseidon_leaf_hash_c93", 20, 1) == 8 * (1 << 21) * 93 + 32 * (1 << 21)
    assert bench.algorithmic_bytes_per_launch("transcript", 20, 1) is None


def test_a_failing_transaction_child_becomes_an_error_record_not_a_lost_line(capfd):
    """The default line's transaction record is measured by a child process (bench.prove_root_record): whatever happens to that child --
    here it cannot create a context (no GPU), on a GPU box it might fault or hang -- the parent gets an error record, with the child's
    stderr passed on, and goes on to print its line."""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("the child would succeed on a GPU box")
    import importlib.util
    import types
    spec = importlib.util.spec_from_file_location("bench_under_test", BENCH)
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    rec = bench.prove_root_record(types.SimpleNamespace(prepare="never"), timeout_s=240)
    assert set(rec) >= {"error", "gpu_memory_fault", "command"} and "exited with code" in rec["error"] and rec["gpu_memory_fault"] is False
    assert "--workload prove_root" in rec["command"]
    assert capfd.readouterr().err.strip(), "the child's stderr was not passed on"
    # the 2^23-row record goes through the same child mechanism
    rec = bench.log23_record(types.SimpleNamespace(), timeout_s=240)
    assert set(rec) >= {"error", "gpu_memory_fault", "command"} and "--log-n 23" in rec["command"]
