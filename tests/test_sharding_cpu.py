"""N>1 path on CPU: world_size-2 gloo processes shard units round-robin and hand finished proofs to rank 0."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r)
    import numpy as np, torch.distributed as dist
    import eth_tx_proof_amd
    from eth_tx_proof_amd import sharding
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    n_units = 5
    mine = sharding.shard_units(n_units, world, rank)
    assert mine == list(range(rank, n_units, world))
    # stand-in proofs of DIFFERENT lengths (units with different table shapes)
    standin = lambda u: np.full(37 + 3 * u, 1000 + u, dtype=np.uint64) + np.arange(37 + 3 * u, dtype=np.uint64)
    local = {u: standin(u) for u in mine}
    got = sharding.gather_proofs(local, n_units)
    again = sharding.gather_proofs(local, n_units, group=sharding.host_group())      # the hand-off group bench.py uses on GPUs
    assert (got is None) == (again is None) and (got is None or all((a == b).all() for a, b in zip(got, again)))
    if rank == 0:
        assert len(got) == n_units
        for u in range(n_units):
            assert got[u].shape == standin(u).shape and (got[u] == standin(u)).all()
        print("GATHER_OK")
    else:
        assert got is None
    dist.destroy_process_group()
""") % ROOT


def test_two_rank_gloo_shard_and_gather(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29611", str(script)],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "GATHER_OK" in out.stdout


HANDOFF_WORKER = textwrap.dedent("""
    import os, sys, time
    sys.path.insert(0, %r)
    import numpy as np, torch.distributed as dist
    import eth_tx_proof_amd
    from eth_tx_proof_amd import sharding
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    n_units = 7
    standin = lambda u: np.full(50 + 11 * u, 7000 + u, dtype=np.uint64) + np.arange(50 + 11 * u, dtype=np.uint64)
    # ---- proofs handed over one by one while the 'proving' goes on; rank 0 only waits for what is still in flight at the end
    ho = sharding.Handoff(n_units)                 # group None: the cached host (gloo) group
    for u in sharding.shard_units(n_units, world, rank):
        time.sleep(0.05)                            # the next unit is being proved
        ho.put(u, standin(u))
    got = ho.finish(ok=True)
    if rank == 0:
        assert len(got) == n_units and all((got[u] == standin(u)).all() for u in range(n_units))
        assert ho.bytes == sum(8 * standin(u).size for r in range(1, world) for u in sharding.shard_units(n_units, world, r))
        # the hand-off hid behind the proving (three rank processes taking turns on a busy 8-core box get a wider margin)
        assert ho.exposed_s < (0.25 if os.environ.get("ZK_HANDOFF") == "rccl" else 0.04), ho.exposed_s
    else:
        assert got is None
    # ---- a rank whose map stage failed: EVERY rank raises instead of waiting for proofs that never come
    ho = sharding.Handoff(n_units)
    mine = sharding.shard_units(n_units, world, rank)
    for u in mine[:1]:
        ho.put(u, standin(u))
    try:
        ho.finish(ok=(rank != 1))
        raised = False
    except sharding.HandoffError:
        raised = True
    assert raised
    dist.barrier()
    print("HANDOFF_OK_%%d" %% rank)
    dist.destroy_process_group()
""") % ROOT


def test_two_rank_overlapped_handoff_and_failure_propagation(tmp_path):
    """sharding.Handoff (what leader.ProverInput.prove and bench.py --gpus N use): per-proof sends from a background thread, receive
    loop on rank 0, one status broadcast at the end so that a failed rank cannot leave the others blocked."""
    script = tmp_path / "handoff_worker.py"
    script.write_text(HANDOFF_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29613", str(script)],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "HANDOFF_OK_0" in out.stdout and "HANDOFF_OK_1" in out.stdout


def test_two_rank_ordered_handoff_the_rccl_protocol_over_gloo(tmp_path):
    """ZK_HANDOFF=rccl (opt-in): headers and payloads over the job's DEFAULT group with source-specific receives -- the protocol RCCL's
    send / recv pairs need (no receive from any source).  With a gloo default group the same code path runs on host tensors: the same
    worker script as above must pass, proofs complete and in order, failure propagated to every rank."""
    script = tmp_path / "handoff_worker_ordered.py"
    script.write_text(HANDOFF_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", ZK_HANDOFF="rccl")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=3",
                          "--master-addr", "127.0.0.1", "--master-port", "29617", str(script)],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    assert all("HANDOFF_OK_%d" % r in out.stdout for r in range(3))


def test_shard_units_cover_everything_once():
    sys.path.insert(0, ROOT)
    import eth_tx_proof_amd  # noqa: F401
    from eth_tx_proof_amd import sharding
    for n in (0, 1, 7, 16):
        for w in (1, 2, 8):
            allu = sorted(u for r in range(w) for u in sharding.shard_units(n, w, r))
            assert allu == list(range(n))


def test_runtime_fold_is_a_balanced_pairwise_tree():
    sys.path.insert(0, ROOT)
    import eth_tx_proof_amd  # noqa: F401
    from eth_tx_proof_amd import leader

    class Cat:
        def combine(self, a, b):
            return "(%s%s)" % (a, b)

        def empty(self):
            raise NotImplementedError("empty agg proof")
    assert leader.Runtime.fold(Cat(), list("abcde")) == "(((ab)(cd))e)"
    assert leader.Runtime.fold(Cat(), ["a"]) == "a"
    import pytest
    with pytest.raises(NotImplementedError):
        leader.Runtime.fold(Cat(), [])


def test_runtime_map_requeues_device_errors_once():
    """leader.Runtime.map with a stand-in operator: device errors (ZK_ERR_HIP / ZK_ERR_OOM) are re-queued once, anything else and a
    second failure abort the job with the FatalError (the GPU version of this test is tests/test_ops_gpu.py)."""
    sys.path.insert(0, ROOT)
    import pytest
    import eth_tx_proof_amd  # noqa: F401
    from eth_tx_proof_amd import leader, lib as zl, ops
    old = ops.P_STATE
    ops.P_STATE = None
    try:
        ops.set_prover_state_from_config(0, stark_only=True)

        class Op:
            def __init__(self, fail_unit, times, code):
                self.fail_unit, self.left, self.code = fail_unit, times, code

            def execute(self, x):
                if x == self.fail_unit and self.left > 0:
                    self.left -= 1
                    try:
                        raise zl.ZkError(self.code, "zk_prove_tables", "injected")
                    except zl.ZkError as e:
                        raise ops.FatalError(str(e)) from e
                return x * x
        rt = leader.Runtime(num_workers=3)
        assert rt.map(Op(4, 1, -3), list(range(9))) == [x * x for x in range(9)]
        assert [u for u, _ in rt.requeued] == [4]
        with pytest.raises(ops.FatalError):
            leader.Runtime(num_workers=3).map(Op(4, 2, -2), list(range(9)))
        rt = leader.Runtime(num_workers=3)
        with pytest.raises(ops.FatalError):
            rt.map(Op(4, 1, -1), list(range(9)))
        assert rt.requeued == []
    finally:
        ops.P_STATE = old
