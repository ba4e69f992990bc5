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
    local = {u: np.full(37, 1000 + u, dtype=np.uint64) + np.arange(37, dtype=np.uint64) for u in mine}   # stand-in proofs
    got = sharding.gather_proofs(local, n_units)
    if rank == 0:
        assert len(got) == n_units
        for u in range(n_units):
            assert (got[u] == np.full(37, 1000 + u, dtype=np.uint64) + np.arange(37, dtype=np.uint64)).all()
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


def test_shard_units_cover_everything_once():
    sys.path.insert(0, ROOT)
    import eth_tx_proof_amd  # noqa: F401
    from eth_tx_proof_amd import sharding
    for n in (0, 1, 7, 16):
        for w in (1, 2, 8):
            allu = sorted(u for r in range(w) for u in sharding.shard_units(n, w, r))
            assert allu == list(range(n))
