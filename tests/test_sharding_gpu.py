"""BASELINE configs[4] on one GPU: a 16-unit block is sharded round-robin over ranks, every rank proves its units for real
(HIP prover, one context per in-flight unit), the finished proofs are handed to rank 0 with sharding.gather_proofs, and rank 0
checks every gathered proof with the oracle's verifier -- the caller shape of /root/reference/leader/src/prover.rs:26-30
(IndexedStream.map(&TxProof) then the hand-off to fold).  Two ranks share the one GPU of the test box (gloo rendezvous: RCCL
refuses two ranks on one device), each with four contexts, i.e. eight workers with two units each."""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys, threading
    sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "oracle"))
    import numpy as np, torch.distributed as dist
    import eth_tx_proof_amd as E
    from eth_tx_proof_amd import air, lib as zl, sharding
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    N_UNITS, LOG_N, COLS, HELPERS, NCTX = 16, 16, 93, 19, 4
    cfg = E.standard_fast_config()
    prog = air.cpu_like_air(COLS, HELPERS)
    mine = sharding.shard_units(N_UNITS, world, rank)
    assert len(mine) == N_UNITS // world
    ctxs = [zl.Context(0) for _ in range(NCTX)]
    local, pubs, errors = {}, {}, []
    def worker(w):                                  # worker w proves units mine[w], mine[w + NCTX], ... on its own stream
        try:
            for u in mine[w::NCTX]:
                d, pub = zl.synth_trace_dev(ctxs[w], COLS, LOG_N, 0x5EED0000 + 1000 * u)
                local[u] = zl.prove_tables(ctxs[w], cfg, [(prog, LOG_N, d, pub)]).copy()
                pubs[u] = pub
                ctxs[w].dev_free(d)
        except Exception as e:
            errors.append(e)
    th = [threading.Thread(target=worker, args=(w,)) for w in range(NCTX)]
    [t.start() for t in th]; [t.join() for t in th]
    assert not errors, errors
    got = sharding.gather_proofs(local, N_UNITS)
    allpubs = [None] * world
    dist.all_gather_object(allpubs, {u: p.tolist() for u, p in pubs.items()})
    if rank == 0:
        import orc
        pub_of = {u: np.array(p, dtype=np.uint64) for d in allpubs for u, p in d.items()}
        assert len(got) == N_UNITS and all(g is not None for g in got)
        for u in range(N_UNITS):
            assert orc.verify(cfg, [(prog, LOG_N, None, pub_of[u])], got[u]) == 0, u
        assert orc.verify(cfg, [(prog, LOG_N, None, pub_of[1])], got[0]) != 0           # a proof does not verify as another unit's
        bad = got[3].copy(); bad[len(bad) // 2] ^= np.uint64(1)
        assert orc.verify(cfg, [(prog, LOG_N, None, pub_of[3])], bad) != 0
        # unit u's proof is the one a single context produces for the same seed (sharding changes nothing)
        d, pub = zl.synth_trace_dev(ctxs[0], COLS, LOG_N, 0x5EED0000 + 1000 * 9)
        assert (zl.prove_tables(ctxs[0], cfg, [(prog, LOG_N, d, pub)]) == got[9]).all()
        print("BLOCK_OK")
    else:
        assert got is None
    for c in ctxs: c.close()
    dist.destroy_process_group()
""") % (ROOT, ROOT)


def test_sixteen_unit_block_sharded_proved_gathered_verified(tmp_path):
    script = tmp_path / "block_worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29633", str(script)],
                         capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, (out.stdout[-1500:] + out.stderr[-3000:])
    assert "BLOCK_OK" in out.stdout


def test_bench_two_rank_path_rehearsed_on_one_gpu(tmp_path):
    """bench.py's N > 1 path -- units sharded over ranks, pipelined steps, finished proofs handed to rank 0 over the host group, max-over-
    ranks timing -- run for real with two rank processes on this one card (ZK_BENCH_BACKEND=gloo: RCCL refuses two ranks on one GPU;
    the driver's scaling run uses RCCL on a whole node).  Small shape so that it takes seconds."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ZK_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", "29677", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--inflight", "2",
                          "--log-n", "16", "--no-pcie", "--no-cpu-baseline", "--no-profile"], capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["value"] > 0 and line["scaling"] == "weak"
    assert line["config"]["units_per_step_per_gpu"] == 2


LEADER_WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "oracle"))
    import numpy as np, torch.distributed as dist
    import eth_tx_proof_amd as E
    from eth_tx_proof_amd import air, leader, ops, plonk, sharding
    import orc
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    orc.lib().orc_set_threads(8)
    rec = plonk.SyntheticRecursion(size_shift=-7, threshold=4, seed=5)
    st = ops.set_prover_state_from_config(0, recursion=rec)
    prog = air.cpu_like_air(9, 2)
    units, unit_items = [], []
    for u in range(6):                                     # every rank holds the job description (the reference's proof_gen_ir)
        tr, pub = orc.synth_trace(9, 8, 300 + u)
        unit_items.append([(prog, 8, tr, pub)])
        units.append(ops.GenerationInputs([ops.TableInput(prog, 8, tr, pub)], txn_number_before=u, block_number=1))
    assert [b for _, _, b in rec.shrink_chain(0, prog, 8)] == [5, 5]      # initial wrapper + one shrinking wrapper per unit
    block = leader.ProverInput(units).prove(leader.Runtime(num_workers=2), group=sharding.host_group())
    if rank != 0:
        assert block is None
    else:
        # the same job through the oracle, single process: prove_root per unit, the balanced fold, the block proof
        ref, pubs, _ = orc.prove_block(st.config, rec.cfg, unit_items, rec)
        assert (block.intern == ref).all() and block.b_height == 1
        assert orc.verify_plonk(orc._plonk_state_for(rec.cfg, rec.circuit("block")[0]), pubs, block.intern) == 0
        print("LEADER_OK")
    dist.barrier()
    dist.destroy_process_group()
""") % (ROOT, ROOT)


def test_prover_input_prove_over_two_ranks(tmp_path):
    """leader.ProverInput.prove with the map stage sharded over two rank processes (both on this one card, gloo): units proved where
    they land, proofs handed to rank 0 over the host group, fold and block proof there -- and the block proof is the one the oracle
    produces for the whole job in a single process."""
    script = tmp_path / "leader_worker.py"
    script.write_text(LEADER_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", "29688", str(script)], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "LEADER_OK" in out.stdout


def test_bench_two_ranks_at_configs4_full_unit_size(tmp_path):
    """BASELINE configs[4] rehearsed at FULL unit size: 16 units of 2^20 rows x 93 + 40 columns -- two rank processes on
    this one card (gloo), four units in flight each, two steps -- through bench.py's N > 1 path: sharded units, per-proof hand-off to rank
    0 while the next units are proved, max-over-ranks timing.  The line carries what the first run on a whole node will be read by:
    every rank's own proofs/s, the exposed part of the hand-off, and the timed proof against the oracle's committed digest."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ZK_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", "29699", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--inflight", "4",
                          "--log-n", "20", "--no-pcie", "--no-cpu-baseline", "--no-profile", "--no-prove-root"], capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["config"]["units_per_step_per_gpu"] == 4
    assert [r["rank"] for r in line["per_rank"]] == [0, 1] and all(r["units"] == 8 and r["proofs_per_s"] > 0 for r in line["per_rank"])
    assert line["handoff"]["MB_into_rank0"] > 0 and line["handoff"]["exposed_ms"] >= 0
    assert line["timed_proof_check"]["identical_to_oracle"] is True
    art = os.environ.get("ZK_TEST_ARTIFACTS")
    if art:
        os.makedirs(art, exist_ok=True)
        open(os.path.join(art, "bench_2ranks_full_size_gloo_rehearsal.json"), "w").write(json.dumps(line) + "\n")


@pytest.mark.gpu
def test_two_ranks_on_one_card_add_up_to_the_single_rank_rate(tmp_path):
    """`--gpus 1` and the two-rank rehearsal must agree -- with the same number of units in flight on the card (4), one
    rank with four workers and two ranks with two workers each do the same work on the same hardware, so the ranks' own proofs/s must
    add up to the single rank's rate (within 30 %: two processes share the card's queues, not its arithmetic) and the whole-job value of
    the two-rank line must be that sum.  Also: the N > 1 line carries the cpu_baseline record the N = 1 run cached, marked as cached."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--steps", "3", "--warmup", "1", "--log-n", "18", "--no-pcie", "--no-profile", "--no-prove-root", "--no-log23"]      # (at 2^16 the units are latency-bound and two processes x two threads are 30 % behind one x four: not what is compared here)
    one = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--inflight", "4"] + common, capture_output=True, text=True, timeout=900, cwd=root)
    assert one.returncode == 0, one.stderr[-3000:]
    l1 = json.loads(one.stdout.strip().splitlines()[-1])
    assert l1["cpu_baseline"] and "cached_from" not in l1["cpu_baseline"]
    env = dict(os.environ, ZK_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29703",
                          os.path.join(root, "bench.py"), "--gpus", "2", "--inflight", "2"] + common, capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert two.returncode == 0, two.stderr[-3000:]
    l2 = json.loads(two.stdout.strip().splitlines()[-1])
    per = [r["proofs_per_s"] for r in l2["per_rank"]]
    assert len(per) == 2 and abs(sum(per) - l1["value"]) < 0.3 * l1["value"], (per, l1["value"])
    assert abs(l2["value"] - sum(per)) < 0.25 * l2["value"], (l2["value"], per)          # max-over-ranks time against each rank's own clock
    assert l2["cpu_baseline"] and "cached_from" in l2["cpu_baseline"] and l2["cpu_baseline"]["value"] == l1["cpu_baseline"]["value"]


@pytest.mark.gpu
@pytest.mark.parametrize("handoff", ["gloo", "rccl"])
def test_bench_rccl_branch_rehearsed_with_one_rank(handoff):
    """VERDICT r4 missing #2: the nccl (= RCCL) branch of bench.py had never executed on hardware.  A single-GPU box cannot run two RCCL
    ranks (one communicator rank per device), but it can run ONE: with ZK_BENCH_FORCE_DIST=1 `torch.distributed.run --nproc-per-node 1
    bench.py --gpus 1` creates the RCCL default group bound to the device and the gloo host group in one process, and goes through the
    barrier, the max-over-ranks all_reduce on a device tensor, the per-rank gather and the Handoff objects of the N > 1 path -- with
    ZK_HANDOFF=rccl the hand-off object is built for the RCCL transport (device tensors on the default group).  The line must be the
    N > 1 form (per_rank, handoff) and the timed proof must hash to the oracle's digest."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ZK_BENCH_FORCE_DIST="1", ZK_HANDOFF=handoff, MASTER_ADDR="127.0.0.1")
    env.pop("ZK_BENCH_BACKEND", None)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1", "--master-port", "29708" if handoff == "rccl" else "29707",
                          os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--inflight", "2", "--log-n", "18",
                          "--no-pcie", "--no-cpu-baseline", "--no-profile", "--no-prove-root", "--no-log23"], capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert out.returncode == 0, out.stderr[-3000:]
    import json
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and [r["rank"] for r in line["per_rank"]] == [0]
    assert "handoff" in line and line["handoff"]["backend"].startswith("RCCL") == (handoff == "rccl")
    assert line["timed_proof_check"]["identical_to_oracle"] in (True, None)
