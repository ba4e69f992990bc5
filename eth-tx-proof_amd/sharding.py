"""Unit sharding across ranks and the final proof hand-off.

Units (= TxProof inputs) are independent (`IndexedStream::from(ir).map(&TxProof)`, /root/reference/leader/src/prover.rs:26-27),
so rank r proves units r, r+W, r+2W, ...; there is no data-path collective.  Finished proofs are gathered on rank 0,
where the reference's `fold(&AggProof)` consumes them (prover.rs:28).  Backend: nccl (= RCCL) on GPUs, gloo on CPU.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_units(n_units, world, rank):
    return list(range(rank, n_units, world))


def host_group():
    """A gloo group over all ranks for the proof hand-off: finished proofs sit in HOST memory (zk_prove_tables downloads them), so
    they travel between the rank processes as host buffers; RCCL (the default group on GPUs) carries only the job's timing
    collectives -- the data path itself has no collective (units are independent)."""
    return dist.new_group(backend="gloo")


def gather_proofs(local_proofs, n_units, device=None, dst=0, group=None):
    """local_proofs: {unit_index: np.uint64 array}; lengths may differ between units (different table shapes).  Returns the
    list of all proofs in unit order on rank `dst`, None elsewhere.  One collective on fixed-size rows: word 0 of a row is the
    proof's length, the row is padded to the longest proof of the job (one all_reduce MAX finds it).  `group`: the process group
    to use (host_group() for host buffers); `device`: where the tensors of that group must live (None = CPU)."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    if world == 1:
        return [local_proofs[i] for i in range(n_units)]
    per = (n_units + world - 1) // world
    plen = max((len(p) for p in local_proofs.values()), default=0)
    ln = torch.tensor([plen], dtype=torch.int64, device=device)
    dist.all_reduce(ln, op=dist.ReduceOp.MAX, group=group)
    plen = int(ln.item())
    buf = np.zeros((per, 1 + plen), dtype=np.uint64)
    for k, u in enumerate(shard_units(n_units, world, rank)):
        p = local_proofs[u]
        buf[k, 0] = len(p)
        buf[k, 1:1 + len(p)] = p
    mine = torch.from_numpy(buf.view(np.int64))
    if device is not None:
        mine = mine.to(device)
    out = [torch.empty_like(mine) for _ in range(world)] if rank == dst else None
    dist.gather(mine, out, dst=dst, group=group)
    if rank != dst:
        return None
    res = [None] * n_units
    for r in range(world):
        arr = out[r].cpu().numpy().view(np.uint64)
        for k, u in enumerate(shard_units(n_units, world, r)):
            res[u] = arr[k, 1:1 + int(arr[k, 0])].copy()
    return res
