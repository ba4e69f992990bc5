"""The caller of the operators: `ProverInput::prove` (/root/reference/leader/src/prover.rs:14-43):

    IndexedStream::from(proof_gen_ir).map(&TxProof).fold(&AggProof)  ->  Literal(agg).map(&BlockProof { prev })

Paladin's distributed runtime (AMQP, worker processes) is control plane and out of scope; `Runtime` here is the
in-process equivalent of `Runtime::in_memory` (leader/src/main.rs): N worker threads, each with its own zk_ctx / HIP stream, one
queue of units.  With torch.distributed initialised, the map stage is sharded over ranks (unit i -> rank i mod W, sharding.py) and
the proofs are gathered on rank 0, which runs the fold and the block proof.

Failure handling: the reference turns every prover error into FatalError and Paladin aborts the job; the
same here, except that a unit that failed with a DEVICE error (ZK_ERR_HIP / ZK_ERR_OOM: the card, not the input) goes back
on the queue once -- the worker that hit the error drops its context first -- before the job is given up."""
import queue
import threading
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from . import lib as zl
from . import ops, sharding

DEVICE_ERRORS = (-2, -3)        # ZK_ERR_OOM, ZK_ERR_HIP (include/zkgl.h)


class Runtime:
    """num_workers threads over one P_STATE (paladin `--num-workers N`)."""

    def __init__(self, num_workers=4, max_requeue=1):
        self.num_workers, self.max_requeue = max(1, num_workers), max_requeue
        self.requeued = []          # (unit index, error text) of every re-queued unit, for the log

    def map(self, op, inputs, indices=None, on_result=None):
        """Runs op.execute over `inputs` on the worker threads; results in input order.  Raises the first FatalError.
        on_result(index, result): called from the worker thread as soon as a unit is done (the proof hand-off starts there)."""
        idx = list(range(len(inputs))) if indices is None else list(indices)
        q = queue.Queue()
        for k, i in enumerate(idx):
            q.put((k, i, 0))
        res, errors, lock = [None] * len(inputs), [], threading.Lock()

        def worker(wid):
            while not errors:
                try:
                    k, i, tries = q.get_nowait()
                except queue.Empty:
                    return
                try:
                    res[k] = op.execute(inputs[k])
                    if on_result is not None:
                        on_result(i, res[k])
                except ops.FatalError as e:
                    code = getattr(e.__cause__, "code", None)
                    # a unit whose traces live in the failed context's device memory cannot go to another context: only units with
                    # host-resident inputs are re-queued
                    movable = all(not isinstance(t.trace, (int, np.integer)) for t in getattr(inputs[k], "tables", []))
                    if code in DEVICE_ERRORS and tries < self.max_requeue and movable:
                        with lock:
                            self.requeued.append((i, str(e)))
                        ops.p_state().drop_ctx()        # this worker starts over on a fresh context / stream / allocator
                        q.put((k, i, tries + 1))        # whichever worker is free next takes the unit
                    else:
                        errors.append(e)
        th = [threading.Thread(target=worker, args=(w,)) for w in range(min(self.num_workers, max(1, len(inputs))))]
        for t in th:
            t.start()
        for t in th:
            t.join()
        if errors:
            raise errors[0]
        return res

    @staticmethod
    def fold(monoid, items):
        """Balanced pairwise fold (combine is associative: a Monoid); one item folds to itself, none to monoid.empty()."""
        items = list(items)
        if not items:
            return monoid.empty()
        while len(items) > 1:
            nxt = [monoid.combine(items[j], items[j + 1]) for j in range(0, len(items) - 1, 2)]
            if len(items) & 1:
                nxt.append(items[-1])
            items = nxt
        return items[0]


@dataclass
class ProverInput:
    proof_gen_ir: List[ops.GenerationInputs] = field(default_factory=list)

    def prove(self, runtime: Optional[Runtime] = None, previous: Optional[np.ndarray] = None, device=None, group=None):
        """-> GeneratedBlockProof on rank 0 (None on the other ranks).  prover.rs:20-43."""
        import torch.distributed as dist
        runtime = runtime or Runtime()
        n = len(self.proof_gen_ir)
        world = dist.get_world_size() if dist.is_initialized() else 1
        rank = dist.get_rank() if dist.is_initialized() else 0
        mine = sharding.shard_units(n, world, rank)
        if world > 1:
            # every finished proof goes to rank 0 while the next unit is being proved; a rank whose map stage fails tells the others
            # (sharding.Handoff.finish), so nobody is left waiting in a collective.  group: a HOST (gloo) group -- finished proofs
            # are host buffers; None = sharding.cached_host_group(), whatever the default group's backend is
            ho = sharding.Handoff(n, group=group)
            failed = None
            try:
                runtime.map(ops.TxProof(), [self.proof_gen_ir[i] for i in mine], indices=mine,                 # .map(&TxProof)
                            on_result=lambda i, p: ho.put(i, p.proof))
            except Exception as e:      # FatalError or anything else: the other ranks must still hear about it
                failed = e
            try:
                words = ho.finish(ok=failed is None)
            except sharding.HandoffError as e:
                raise (failed or ops.FatalError(str(e))) from e
            if failed is not None:      # finish(ok=False) raises on every rank; kept for a hand-off that says otherwise
                raise failed
            self.handoff = {"exposed_s": ho.exposed_s, "bytes": ho.bytes}
            if rank != 0:
                return None
            # public values travel with the inputs (every rank holds proof_gen_ir), the proof words with the hand-off
            txs = [ops.AggregatableProof("Txn", w, [t.public_inputs for t in self.proof_gen_ir[i].tables]) for i, w in enumerate(words)]
        else:
            txs = runtime.map(ops.TxProof(), [self.proof_gen_ir[i] for i in mine], indices=mine)              # .map(&TxProof)
        agg = Runtime.fold(ops.AggProof(), txs)                                                    # .fold(&AggProof)
        if agg.kind != "Agg":
            raise RuntimeError("AggProof is is not GeneratedAggProof")                              # prover.rs:41 (bail!)
        return ops.BlockProof(prev=previous).execute(agg)                                            # Literal(proof).map(&BlockProof)
