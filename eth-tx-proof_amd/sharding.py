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


class HandoffError(RuntimeError):
    """a rank reported a failed map stage (or went silent): the whole job is given up, as paladin aborts on any FatalError"""


_host_group_cache = {}


def cached_host_group(timeout_s=1800):
    """host_group() once per process (creating a group is itself a collective over all ranks)"""
    from datetime import timedelta
    key = (dist.get_world_size(), dist.get_rank())
    if key not in _host_group_cache:
        _host_group_cache[key] = dist.new_group(backend="gloo", timeout=timedelta(seconds=timeout_s))
    return _host_group_cache[key]


class Handoff:
    """The hand-off of finished proofs to rank 0, OUT of the critical path: every worker rank hands each proof to a background sender
    the moment it is done (`put`), rank 0 receives concurrently with its own proving, so that when the last unit finishes only that
    unit's proof is still in flight.  The reference's map stage behaves the same way: a paladin worker returns each result as it
    completes and the leader's fold consumes them (/root/reference/leader/src/prover.rs:26-30).

    Protocol on the host (gloo) group, point to point: header [unit index, words, status] then the words; a rank's last message is
    the marker [-1, 0, ok].  `finish(ok)` ends the stage on every rank: rank 0 waits for all markers, then ONE broadcast tells every
    rank whether any map stage failed -- a rank that hit a FatalError no longer leaves the others blocked in a collective
    (a rank that dies without a marker runs into the group's timeout).  Timings: `exposed_s` = what rank 0 waited in finish() after its
    own last unit (the part of the hand-off that was NOT hidden behind proving), `bytes` = payload received."""

    HDR_TAG, PAY_TAG = 11, 12

    def __init__(self, n_units, group=None, dst=0, mode=None):
        """mode (default: environment ZK_HANDOFF, else "gloo"):
        "gloo"  -- the host (gloo) group, proofs travel as host buffers, rank dst receives from whichever rank has something;
        "rccl"  -- opt-in (north_star's "RCCL only for the final hand-off over xGMI"): the job's DEFAULT process group (nccl =
                   RCCL on GPUs) carries headers and payloads as device tensors, ncclSend / ncclRecv pairs.  RCCL has no receive from
                   "any source", so rank dst asks the open ranks in turn for their next message (a rank's proofs arrive in its own order,
                   ranks run at the same pace: the turn costs nothing while every rank has as many units).  The same ordered protocol
                   runs over a gloo default group with host tensors -- that is how the CPU suite covers it; on hardware it has not run yet."""
        import os
        import queue
        import threading
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.n_units, self.dst = n_units, dst
        self.mode = (mode or os.environ.get("ZK_HANDOFF", "gloo")).lower()
        if self.mode not in ("gloo", "rccl"):
            raise ValueError("ZK_HANDOFF / mode must be gloo or rccl, not %r" % self.mode)
        self.ordered = self.mode == "rccl" and self.world > 1
        self.tdev = None                         # where the tensors of the ordered protocol live: the rank's GPU under nccl, the host under gloo
        if self.ordered:
            group = dist.group.WORLD
            if dist.get_backend(group) == "nccl":
                self.tdev = torch.device("cuda", torch.cuda.current_device())
        self.group = group if (group is not None or self.world == 1) else cached_host_group()
        self.res = [None] * n_units
        self.bytes, self.exposed_s, self.error = 0, 0.0, None
        self._status = {}                        # rank dst: rank -> ok, from the ranks' closing markers
        self._q, self._th = queue.Queue(), None
        if self.world > 1:
            self._th = threading.Thread(target=self._recv_loop if self.rank == dst else self._send_loop, daemon=True)
            self._th.start()

    # tensors of the wire: host tensors on the gloo group, device tensors on the RCCL group (tags are a gloo notion: RCCL pairs sends and
    # receives of two ranks in program order, which the ordered protocol guarantees)
    def _t(self, x):
        return x if self.tdev is None else x.to(self.tdev)

    def _send(self, t, tag):
        if self.ordered:
            dist.send(self._t(t), self.dst, group=self.group)
        else:
            dist.send(t, self.dst, group=self.group, tag=tag)

    def put(self, unit, words):
        """a finished proof (np.uint64 words) of unit `unit`; returns at once"""
        if self.world == 1 or self.rank == self.dst:
            self.res[unit] = np.array(words, dtype=np.uint64, copy=True)
        else:
            self._q.put((unit, np.array(words, dtype=np.uint64, copy=True)))

    def _send_loop(self):
        if self.tdev is not None:
            torch.cuda.set_device(self.tdev)
        owed = None                              # words of a payload whose header went out: rank dst is (or will be) inside recv for exactly that many
        try:
            while True:
                unit, w = self._q.get()          # (unit, words) or the marker (None, ok)
                if unit is None:
                    self._send(torch.tensor([-1, 0, 1 if w else 0], dtype=torch.int64), self.HDR_TAG)
                    return
                self._send(torch.tensor([unit, w.size, 1], dtype=torch.int64), self.HDR_TAG)
                owed = w.size
                self._send(torch.from_numpy(w.view(np.int64)), self.PAY_TAG)
                owed = None
        except Exception as e:          # surfaced by finish()
            self.error = e
            try:
                # rank dst may be blocked in recv(PAY_TAG) for the payload whose header it has: give it that many words (zeros;
                # the failure marker that follows voids the job), THEN the marker "this rank is out" that it counts
                if owed is not None:
                    self._send(torch.zeros(owed, dtype=torch.int64), self.PAY_TAG)
                self._send(torch.tensor([-1, 0, 0], dtype=torch.int64), self.HDR_TAG)
            except Exception:
                pass    # transport gone: rank dst runs into the group's timeout (cached_host_group(timeout_s))

    def _recv_loop_ordered(self):
        """rank dst under mode "rccl": the open ranks are asked in turn for their next message (header, then payload, from THAT rank)"""
        try:
            if self.tdev is not None:
                torch.cuda.set_device(self.tdev)
            status = self._status
            open_ranks = [r for r in range(self.world) if r != self.dst]
            while open_ranks:
                for r in list(open_ranks):
                    h = self._t(torch.zeros(3, dtype=torch.int64))
                    dist.recv(h, src=r, group=self.group)
                    unit, n, ok = (int(v) for v in h.cpu())
                    if unit < 0:
                        status[r] = ok
                        open_ranks.remove(r)
                        continue
                    buf = self._t(torch.empty(n, dtype=torch.int64))
                    dist.recv(buf, src=r, group=self.group)
                    self.res[unit] = buf.cpu().numpy().view(np.uint64)
                    self.bytes += 8 * n
        except Exception as e:
            self.error = e

    def _recv_loop(self):
        if self.ordered:
            return self._recv_loop_ordered()
        try:
            status, open_ranks = self._status, self.world - 1          # filled as markers arrive: a failing receiver still knows who reported what
            h = torch.zeros(3, dtype=torch.int64)
            while open_ranks:
                r = dist.recv(h, group=self.group, tag=self.HDR_TAG)          # from whichever rank has something (gloo: any source)
                unit, n, ok = int(h[0]), int(h[1]), int(h[2])
                if unit < 0:
                    status[r] = ok
                    open_ranks -= 1
                    continue
                buf = torch.empty(n, dtype=torch.int64)
                dist.recv(buf, src=r, group=self.group, tag=self.PAY_TAG)
                self.res[unit] = buf.numpy().view(np.uint64)
                self.bytes += 8 * n
        except Exception as e:
            self.error = e

    def finish(self, ok=True):
        """Ends the stage.  Rank dst: the list of all proofs in unit order; other ranks: None.  Raises HandoffError on every rank when
        any rank finished with ok=False."""
        import time as _t
        if self.world == 1:
            if not ok:
                raise HandoffError("map stage failed")
            return self.res
        t0 = _t.perf_counter()
        if self.rank != self.dst:
            self._q.put((None, bool(ok)))
        self._th.join()
        self.exposed_s = _t.perf_counter() - t0
        # A local transport error does NOT skip the status broadcast: the other ranks are (or will be) waiting in it.  Rank dst with
        # a failed receiver broadcasts 0; a worker with a failed sender has told rank dst so (marker with ok = 0) and still listens.
        flag = torch.tensor([1], dtype=torch.int64)
        if self.rank == self.dst:
            status = self._status
            bad = [r for r, s in status.items() if not s] + ([self.dst] if (not ok or self.error is not None) else [])
            missing = [u for u in range(self.n_units) if self.res[u] is None] if not bad else []
            flag[0] = 0 if (bad or missing) else 1
        try:
            flag = self._t(flag)
            dist.broadcast(flag, src=self.dst, group=self.group)
            flag = flag.cpu()
        except Exception as e:
            raise HandoffError("proof hand-off failed: %s (status broadcast: %s)" % (self.error, e))
        if self.error is not None:
            raise HandoffError("proof hand-off failed: %s" % self.error)
        if int(flag[0]) != 1:
            raise HandoffError("the map stage failed on a rank: every rank gives the job up")
        return self.res if self.rank == self.dst else None
