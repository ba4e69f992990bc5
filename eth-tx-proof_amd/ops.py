"""Host-side mirror of the reference's operator surface (/root/reference/ops/src/lib.rs).

  reference                                   here
  P_STATE: OnceLock<ProverState>              P_STATE / set_prover_state_from_config   (common/src/prover_state/mod.rs:31,59-95)
  TxProof: Operation  (lib.rs:22-62)          TxProof.execute(input) -> AggregatableProof.Txn   = prove_root: prove (the STARK proofs of all
                                              tables, zk_prove_tables) -> per table initial wrapper + shrinking wrappers -> root proof
  AggProof: Monoid    (lib.rs:64-82)          AggProof.combine / .empty  (zk_prove_plonk over the circuits in p_state().recursion)
  BlockProof: Operation (lib.rs:84-101)       BlockProof.execute          (same)

The Plonk half: the PROVER is here (lib.prove_plonk, csrc/prover_plonk.hip); the recursion CIRCUITS (gate sets, recursive
verifier gadgets, witness generators of AllRecursiveCircuits) live in crates absent from the reference tree, so they
are an input: `ProverState(recursion=...)` takes an object implementing RecursionCircuits.  Without one, combine / execute raise
FatalError as the reference does for any prover failure.

`GenerationInputs` upstream is the witness of one transaction; trace generation from it is a CPU EVM interpreter that
is out of scope, so here it carries the already generated traces of the unit's tables.
"""
import threading
import time
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from . import lib as zl
from ._abi import standard_fast_config


class FatalError(RuntimeError):
    """paladin::operation::FatalError -- every prover failure aborts the job (lib.rs:52,72,96: map_err(FatalError::from))."""


@dataclass
class TableInput:
    air: object                 # air.AirProgram
    log_n: int
    trace: object               # numpy (n_cols, n) host array or int device pointer
    public_inputs: np.ndarray


@dataclass
class GenerationInputs:
    tables: List[TableInput]
    txn_number_before: int = 0
    block_number: int = 0
    signed_txn: Optional[bytes] = None


@dataclass
class AggregatableProof:
    kind: str                   # "Txn" | "Agg"
    proof: np.ndarray
    public_values: list = field(default_factory=list)


@dataclass
class GeneratedBlockProof:
    """proof_gen::proof_types::GeneratedBlockProof { b_height, intern } (/root/reference/leader/src/prover.rs:9,36-40)."""
    b_height: int
    intern: np.ndarray


class RecursionCircuits:
    """The circuits of the Plonk half as data + their witness generators (upstream: AllRecursiveCircuits inside ProverState,
    /root/reference/common/src/prover_state/mod.rs:31-57: by_table[t].by_stark_size[degree_bits] = {initial_wrapper,
    shrinking_wrappers}, root, aggregation, block).  Each proving method returns (lib.PlonkState, wires, public_inputs): the circuit's
    prover data (on the caller's device), a (num_wires, n) witness (host array or device pointer) and the public inputs of THIS proof."""

    def shrink_chain(self, table, air, log_n):
        """the circuits for this table at this STARK size, initial wrapper first; raises KeyError when none were built for the size
        (upstream: "Missing preprocessed circuits for {table} table with size {degree_bits}")"""
        raise NotImplementedError

    def initial_wrapper(self, ctx, table, air, log_n, stark_proof_words):
        raise NotImplementedError

    def shrinking_wrapper(self, ctx, table, air, log_n, step, inner_proof):
        raise NotImplementedError

    def root(self, ctx, shrunk_proofs, stark_sizes):
        raise NotImplementedError

    def aggregation(self, ctx, lhs: "AggregatableProof", rhs: "AggregatableProof"):
        raise NotImplementedError

    def block(self, ctx, prev: Optional[np.ndarray], agg: "AggregatableProof"):
        raise NotImplementedError


class ProverState:
    """Per-process immutable prover state: StarkConfig + one zk_ctx per worker thread (+ the recursion circuits, if supplied)."""

    def __init__(self, device=0, config=None, recursion: Optional[RecursionCircuits] = None, stark_only=False, chain_streams=3, chain_overlap="auto",
                 host_transcript=True):
        """stark_only: TxProof.execute stops after the STARK proofs (BASELINE's headline unit) instead of failing for want of
        recursion circuits; never set together with `recursion`.  chain_streams: how many of a transaction's (independent) shrink
        chains one worker proves side by side, each on a context / stream of its own (1: one after another, as upstream's loop; 3 measured
        best on one MI355X: 147 -> 110 ms for one transaction alone, 16.1 -> 20.9 tx/s with eight in flight)."""
        self.device = device
        self.chain_streams = max(1, int(chain_streams))
        # chain_overlap: a table's shrink chain starts when that table's StarkProof is final, under the STARK proofs of the later tables
        # (zk_prove_tables_notify); False: prove(), then the chains (chain_streams of them side by side), upstream's order in time.
        # Measured on one MI355X: alone a transaction takes 77 ms instead of 89, but with 4 / 8 transactions in flight the GPU has no idle
        # issue slots left to hide the chains in and the extra streams cost 10 % (19.4 against 21.7 / 22.6 tx/s).  "auto" (default): overlap
        # while at most `overlap_max_inflight` prove_root calls of this process are running.
        # host_transcript: in the latency regime (the rule of chain_overlap) the sponge runs on a host thread; False keeps it on the device always
        # ("always" -- also what ZK_TRANSCRIPT_HOST=1 in the environment means here -- keeps it on the host in every regime)
        import os as _os
        # ZK_TRANSCRIPT_HOST in the environment decides when set: "0" = never on the host, anything else = always
        env_ht = _os.environ.get("ZK_TRANSCRIPT_HOST", "")
        if env_ht != "":
            self.host_transcript = False if env_ht == "0" else "always"
        else:
            self.host_transcript = "always" if host_transcript == "always" else bool(host_transcript)
        self.chain_overlap = chain_overlap if chain_overlap in (True, False) else "auto"
        self.overlap_max_inflight = 2
        self._active, self._active_lock = 0, threading.Lock()
        self.config = config or standard_fast_config()
        self.recursion = recursion
        self.stark_only = bool(stark_only) and recursion is None
        self._tls = threading.local()
        self._state_ctx, self._state_lock = None, threading.Lock()

    def state_ctx(self):
        """The context that owns process-wide prover data (circuit commitments): never dropped on a worker's device error."""
        with self._state_lock:
            if self._state_ctx is None:
                self._state_ctx = zl.Context(self.device)
            return self._state_ctx

    def ctx(self):
        c = getattr(self._tls, "ctx", None)
        if c is None:
            c = zl.Context(self.device)
            self._tls.ctx = c
        return c

    def side_ctxs(self, k):
        """k more contexts (streams) of the calling worker thread, for proofs that may run side by side with its main stream"""
        lst = getattr(self._tls, "side", None)
        if lst is None:
            lst = self._tls.side = []
        while len(lst) < k:
            lst.append(zl.Context(self.device))
        return lst[:k]

    def drop_ctx(self):
        """Close the calling thread's context (after a device error): the next ctx() call makes a fresh one."""
        for c in getattr(self._tls, "side", None) or []:
            try:
                c.close()
            except Exception:
                pass
        self._tls.side = None
        c = getattr(self._tls, "ctx", None)
        self._tls.ctx = None
        if c is not None:
            try:
                c.close()
            except Exception:
                pass

    def prepare(self, tables):
        """Build the compiled part of the prover state for these table shapes (zk_prover_state_prepare): the reference builds its
        circuits once at start-up (common/src/prover_state/mod.rs:59-95).  Call from the thread that initialised the state."""
        items = [(t.air, t.log_n, t.trace, t.public_inputs) for t in tables]
        zl.prover_state_prepare(self.ctx(), self.config, items)


P_STATE: Optional[ProverState] = None
_lock = threading.Lock()


def set_prover_state_from_config(device=0, config=None, recursion=None, stark_only=False, chain_streams=3, chain_overlap="auto", host_transcript=True):
    """Initialises the global prover state once; a second call leaves the first state in place and returns it
    (the reference only warns on double initialisation, worker/src/main.rs:25-27)."""
    global P_STATE
    with _lock:
        if P_STATE is None:
            P_STATE = ProverState(device, config, recursion, stark_only, chain_streams, chain_overlap, host_transcript)
        return P_STATE


def p_state():
    if P_STATE is None:
        raise RuntimeError("Prover state is not initialized")     # lib.rs:17 `.expect(..)`
    return P_STATE


def split_all_proof(cfg, tables, all_proof):
    """The per-table StarkProof words of an AllProof (zkgl.h proof layout: "ZKGLPRF1", n_tables, then the tables back to back)."""
    out, at = [], 2
    for t in tables:
        n = zl.proof_size_words(cfg, [(t.air, t.log_n, t.trace, t.public_inputs)]) - 2
        out.append(all_proof[at:at + n])
        at += n
    if at != len(all_proof):
        raise FatalError("AllProof of %d words does not split into its %d tables" % (len(all_proof), len(tables)))
    return out


def prove(input: "GenerationInputs") -> np.ndarray:
    """evm_arithmetization::prover::prove: the STARK proofs of all tables of one transaction on one challenger (zk_prove_tables)."""
    st = p_state()
    items = [(t.air, t.log_n, t.trace, t.public_inputs) for t in input.tables]
    return zl.prove_tables(st.ctx(), st.config, items)


def prove_root(input: "GenerationInputs", timing: Optional[dict] = None):
    """AllRecursiveCircuits::prove_root (what generate_txn_proof at /root/reference/ops/src/lib.rs:52 calls): prove(), then for every
    table  by_table[t].by_stark_size[degree_bits].shrink(stark_proof)  = initial_wrapper.prove, then every shrinking wrapper, then
    root.circuit.prove over the shrunk proofs.  Returns (root proof words, public values, all_proof words)."""
    st = p_state()
    rc = st.recursion
    if rc is None:
        raise FatalError("generate_txn_proof: no recursion circuits in the prover state (SURVEY.md 8f-2, A10): pass ProverState(recursion=...)")
    ctx = st.ctx()
    with st._active_lock:
        st._active += 1
        overlap = st.chain_overlap is True or (st.chain_overlap == "auto" and st._active <= st.overlap_max_inflight)
    try:
        return _prove_root(st, rc, ctx, input, timing, overlap)
    finally:
        with st._active_lock:
            st._active -= 1


def _prove_root(st, rc, ctx, input, timing, overlap):
    t0 = time.time()
    ctx.set_side_commit(overlap)          # the same rule for the other latency device: the widest table's trace commitment on a second stream
    # ... and for the third: the Fiat-Shamir sponge on a host thread (zk_ctx_set_host_transcript: a CPU core permutes in ~3 us, the one-wave
    # kernel in 7.5; one transaction alone 75 -> 62 ms, with eight in flight no difference), on every context this call proves on
    host_tr = (overlap and st.host_transcript) or st.host_transcript == "always"
    ctx.set_host_transcript(host_tr)
    try:            # by_table[t].by_stark_size[degree_bits]: a size without circuits is an error whatever else happens
        chains = [rc.shrink_chain(ti, t.air, t.log_n) for ti, t in enumerate(input.tables)]
    except KeyError as e:
        raise FatalError(str(e.args[0])) from e
    shrunk = [None] * len(input.tables)
    collect = timing.get("collect") if timing is not None else None      # a list: receives every chain proof as (table, step, words)

    def shrink(c, ti, t, words, chain):
        """by_table[ti].by_stark_size[log_n].shrink(stark_proof): initial wrapper, then every shrinking wrapper, on context c"""
        c.set_host_transcript(host_tr)
        state, wires, pubs = rc.initial_wrapper(c, ti, t.air, t.log_n, words)
        proof = zl.prove_plonk(state, wires, pubs, ctx=c)
        if collect is not None:
            collect.append((ti, 0, proof))
        for step in range(1, len(chain)):
            state, wires, pubs = rc.shrinking_wrapper(c, ti, t.air, t.log_n, step, proof)
            proof = zl.prove_plonk(state, wires, pubs, ctx=c)
            if collect is not None:
                collect.append((ti, step, proof))
        shrunk[ti] = proof
    k = min(st.chain_streams, len(input.tables))
    if k <= 1 or not overlap:
        # upstream's order: prove(), then the chains -- one after another, or k of them side by side (independent until the root
        # proof), the longest first, each on a stream of its own
        all_proof = prove(input)
        t1 = time.time()
        jobs = [(ti, t, words, chains[ti]) for ti, (t, words) in enumerate(zip(input.tables, split_all_proof(st.config, input.tables, all_proof)))]
        if k <= 1:
            for j in jobs:
                shrink(ctx, *j)
        else:
            order = sorted(jobs, key=lambda j: -sum(1 << b for _, _, b in j[3]))
            lock, errs = threading.Lock(), []

            def run_sorted(c):
                while True:
                    with lock:
                        if not order or errs:
                            return
                        j = order.pop(0)
                    try:
                        shrink(c, *j)
                    except Exception as e:
                        errs.append(e)
            th = [threading.Thread(target=run_sorted, args=(c,)) for c in [ctx] + st.side_ctxs(k - 1)]
            for x in th:
                x.start()
            for x in th:
                x.join()
            if errs:
                raise errs[0]
    else:
        # A table's chain needs that table's StarkProof and nothing else, and the tables are proved one after another on one challenger:
        # k workers (a stream of their own each) start a chain the moment its table's proof is on the host (zk_prove_tables_notify),
        # under the STARK proofs of the tables that follow.  Same values as upstream's sequence, in another order in time.
        import queue
        q, errs = queue.Queue(), []

        def run(c):
            while True:
                j = q.get()
                if j is None:
                    return
                if errs:
                    continue
                try:
                    shrink(c, *j)
                except Exception as e:
                    errs.append(e)
        th = [threading.Thread(target=run, args=(c,)) for c in st.side_ctxs(k)]
        for x in th:
            x.start()
        try:
            items = [(t.air, t.log_n, t.trace, t.public_inputs) for t in input.tables]
            all_proof = zl.prove_tables(ctx, st.config, items,
                                        on_table=lambda ti, words: q.put((ti, input.tables[ti], words, chains[ti])))
            t1 = time.time()
        except BaseException:
            errs.append(None)            # what the workers derived from a refused proof is dropped
            raise
        finally:
            for _ in th:
                q.put(None)
            for x in th:
                x.join()
        if errs:
            raise errs[0]
    t2 = time.time()
    state, wires, pubs = rc.root(ctx, shrunk, [t.log_n for t in input.tables])
    root = zl.prove_plonk(state, wires, pubs, ctx=ctx)
    if timing is not None:
        timing.update(stark_s=t1 - t0, shrink_s=t2 - t1, root_s=time.time() - t2)
    return root, [t.public_inputs for t in input.tables], all_proof


class TxProof:
    """Operation: Input = GenerationInputs, Output = AggregatableProof (lib.rs:25-29)."""

    def execute(self, input: GenerationInputs) -> AggregatableProof:
        ident = input.signed_txn.hex()[:16] if input.signed_txn else \
            "Dummy txn for block %d at %d" % (input.block_number, input.txn_number_before)      # lib.rs:32-46
        st = p_state()
        t0 = time.time()
        try:
            if st.stark_only:       # the STARK half alone (BASELINE's unit), on explicit request only
                proof, pv = prove(input), [t.public_inputs for t in input.tables]
            else:
                proof, pv, _ = prove_root(input)
        except FatalError:
            raise
        except Exception as e:      # generate_txn_proof(..).map_err(FatalError::from)?   lib.rs:52
            raise FatalError(str(e)) from e
        self.last = (ident, time.time() - t0)                                                     # lib.rs:53-58 log line
        return AggregatableProof("Txn", proof, pv)


def _prove_recursive(what, build):
    st = p_state()
    if st.recursion is None:
        raise FatalError("%s: no recursion circuits in the prover state (SURVEY.md 8f-2, A10): pass ProverState(recursion=...); "
                         "zk_prove_plonk proves circuits given as data" % what)
    try:
        ctx = st.ctx()
        state, wires, pubs = build(st.recursion, ctx)
        return zl.prove_plonk(state, wires, pubs, ctx=ctx), np.asarray(pubs, dtype=np.uint64)
    except FatalError:
        raise
    except Exception as e:          # .map_err(FatalError::from)?   lib.rs:72,96
        raise FatalError("%s: %s" % (what, e)) from e


class AggProof:
    """Monoid over AggregatableProof (lib.rs:64-82): combine = generate_agg_proof(p_state(), lhs, rhs) -> Agg."""

    def combine(self, a: AggregatableProof, b: AggregatableProof) -> AggregatableProof:
        proof, pubs = _prove_recursive("generate_agg_proof", lambda rc, ctx: rc.aggregation(ctx, a, b))
        return AggregatableProof("Agg", proof, [pubs])

    def empty(self):
        raise NotImplementedError("empty agg proof")          # lib.rs:78-81 unimplemented!


class BlockProof:
    """Operation: Input = GeneratedAggProof, Output = GeneratedBlockProof (lib.rs:84-101)."""

    def __init__(self, prev=None):
        self.prev = prev

    def execute(self, input: AggregatableProof) -> GeneratedBlockProof:
        proof, pubs = _prove_recursive("generate_block_proof", lambda rc, ctx: rc.block(ctx, self.prev, input))
        height = int(pubs[0]) if len(pubs) else 0
        return GeneratedBlockProof(height, proof)
