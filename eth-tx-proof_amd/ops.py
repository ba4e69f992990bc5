"""Host-side mirror of the reference's operator surface (/root/reference/ops/src/lib.rs).

  reference                                   here
  P_STATE: OnceLock<ProverState>              P_STATE / set_prover_state_from_config   (common/src/prover_state/mod.rs:31,59-95)
  TxProof: Operation  (lib.rs:22-62)          TxProof.execute(input) -> AggregatableProof.Txn
  AggProof: Monoid    (lib.rs:64-82)          AggProof.combine / .empty  (recursion not built: SURVEY 8f-2 -> FatalError)
  BlockProof: Operation (lib.rs:84-101)       BlockProof.execute          (same)

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


class ProverState:
    """Per-process immutable prover state: StarkConfig + one zk_ctx per worker thread."""

    def __init__(self, device=0, config=None):
        self.device = device
        self.config = config or standard_fast_config()
        self._tls = threading.local()

    def ctx(self):
        c = getattr(self._tls, "ctx", None)
        if c is None:
            c = zl.Context(self.device)
            self._tls.ctx = c
        return c

    def prepare(self, tables):
        """Build the compiled part of the prover state for these table shapes (zk_prover_state_prepare): the reference builds its
        circuits once at start-up (common/src/prover_state/mod.rs:59-95).  Call from the thread that initialised the state."""
        items = [(t.air, t.log_n, t.trace, t.public_inputs) for t in tables]
        zl.prover_state_prepare(self.ctx(), self.config, items)


P_STATE: Optional[ProverState] = None
_lock = threading.Lock()


def set_prover_state_from_config(device=0, config=None):
    """Initialises the global prover state once; a second call leaves the first state in place and returns it
    (the reference only warns on double initialisation, worker/src/main.rs:25-27)."""
    global P_STATE
    with _lock:
        if P_STATE is None:
            P_STATE = ProverState(device, config)
        return P_STATE


def p_state():
    if P_STATE is None:
        raise RuntimeError("Prover state is not initialized")     # lib.rs:17 `.expect(..)`
    return P_STATE


class TxProof:
    """Operation: Input = GenerationInputs, Output = AggregatableProof (lib.rs:25-29)."""

    def execute(self, input: GenerationInputs) -> AggregatableProof:
        ident = input.signed_txn.hex()[:16] if input.signed_txn else \
            "Dummy txn for block %d at %d" % (input.block_number, input.txn_number_before)      # lib.rs:32-46
        st = p_state()
        t0 = time.time()
        try:
            items = [(t.air, t.log_n, t.trace, t.public_inputs) for t in input.tables]
            proof = zl.prove_tables(st.ctx(), st.config, items)
        except Exception as e:      # generate_txn_proof(..).map_err(FatalError::from)?   lib.rs:52
            raise FatalError(str(e)) from e
        self.last = (ident, time.time() - t0)                                                     # lib.rs:53-58 log line
        return AggregatableProof("Txn", proof, [t.public_inputs for t in input.tables])


class AggProof:
    """Monoid over AggregatableProof (lib.rs:67-82): Plonk recursion, not built yet."""

    def combine(self, a, b):
        # the Plonk PROVER of the recursion half exists (lib.prove_plonk, bit-exact against its oracle on circuits given as data);
        # the aggregation CIRCUIT (gate set, recursive verifier gadgets, witness generator) lives in crates that are absent from the
        # reference tree, so there is nothing honest to prove here
        raise FatalError("generate_agg_proof: the aggregation circuit is not available (SURVEY.md 8f-2, A10); zk_prove_plonk proves "
                         "circuits given as data")

    def empty(self):
        raise NotImplementedError("empty agg proof")          # lib.rs:78-81 unimplemented!


class BlockProof:
    def __init__(self, prev=None):
        self.prev = prev

    def execute(self, input):
        raise FatalError("generate_block_proof: the block circuit is not available (SURVEY.md 8f-2, A10); zk_prove_plonk proves "
                         "circuits given as data")
