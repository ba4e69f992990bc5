"""Plonk-shaped circuits as plain data (include/zkgl.h: zk_plonk_circuit) and a synthetic family with satisfying witnesses.

The reference proves its recursion circuits (shrink / root / aggregation / block: /root/reference/ops/src/lib.rs:52,72,95) with
plonky2's CircuitConfig::standard_recursion_config(): 135 wires, 80 routed, quotient degree factor 8.  Those circuits' gate
sets and witness generators live in crates that are absent from the reference tree, so -- like the tables' AIRs --
a circuit here is data: constant/selector columns, the routed-wire permutation as sigma values, and a straight-line gate program.
`synthetic_circuit` builds one with the recursion config's shape and a witness that satisfies it:
  * copy constraints: routed columns 2j and 2j+1 are tied cell by cell (2-cycles), columns of every fourth pair additionally
    tie row r to row r^1 (4-cycles), the rest of the permutation is the identity;
  * gates (all filtered by a 0/1 selector column, so the degree goes up by one): multiply-add  c0*a*b + c1*c - d  on routed
    wires with the result in an advice wire, and the degree-7 s-box  a^7 - e  (e an advice wire): degree 8 with the selector.
"""
import ctypes as C
import os
import time

import numpy as np

from ._abi import Air, HelperDesc, Instr, OP_ADD, OP_EMIT, OP_MUL, OP_SUB, OPND_CONST, OPND_LOCAL, OPND_PUBLIC, OPND_REG, opnd, u32, u64
from .air import AirProgram

P = 0xFFFFFFFF00000001
GEN = 14293326489335486720


class PlonkCircuit(C.Structure):
    """zk_plonk_circuit (include/zkgl.h)"""
    _fields_ = [("degree_bits", u32), ("num_wires", u32), ("num_routed_wires", u32), ("num_constants", u32),
                ("quotient_degree_factor", u32), ("n_domain_separator", u32), ("gates", C.c_void_p), ("constants", C.c_void_p),
                ("sigmas", C.c_void_p), ("domain_separator", C.c_void_p)]


class Circuit:
    """Owns the arrays behind a zk_plonk_circuit; `.ref` is a pointer to the struct."""

    def __init__(self, degree_bits, num_wires, num_routed, constants, sigmas, gates, qdf=8, domain_separator=None):
        self.degree_bits, self.num_wires, self.num_routed, self.qdf = degree_bits, num_wires, num_routed, qdf
        self.constants = np.ascontiguousarray(constants, dtype=np.uint64)
        self.sigmas = np.ascontiguousarray(sigmas, dtype=np.uint64)
        self.num_constants = self.constants.shape[0]
        self.gates = gates
        self.domain_separator = np.ascontiguousarray([] if domain_separator is None else domain_separator, dtype=np.uint64)
        self.c = PlonkCircuit(degree_bits, num_wires, num_routed, self.num_constants, qdf, self.domain_separator.size,
                              C.cast(gates.ref, C.c_void_p), self.constants.ctypes.data, self.sigmas.ctypes.data,
                              self.domain_separator.ctypes.data if self.domain_separator.size else None)
        self.ref = C.pointer(self.c)


def _pow(b, e):
    return pow(int(b), int(e), P)


def _root(k):
    return _pow(7277203076849721926, 1 << (32 - k))


def synthetic_circuit(degree_bits=12, num_wires=135, num_routed=80, n_muladd=16, n_sbox=8, seed=1, public_input_gate=False):
    """Returns (Circuit, wires) with wires a (num_wires, n) uint64 witness satisfying gates and copy constraints.
    public_input_gate: adds upstream's PublicInputGate -- four advice wires whose row-0 cells must equal the four words of
    public_inputs_hash (PUBLIC operands 0..3), filtered by a selector column that is 1 in row 0.  The witness then depends on the public
    inputs: `bind_public_inputs(circuit, wires, public_inputs_hash)` writes the four cells (Circuit.pi_wires names the columns)."""
    n = 1 << degree_bits
    rng = np.random.default_rng(seed)

    def rand(shape):
        a = rng.integers(0, 1 << 63, size=shape, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=shape, dtype=np.uint64)
        return np.where(a >= np.uint64(P), a - np.uint64(P), a)
    n_pi = 4 if public_input_gate else 0
    assert num_routed % 8 == 0 and num_routed <= num_wires and num_wires - num_routed >= n_muladd + n_sbox + n_pi
    # ---- permutation: cell (j, r) -> (j2, r2); sigma_j(w^r) = k_{j2} * w^{r2}, k_j = g^j
    col = np.tile(np.arange(num_routed, dtype=np.int64)[:, None], (1, n))
    row = np.tile(np.arange(n, dtype=np.int64)[None, :], (num_routed, 1))
    tcol, trow = col.copy(), row.copy()
    for j in range(0, num_routed, 2):
        if (j // 2) % 4 == 0:      # 4-cycles: (j, r) -> (j+1, r) -> (j, r^1) -> (j+1, r^1) -> (j, r)
            tcol[j], trow[j] = j + 1, row[j]
            tcol[j + 1], trow[j + 1] = j, row[j + 1] ^ 1
        elif (j // 2) % 4 in (1, 2):   # 2-cycles within the row
            tcol[j], tcol[j + 1] = j + 1, j
        # else: identity
    ks = [_pow(GEN, j) for j in range(num_routed)]
    w = _root(degree_bits)
    wpow = np.empty(n, dtype=object)
    x = 1
    for r in range(n):
        wpow[r] = x
        x = x * w % P
    sig = np.empty((num_routed, n), dtype=np.uint64)
    for j in range(num_routed):
        kk = np.array([ks[c] for c in tcol[j]], dtype=object)
        sig[j] = np.array([(int(a) * int(b)) % P for a, b in zip(kk, wpow[trow[j]])], dtype=np.uint64)
    # ---- witness: routed wires constant on the cycles
    wires = rand((num_wires, n))
    for j in range(0, num_routed, 2):
        kind = (j // 2) % 4
        if kind == 0:
            base = wires[j].copy()
            base[1::2] = base[0::2]
            wires[j] = base; wires[j + 1] = base
        elif kind in (1, 2):
            wires[j + 1] = wires[j]
    # ---- constants: c0, c1 (arbitrary), selector (0/1)
    consts = rand((3 + (1 if public_input_gate else 0), n))
    consts[2] = rng.integers(0, 2, size=n, dtype=np.uint64)
    if public_input_gate:
        consts[3] = 0
        consts[3][0] = 1
    W = num_wires
    Lc = lambda i: opnd(OPND_LOCAL, i)
    R = lambda i: opnd(OPND_REG, i)
    ins = []
    adv = num_routed

    def obj(a):
        return a.astype(object)
    for m in range(n_muladd):
        a, b, c_ = (5 * m) % num_routed, (5 * m + 2) % num_routed, (5 * m + 3) % num_routed
        d = adv; adv += 1
        wires[d] = np.array([(int(c0) * int(x) % P * int(y) + int(c1) * int(z)) % P
                             for c0, c1, x, y, z in zip(consts[0], consts[1], wires[a], wires[b], wires[c_])], dtype=np.uint64)
        ins += [(OP_MUL, 0, Lc(a), Lc(b)), (OP_MUL, 0, R(0), Lc(W + 0)), (OP_MUL, 1, Lc(c_), Lc(W + 1)), (OP_ADD, 0, R(0), R(1)),
                (OP_SUB, 0, R(0), Lc(d)), (OP_MUL, 0, R(0), Lc(W + 2)), (OP_EMIT, 0, R(0), 0)]
    for m in range(n_sbox):
        a = (7 * m + 1) % num_routed
        e = adv; adv += 1
        wires[e] = np.array([pow(int(x), 7, P) for x in wires[a]], dtype=np.uint64)
        ins += [(OP_MUL, 0, Lc(a), Lc(a)), (OP_MUL, 1, R(0), Lc(a)), (OP_MUL, 0, R(0), R(0)), (OP_MUL, 0, R(0), R(1)),   # a^2, a^3, a^4, a^7
                (OP_SUB, 0, R(0), Lc(e)), (OP_MUL, 0, R(0), Lc(W + 2)), (OP_EMIT, 0, R(0), 0)]
    pi_wires = []
    for k in range(n_pi):       # PublicInputGate: sel_pi * (wire - public_inputs_hash[k])
        c_ = adv; adv += 1
        pi_wires.append(c_)
        ins += [(OP_SUB, 0, Lc(c_), opnd(OPND_PUBLIC, k)), (OP_MUL, 0, R(0), Lc(W + 3)), (OP_EMIT, 0, R(0), 0)]
    gates = AirProgram(W + consts.shape[0], 0, [], [1], ins, 2)
    circ = Circuit(degree_bits, num_wires, num_routed, consts, sig, gates)
    circ.pi_wires = pi_wires
    return circ, np.ascontiguousarray(wires, dtype=np.uint64)


def bind_public_inputs(circuit, wires, public_inputs_hash):
    """The witness of a circuit with a PublicInputGate for THESE public inputs: a copy of `wires` with the gate's four row-0 cells set
    to the words of public_inputs_hash = hash_no_pad(public_inputs)."""
    w = np.array(wires, dtype=np.uint64, copy=True)
    for k, col in enumerate(circuit.pi_wires):
        w[col, 0] = np.uint64(int(public_inputs_hash[k]))
    return w


class SyntheticRecursion:
    """STAND-IN for AllRecursiveCircuits (ops.RecursionCircuits; upstream: evm_arithmetization::fixed_recursive_verifier, held by the
    prover state, /root/reference/common/src/prover_state/mod.rs:31-57) so that the reference's whole flow

        TxProof = prove_root:  7 STARK proofs -> per table initial_wrapper + shrinking_wrappers (until degree_bits <= 13) -> root
        map(&TxProof) -> fold(&AggProof) -> BlockProof                                   (/root/reference/leader/src/prover.rs:26-36)

    runs end to end on the GPU Plonk prover.  NOT recursion: every circuit is a `recursion_circuit` -- the gate set, selector
    polynomials, gate mix and routing of plonky2's recursion circuits (gates.py) at the SIZE the real circuit is estimated to have --
    and verifies nothing in-circuit.  A child is bound to its parent through the parent's public inputs: the Poseidon Merkle roots
    (`digest`, on the GPU) of the children's proof words, which the circuit's PublicInputGate constrains (its wires carry
    public_inputs_hash).  The real circuits are data a maintainer supplies the same way.

    Sizes (degree bits) -- estimates of the real circuits' row counts, NOT taken from the reference (absent):
      initial wrapper of a table  = a circuit verifying one STARK proof: 84 queries x (leaf hashing ceil(C/8) + ceil(A/8) rows, three
                                    Merkle paths, the FRI layers' leaves / paths / interpolation, reducing 2C + 2A + Q openings with alpha)
                                    + hashing the openings + the constraints at zeta           (`initial_wrapper_degree_bits`)
      shrinking wrapper           = a circuit verifying one recursion-config proof: 2^12 rows (plonky2's own figure for its recursion
                                    circuit); appended while the previous circuit has more than 2^threshold rows (upstream: 13)
      root 2^15 (seven 2^12-row verifiers and the public-value plumbing), aggregation 2^14 (two verifiers), block 2^14.
    size_shift moves every size (tests run the same flow on small circuits: size_shift=-6, threshold=7)."""

    ROOT_BITS, AGG_BITS, BLOCK_BITS, SHRINK_BITS, THRESHOLD_DEGREE_BITS = 15, 14, 14, 12, 13

    def __init__(self, cfg=None, degree_bits=None, seed=1, size_shift=0, threshold=None, degree_bits_ranges=None, stark_cfg=None, witness_standin=None):
        """degree_bits: (legacy) fixed size of the aggregation / block circuits.  degree_bits_ranges: per table (lo, hi) half-open
        ranges of STARK sizes circuits exist for, like the reference's *_CIRCUIT_SIZE settings
        (/root/reference/common/src/prover_state/circuit.rs:76-100); None = every size."""
        import threading
        from ._abi import standard_fast_config, standard_recursion_config
        self.cfg = cfg or standard_recursion_config()
        self.stark_cfg = stark_cfg or standard_fast_config()
        self.seed, self.shift = seed, size_shift
        # per-proof witness generation on the CPU (stand-in, generate_witness_standin): ZK_WITNESS_STANDIN=0 / witness_standin=False keeps round 4's
        # behaviour (one precomputed witness per circuit, no generator work inside the transaction)
        self.witness_standin = os.environ.get("ZK_WITNESS_STANDIN", "1") != "0" if witness_standin is None else bool(witness_standin)
        self.witness_standin_seconds, self.witness_standin_rows = 0.0, 0
        self._standin_lock = threading.Lock()
        self.threshold = (self.THRESHOLD_DEGREE_BITS + size_shift) if threshold is None else threshold
        self.ranges = degree_bits_ranges
        self.fixed_bits = degree_bits
        self._states, self._lock, self._pinned = {}, threading.Lock(), {}

    # ---- sizes
    def _bits(self, b):
        return max(3, b + self.shift)

    def initial_wrapper_degree_bits(self, n_cols, n_aux, log_n):
        c = self.stark_cfg
        q = c.num_challenges << c.rate_bits
        lg = log_n + c.rate_bits
        layers, d = [], log_n
        while d > c.fri_final_poly_bits and d + c.rate_bits - c.fri_arity_bits >= c.cap_height:
            d -= c.fri_arity_bits
            layers.append(d + c.rate_bits)
        per_query = -(-n_cols // 8) + -(-n_aux // 8) + 3 * (lg - c.cap_height) + sum(4 + (l - c.cap_height) + 2 for l in layers) \
            + -(-(2 * n_cols + 2 * n_aux + q) // 43)
        rows = c.num_query_rounds * per_query + (4 * n_cols + 4 * n_aux + 2 * q) // 8 + (n_cols + n_aux) // 2 + 64
        return self._bits(max(self.SHRINK_BITS, (rows - 1).bit_length()))

    def chain_degree_bits(self, n_cols, n_aux, log_n):
        """[initial wrapper, shrinking wrappers ..]: RecursiveCircuitsForTableSize::new shrinks until degree_bits <= threshold"""
        bits = [self.initial_wrapper_degree_bits(n_cols, n_aux, log_n)]
        while bits[-1] > self.threshold:
            bits.append(self._bits(self.SHRINK_BITS))
            if bits[-1] >= bits[-2]:
                break
        return bits

    # ---- circuits and their prover data
    def circuit(self, kind, key=0, degree_bits=None):
        """(Circuit, wires) of one stand-in circuit; kind in initial / shrink / root / agg / block, key = (table, stark size, step)"""
        if degree_bits is None:
            degree_bits = {"root": self._bits(self.ROOT_BITS), "agg": self.fixed_bits or self._bits(self.AGG_BITS),
                           "block": self.fixed_bits or self._bits(self.BLOCK_BITS)}[kind]
        import zlib
        sd = zlib.crc32(repr((kind, key)).encode()) & 0xFFFF
        return recursion_circuit(degree_bits=degree_bits, seed=self.seed * 65536 + sd)

    @property
    def circuits(self):
        return {"agg": self.circuit("agg"), "block": self.circuit("block")}

    def state(self, ctx, kind, key=0, degree_bits=None):
        """The circuit's prover data, built once per device (upstream: ProverState holds the circuits for the process; every worker
        proves with the same data).  Built on the prover state's own context when there is one, so that a worker's context can be
        dropped after a device error without taking the circuits with it."""
        from . import lib as zl
        k = (ctx.device, kind, key, degree_bits)
        with self._lock:
            if k not in self._states:
                from . import ops
                owner = ops.P_STATE.state_ctx() if ops.P_STATE is not None and ops.P_STATE.device == ctx.device else ctx
                self._states[k] = zl.PlonkState(owner, self.cfg, self.circuit(kind, key, degree_bits)[0])
            return self._states[k]

    @staticmethod
    def digest(ctx, words):
        """The digest that binds a child proof into its parent's public inputs: the Merkle cap of height 0 (the root) over the proof
        words in leaves of eight, zero-padded to a power of two of leaves -- one zk_merkle_build on the GPU.  (A sponge over the whole
        proof is a serial chain of words / 8 permutations: seconds for a wide table's StarkProof.)"""
        w = np.ascontiguousarray(words, dtype=np.uint64).ravel()
        n_leaves = max(2, 1 << (max(1, -(-w.size // 8)) - 1).bit_length())
        leaves = np.zeros((n_leaves, 8), dtype=np.uint64)
        leaves.ravel()[:w.size] = w
        return ctx.merkle_build(leaves, 0, want_digests=False)[0][0]

    @staticmethod
    def public_inputs_hash(ctx, pubs):
        """hash_no_pad(public inputs): always the sponge (hash_rows copies rows of <= 4 words: pad to the sponge by hand)"""
        pubs = np.ascontiguousarray(pubs, dtype=np.uint64)
        if pubs.size > 4:
            return ctx.hash_rows(pubs[None, :])[0]
        st = np.zeros(12, dtype=np.uint64)
        st[:pubs.size] = pubs
        return ctx.poseidon(st[None, :])[0][:4]

    def witness(self, ctx, kind, key, degree_bits, pih):
        """The circuit's witness for THIS proof: the proving context's own copy of the stand-in witness in pinned host memory (one
        asynchronous upload at PCIe rate inside zk_prove_plonk, as for a trace from zk_host_alloc), PublicInputGate cells rewritten
        in place.  Upstream generates the witness of every proof on the CPU (plonky2::iop::generator) -- not reproduced here."""
        from . import lib as zl, ops
        circ, wires = self.circuit(kind, key, degree_bits)
        k = (id(ctx), kind, key, degree_bits)                  # one copy per context: a context serves one proof at a time
        ent = self._pinned.get(k)
        buf = ent[1] if ent is not None and ent[0] is ctx else None
        if buf is None:
            owner = ops.P_STATE.state_ctx() if ops.P_STATE is not None and ops.P_STATE.device == ctx.device else ctx
            buf = zl.pinned_empty(owner, wires.shape)
            buf[...] = wires
            self._pinned[k] = (ctx, buf)
        for i, col in enumerate(circ.pi_wires):
            buf[col, 0] = np.uint64(int(pih[i]))
        if self.witness_standin:
            self.generate_witness_standin(circ, buf)
        return buf

    def generate_witness_standin(self, circ, buf):
        """CPU stand-in for upstream's per-proof witness generation (plonky2 generate_partial_witness inside generate_txn_proof,
        /root/reference/ops/src/lib.rs:52): the cells PoseidonGenerator / ArithmeticBaseGenerator produce are recomputed from the rows'
        inputs by zk_host_plonk_witness_recompute, in the calling (proving) thread -- 3 us a PoseidonGate row, 8 ms for a 2^12-row
        circuit, 70 ms for the 2^15-row root.  The witness satisfied its circuit, so nothing may change."""
        from . import lib as zl, gates as GT
        rows = getattr(circ, "_standin_rows", None)
        if rows is None:
            NS = len(GT.gate_program(circ.num_wires)[1])
            cs = np.asarray(circ.constants)
            pr = np.ascontiguousarray(circ.gate_rows["PoseidonGate"], dtype=np.uint32)
            ar = np.ascontiguousarray(circ.gate_rows["ArithmeticGate"], dtype=np.uint32)
            rows = circ._standin_rows = (pr, ar, np.ascontiguousarray(cs[NS][ar]), np.ascontiguousarray(cs[NS + 1][ar]))
        pr, ar, c0, c1 = rows
        L = zl.lib()
        L.zk_host_plonk_witness_recompute.restype = C.c_int64
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        t0 = time.perf_counter()
        changed = L.zk_host_plonk_witness_recompute(vp(buf), C.c_uint64(buf.shape[1]), C.c_uint32(buf.shape[0]), vp(pr), C.c_uint64(pr.size),
                                                    vp(ar), vp(c0), vp(c1), C.c_uint64(ar.size))
        if changed != 0:
            raise RuntimeError("witness stand-in: %d cells changed (or error): the stand-in witness did not satisfy its gates" % changed)
        with self._standin_lock:
            self.witness_standin_seconds += time.perf_counter() - t0
            self.witness_standin_rows += int(pr.size + ar.size)

    def job(self, ctx, kind, pubs, key=0, degree_bits=None):
        return self.state(ctx, kind, key, degree_bits), self.witness(ctx, kind, key, degree_bits, self.public_inputs_hash(ctx, pubs)), pubs

    # ---- ops.RecursionCircuits
    def shrink_chain(self, table, air, log_n):
        """the circuits of by_table[table].by_stark_size[log_n]: [("initial", key, bits), ("shrink", key, bits) ..]"""
        if self.ranges is not None:
            lo, hi = self.ranges[table]
            if not lo <= log_n < hi:
                raise KeyError("Missing preprocessed circuits for table %d with size %d." % (table, log_n))
        bits = self.chain_degree_bits(air.n_trace_cols, air.n_aux_cols, log_n)
        return [("initial" if i == 0 else "shrink", (table, log_n, i), b) for i, b in enumerate(bits)]

    def initial_wrapper(self, ctx, table, air, log_n, stark_proof_words):
        kind, key, bits = self.shrink_chain(table, air, log_n)[0]
        pubs = np.concatenate([np.array([table, log_n], dtype=np.uint64), self.digest(ctx, stark_proof_words)])
        return self.job(ctx, kind, pubs, key, bits)

    def shrinking_wrapper(self, ctx, table, air, log_n, step, inner_proof):
        kind, key, bits = self.shrink_chain(table, air, log_n)[step]
        return self.job(ctx, kind, self.digest(ctx, inner_proof), key, bits)

    def root(self, ctx, shrunk_proofs, stark_sizes):
        pubs = np.concatenate([np.array(stark_sizes, dtype=np.uint64)] + [self.digest(ctx, p) for p in shrunk_proofs])
        return self.job(ctx, "root", pubs)

    def aggregation(self, ctx, lhs, rhs):
        pubs = np.concatenate([self.digest(ctx, lhs.proof), self.digest(ctx, rhs.proof)])
        return self.job(ctx, "agg", pubs)

    def block(self, ctx, prev, agg, height=1):
        prev_d = self.digest(ctx, prev) if prev is not None else np.zeros(4, dtype=np.uint64)
        pubs = np.concatenate([np.array([height], dtype=np.uint64), self.digest(ctx, agg.proof), prev_d])
        return self.job(ctx, "block", pubs)


# ------------------------------------------------------------------------------------------------ circuits of the recursion shape
# share of the rows per gate in a recursive-verifier-like circuit: Merkle paths (PoseidonGate) dominate, then the arithmetic of
# the FRI checks and of the vanishing-polynomial evaluation; the rest of the rows up to the power of two are NoopGates
RECURSION_MIX = (("PoseidonGate", 0.60), ("ArithmeticGate", 0.14), ("ArithmeticExtensionGate", 0.08), ("ReducingGate", 0.05),
                 ("BaseSumGate", 0.05), ("ConstantGate", 0.02))
POSEIDON_CHAIN = 4          # PoseidonGate rows come in chains (a Merkle path: digest out -> input of the next row, a copy constraint)
_circuit_cache = {}


def recursion_circuit(degree_bits=12, seed=1, num_wires=135, num_routed=80, domain_separator=None):
    """(Circuit, wires): a circuit with the gate set, selector polynomials, gate mix and wire routing shape of plonky2's recursion
    circuits (gates.py) and a satisfying witness -- the STAND-IN for the shrink / root / aggregation / block circuits of
    AllRecursiveCircuits (/root/reference/common/src/prover_state/circuit.rs:200-206), whose gadgets are not in the reference tree.
    Row 0 is the PublicInputGate (wires 0..3 = public_inputs_hash: `bind_public_inputs`).  It verifies nothing in-circuit."""
    from . import gates as GT
    from . import glnp as G
    key = (degree_bits, seed, num_wires, num_routed, None if domain_separator is None else tuple(int(x) for x in domain_separator))
    if key in _circuit_cache:
        return _circuit_cache[key]
    assert num_wires >= 135 and num_routed >= 80 and num_routed % 8 == 0 and degree_bits >= 3
    n = 1 << degree_bits
    rng = np.random.default_rng([seed, degree_bits])
    prog, groups = GT.gate_program(num_wires)
    NS = len(groups)
    # ---- rows per gate
    counts, left = {}, n - 1
    for name, share in RECURSION_MIX:
        c = min(left, int(share * n))
        if name == "PoseidonGate":
            c -= c % POSEIDON_CHAIN
        counts[name] = c
        left -= c
    gate_of_row = np.full(n, GT.GATE_INDEX["NoopGate"], dtype=np.int64)
    gate_of_row[0] = GT.GATE_INDEX["PublicInputGate"]
    rows, at = {}, 1
    for name, _ in RECURSION_MIX:
        rows[name] = np.arange(at, at + counts[name])
        gate_of_row[rows[name]] = GT.GATE_INDEX[name]
        at += counts[name]
    # ---- constants: selector polynomials, then the gates' local constants
    consts = np.zeros((NS + GT.NUM_GATE_CONSTANTS, n), dtype=np.uint64)
    for gi, (lo, hi) in enumerate(groups):
        inside = (gate_of_row >= lo) & (gate_of_row < hi)
        consts[gi] = np.where(inside, gate_of_row, GT.UNUSED_SELECTOR if NS > 1 else 0).astype(np.uint64)
    consts[NS:] = G.rand(rng, (GT.NUM_GATE_CONSTANTS, n))
    k0, k1 = consts[NS], consts[NS + 1]
    # ---- witness
    wires = G.rand(rng, (num_wires, n))
    wires[:4, 0] = 0                                            # PublicInputGate cells: bind_public_inputs
    pairs = []                                                  # copy constraints: (col_a, rows_a, col_b, rows_b), cell by cell
    pr = rows["PoseidonGate"]
    if pr.size:
        chains = pr.reshape(POSEIDON_CHAIN, -1)                 # chains[t] = the t-th row of every chain
        prev_out = None
        for t in range(POSEIDON_CHAIN):
            m = chains.shape[1]
            inp = G.rand(rng, (12, m))
            if prev_out is not None:
                inp[0:4] = prev_out
            blk = GT.poseidon_rows(inp, rng.integers(0, 2, size=m, dtype=np.uint64))
            wires[:135, chains[t]] = blk
            if prev_out is not None:
                for i in range(4):
                    pairs.append((GT.PG_IN + i, chains[t], GT.PG_OUT + i, chains[t - 1]))
            prev_out = blk[GT.PG_OUT:GT.PG_OUT + 4]
    r_ = rows["ArithmeticGate"]
    if r_.size:
        wires[:135, r_] = GT.arithmetic_rows(rng, r_.size, k0[r_], k1[r_])
        for j in range(1, GT.ARITH_OPS):
            pairs.append((4 * j, r_, 4 * j - 1, r_))
    r_ = rows["ArithmeticExtensionGate"]
    if r_.size:
        wires[:135, r_] = GT.arithmetic_ext_rows(rng, r_.size, k0[r_], k1[r_])
    r_ = rows["ReducingGate"]
    if r_.size:
        wires[:135, r_] = GT.reducing_rows(rng, r_.size)
    r_ = rows["BaseSumGate"]
    if r_.size:
        wires[:135, r_] = GT.base_sum_rows(rng, r_.size)
    r_ = rows["ConstantGate"]
    if r_.size:
        wires[:135, r_] = GT.constant_rows(rng, r_.size, k0[r_], k1[r_])
    # ---- routed-wire permutation: sigma_j(w^r) = k_j' w^r' of the cell that (j, r) maps to; every class above is a 2-cycle
    wpow = G.powers(_root(degree_bits), n)
    ident = np.stack([G.mul(G.scalar(_pow(GEN, j), wpow), wpow) for j in range(num_routed)])
    sig = ident.copy()
    for ca, ra, cb, rb in pairs:
        assert ca < num_routed and cb < num_routed
        sig[ca, ra] = ident[cb, rb]
        sig[cb, rb] = ident[ca, ra]
    circ = Circuit(degree_bits, num_wires, num_routed, consts, sig, prog, domain_separator=domain_separator)
    circ.pi_wires = [0, 1, 2, 3]
    circ.gate_rows = {k: v for k, v in rows.items()}
    circ.gate_of_row = gate_of_row
    out = (circ, np.ascontiguousarray(wires, dtype=np.uint64))
    _circuit_cache[key] = out
    return out
