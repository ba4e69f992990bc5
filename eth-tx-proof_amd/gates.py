"""A gate set of the recursion shape, as data: what plonky2's recursion circuits (shrink / root / aggregation / block, proved behind
/root/reference/ops/src/lib.rs:52,72,95 with CircuitConfig::standard_recursion_config()) evaluate at every point of the 8n coset.

The real circuits (recursive verifier gadgets + witness generators) live in crates absent from the reference tree.
What this module restates from the published plonky2 0.2.2 is the part that decides the PROVER's cost -- the gates themselves:

  NoopGate, ConstantGate{2}, PublicInputGate, BaseSumGate<2>{63 limbs}, ReducingGate<2>{43 coeffs},
  ArithmeticExtensionGate{10 ops}, ArithmeticGate{20 ops}, PoseidonGate (123 constraints of degree 7: swap, 4 deltas, 3 x 12 + 22 +
  4 x 12 s-box inputs, 12 outputs; 135 wires = why the recursion config has 135),

the selector polynomials (plonk::selectors::selector_polynomials: gates sorted by (degree, id), greedy groups with
size + degree < max_degree 8, UNUSED_SELECTOR = 2^32 - 1, filter = prod_{i in group, i != k} (i - s) * (UNUSED - s)), and upstream's
evaluation scheme (vanishing_poly::evaluate_gate_constraints_base_batch): EVERY gate is evaluated at every point, its constraints
multiplied by its filter and ADDED position by position -> max-over-gates = 123 gate terms (ZK_OP_EMIT_SLOT, include/zkgl.h).

PoseidonGate's partial rounds use the sparse form of poseidon_tables.py (same shape and cost as upstream's FAST_PARTIAL_* tables: one
11x11 initial matrix, then per round one s-box, one scalar constant, 12 + 11 multiply-adds; the constant is added after the s-box,
before the sparse matrix, as upstream does).  Whether upstream's derived tables equal these value for value cannot be checked here
(source absent); the permutation they compute is pinned by upstream's four known-answer vectors.

`build(num_wires, num_routed)` returns the gate program (air.AirProgram for zk_plonk_circuit.gates), the selector layout, and per-gate
witness fillers working on numpy column blocks (glnp), so that a 2^15-row witness takes seconds."""
import numpy as np

from . import glnp as G
from . import poseidon_tables as PT
from ._abi import OP_ADD, OP_EMIT_SLOT, OP_MUL, OP_SUB, OPND_CONST, OPND_LOCAL, OPND_PUBLIC, OPND_REG, opnd
from .air import AirProgram

P = G.P_INT
UNUSED_SELECTOR = (1 << 32) - 1
MAX_DEGREE = 8                      # CircuitConfig::max_quotient_degree_factor
NUM_GATE_CONSTANTS = 2              # CircuitConfig::num_constants

# gates in upstream's order: sorted by (degree, id string)
GATES = [("NoopGate", 0, 0), ("ConstantGate", 1, 2), ("PublicInputGate", 1, 4), ("BaseSumGate", 2, 64), ("ReducingGate", 2, 86),
         ("ArithmeticExtensionGate", 3, 20), ("ArithmeticGate", 3, 20), ("PoseidonGate", 7, 123)]
GATE_INDEX = {name: i for i, (name, _, _) in enumerate(GATES)}
NUM_GATE_CONSTRAINTS = max(c for _, _, c in GATES)

# PoseidonGate wire layout (plonky2 gates/poseidon.rs)
PG_IN, PG_OUT, PG_SWAP, PG_DELTA, PG_FULL0, PG_PARTIAL, PG_FULL1 = 0, 12, 24, 25, 29, 65, 87
BASESUM_LIMBS = 63
REDUCING_COEFFS = 43
ARITH_OPS = 20
ARITH_EXT_OPS = 10


def selector_groups():
    """plonk::selectors::selector_polynomials: [(lo, hi)) ranges of gate indices, one selector polynomial per group"""
    if GATES[-1][1] + len(GATES) - 1 <= MAX_DEGREE:
        return [(0, len(GATES))]
    groups, start = [], 0
    while start < len(GATES):
        size = 0
        while start + size < len(GATES) and size + GATES[start + size][1] < MAX_DEGREE:
            size += 1
        groups.append((start, start + size))
        start += size
    return groups


class _B:
    """program builder: registers 0 = filter of the current gate, 1..4 temporaries, 5..16 state S, 17..28 state B"""

    def __init__(self, num_wires, num_selectors):
        self.W, self.NS = num_wires, num_selectors
        self.ins, self.consts, self._ci = [], [], {}

    def wire(self, i):
        return opnd(OPND_LOCAL, i)

    def sel(self, g):
        return opnd(OPND_LOCAL, self.W + g)

    def k(self, i):
        return opnd(OPND_LOCAL, self.W + self.NS + i)

    def c(self, v):
        v = int(v) % P
        if v not in self._ci:
            self._ci[v] = len(self.consts)
            self.consts.append(v)
        return opnd(OPND_CONST, self._ci[v])

    @staticmethod
    def r(i):
        return opnd(OPND_REG, i)

    def add(self, d, a, b):
        self.ins.append((OP_ADD, d, a, b))

    def sub(self, d, a, b):
        self.ins.append((OP_SUB, d, a, b))

    def mul(self, d, a, b):
        self.ins.append((OP_MUL, d, a, b))

    def emit(self, slot, a):
        """constraint `slot` of the current gate: a * filter"""
        self.mul(1, a, self.r(0))
        self.ins.append((OP_EMIT_SLOT, slot, self.r(1), 0))


F, T0, T1, T2, T3 = 0, 1, 2, 3, 4
S0, B0 = 5, 17
N_REGS = 29


def _filter(b, gate, groups):
    """r0 = compute_filter(gate, group range, s = the group's selector column, many_selectors)"""
    gi = next(i for i, (lo, hi) in enumerate(groups) if lo <= gate < hi)
    lo, hi = groups[gi]
    s = b.sel(gi)
    terms = [i for i in range(lo, hi) if i != gate] + ([UNUSED_SELECTOR] if len(groups) > 1 else [])
    if not terms:
        b.add(F, b.c(1), b.c(0))
        return
    b.sub(F, b.c(terms[0]), s)
    for t in terms[1:]:
        b.sub(T0, b.c(t), s)
        b.mul(F, b.r(F), b.r(T0))


def _ext_mul(b, d0, d1, a, bb):
    """(d0, d1) = a * bb in F_p[x]/(x^2 - 7); a, bb = (operand, operand); uses T2, T3; d0/d1 must differ from the inputs' registers"""
    b.mul(T2, a[0], bb[0])
    b.mul(T3, a[1], bb[1])
    b.mul(T3, b.r(T3), b.c(7))
    b.add(d0, b.r(T2), b.r(T3))
    b.mul(T2, a[0], bb[1])
    b.mul(T3, a[1], bb[0])
    b.add(d1, b.r(T2), b.r(T3))


def _constant_gate(b):
    for i in range(NUM_GATE_CONSTANTS):
        b.sub(T2, b.k(i), b.wire(i))
        b.emit(i, b.r(T2))


def _public_input_gate(b):
    for i in range(4):
        b.sub(T2, b.wire(i), opnd(OPND_PUBLIC, i))
        b.emit(i, b.r(T2))


def _base_sum_gate(b):
    # computed_sum = reduce_with_powers(limbs, 2) (Horner from the last limb); constraint 0: computed_sum - sum; then limb (limb - 1)
    b.add(T2, b.wire(BASESUM_LIMBS), b.c(0))
    for i in range(BASESUM_LIMBS - 1, 0, -1):
        b.add(T2, b.r(T2), b.r(T2))
        b.add(T2, b.r(T2), b.wire(i))
    b.sub(T2, b.r(T2), b.wire(0))
    b.emit(0, b.r(T2))
    for i in range(BASESUM_LIMBS):
        b.sub(T2, b.wire(1 + i), b.c(1))
        b.mul(T2, b.r(T2), b.wire(1 + i))
        b.emit(1 + i, b.r(T2))


def reducing_acc_wires(i):
    return (0, 1) if i == REDUCING_COEFFS - 1 else (6 + REDUCING_COEFFS + 2 * i, 7 + REDUCING_COEFFS + 2 * i)


def _reducing_gate(b):
    alpha = (b.wire(2), b.wire(3))
    acc = (b.wire(4), b.wire(5))
    for i in range(REDUCING_COEFFS):
        _ext_mul(b, S0, S0 + 1, acc, alpha)
        b.add(S0, b.r(S0), b.wire(6 + i))
        a0, a1 = reducing_acc_wires(i)
        b.sub(T2, b.r(S0), b.wire(a0))
        b.emit(2 * i, b.r(T2))
        b.sub(T2, b.r(S0 + 1), b.wire(a1))
        b.emit(2 * i + 1, b.r(T2))
        acc = (b.wire(a0), b.wire(a1))


def _arithmetic_ext_gate(b):
    for i in range(ARITH_EXT_OPS):
        w = 8 * i
        _ext_mul(b, S0, S0 + 1, (b.wire(w), b.wire(w + 1)), (b.wire(w + 2), b.wire(w + 3)))
        for d in range(2):
            b.mul(T2, b.r(S0 + d), b.k(0))
            b.mul(T3, b.wire(w + 4 + d), b.k(1))
            b.add(T2, b.r(T2), b.r(T3))
            b.sub(T2, b.wire(w + 6 + d), b.r(T2))
            b.emit(2 * i + d, b.r(T2))


def _arithmetic_gate(b):
    for i in range(ARITH_OPS):
        w = 4 * i
        b.mul(T2, b.wire(w), b.wire(w + 1))
        b.mul(T2, b.r(T2), b.k(0))
        b.mul(T3, b.wire(w + 2), b.k(1))
        b.add(T2, b.r(T2), b.r(T3))
        b.sub(T2, b.wire(w + 3), b.r(T2))
        b.emit(i, b.r(T2))


def _sbox(b, d, x):
    """reg d = x^7 (x an operand; d may be the register behind x)"""
    b.mul(T2, x, x)
    b.mul(T3, b.r(T2), x)
    b.mul(T2, b.r(T2), b.r(T2))
    b.mul(d, b.r(T2), b.r(T3))


def _mds_row(b, d, base, r):
    """reg d = row r of the MDS layer over the state in registers base..base+11 (d outside that block)"""
    first = True
    for i in range(12):
        src = b.r(base + (i + r) % 12)
        coef = PT.MDS_CIRC[i] + (PT.MDS_DIAG[r] if i == 0 else 0)
        if first:
            b.mul(d, src, b.c(coef))
            first = False
        else:
            b.mul(T3, src, b.c(coef))
            b.add(d, b.r(d), b.r(T3))


def _poseidon_gate(b):
    T = PT.tables()
    RC = T["RC"]
    S = lambda i: b.r(S0 + i)
    Bq = lambda i: b.r(B0 + i)
    swap = b.wire(PG_SWAP)
    j = 0
    b.sub(T2, swap, b.c(1))
    b.mul(T2, b.r(T2), swap)
    b.emit(j, b.r(T2)); j += 1
    for i in range(4):
        b.sub(T2, b.wire(PG_IN + i + 4), b.wire(PG_IN + i))
        b.mul(T2, b.r(T2), swap)
        b.sub(T2, b.r(T2), b.wire(PG_DELTA + i))
        b.emit(j, b.r(T2)); j += 1
    for i in range(4):
        b.add(S0 + i, b.wire(PG_IN + i), b.wire(PG_DELTA + i))
        b.sub(S0 + i + 4, b.wire(PG_IN + i + 4), b.wire(PG_DELTA + i))
    for i in range(8, 12):
        b.add(S0 + i, b.wire(PG_IN + i), b.c(0))
    # first full rounds: round 0 from the (swapped) inputs, rounds 1..3 from their s-box-input wires
    for i in range(12):
        b.add(S0 + i, S(i), b.c(RC[i]))
        _sbox(b, S0 + i, S(i))
    for r in range(1, 4):
        for i in range(12):                                   # MDS output i + round constant must equal the wire; then the state IS the wire
            _mds_row(b, T2, S0, i)
            b.add(T2, b.r(T2), b.c(RC[12 * r + i]))
            b.sub(T2, b.r(T2), b.wire(PG_FULL0 + 12 * (r - 1) + i))
            b.emit(j, b.r(T2)); j += 1
        for i in range(12):
            _sbox(b, S0 + i, b.wire(PG_FULL0 + 12 * (r - 1) + i))
    for i in range(12):                                       # MDS of round 3, partial_first_constant_layer
        _mds_row(b, B0 + i, S0, i)
        b.add(B0 + i, Bq(i), b.c(T["PR_FIRST"][i]))
    b.add(S0, Bq(0), b.c(0))                                  # mds_partial_layer_init
    for jj in range(11):
        for k in range(11):
            if k == 0:
                b.mul(S0 + 1 + jj, Bq(1), b.c(T["PR_INIT"][jj][0]))
            else:
                b.mul(T3, Bq(1 + k), b.c(T["PR_INIT"][jj][k]))
                b.add(S0 + 1 + jj, S(1 + jj), b.r(T3))
    for i in range(22):
        b.sub(T2, S(0), b.wire(PG_PARTIAL + i))
        b.emit(j, b.r(T2)); j += 1
        _sbox(b, S0, b.wire(PG_PARTIAL + i))
        if T["PR_K"][i]:
            b.add(S0, S(0), b.c(T["PR_K"][i]))
        b.mul(B0, S(0), b.c(T["M00"]))                        # d = M00 s0 + <VHAT, s[1..]>
        for k in range(11):
            b.mul(T3, S(1 + k), b.c(T["PR_VHAT"][i][k]))
            b.add(B0, Bq(0), b.r(T3))
        for k in range(11):
            b.mul(T3, S(0), b.c(T["PR_W"][i][k]))
            b.add(S0 + 1 + k, S(1 + k), b.r(T3))
        b.add(S0, Bq(0), b.c(0))
    # second full rounds: every one from its s-box-input wires
    for i in range(12):
        b.add(T2, S(i), b.c(RC[12 * 26 + i]))
        b.sub(T2, b.r(T2), b.wire(PG_FULL1 + i))
        b.emit(j, b.r(T2)); j += 1
    for r in range(4):
        for i in range(12):
            _sbox(b, S0 + i, b.wire(PG_FULL1 + 12 * r + i))
        for i in range(12):
            _mds_row(b, T2, S0, i)
            if r < 3:
                b.add(T2, b.r(T2), b.c(RC[12 * (27 + r) + i]))
                b.sub(T2, b.r(T2), b.wire(PG_FULL1 + 12 * (r + 1) + i))
            else:
                b.sub(T2, b.r(T2), b.wire(PG_OUT + i))
            b.emit(j, b.r(T2)); j += 1
    assert j == 123


_EMITTERS = {"NoopGate": lambda b: None, "ConstantGate": _constant_gate, "PublicInputGate": _public_input_gate,
             "BaseSumGate": _base_sum_gate, "ReducingGate": _reducing_gate, "ArithmeticExtensionGate": _arithmetic_ext_gate,
             "ArithmeticGate": _arithmetic_gate, "PoseidonGate": _poseidon_gate}
_program_cache = {}


def gate_program(num_wires=135):
    """(AirProgram, groups): the gate-constraint program of the whole gate set for zk_plonk_circuit.gates.  Constant columns of the
    circuit: one selector column per group, then NUM_GATE_CONSTANTS local constants."""
    if num_wires not in _program_cache:
        assert num_wires >= 135
        groups = selector_groups()
        b = _B(num_wires, len(groups))
        for gi, (name, _, _) in enumerate(GATES):
            if name == "NoopGate":
                continue
            _filter(b, gi, groups)
            _EMITTERS[name](b)
        prog = AirProgram(num_wires + len(groups) + NUM_GATE_CONSTANTS, 0, [], b.consts, b.ins, N_REGS)
        _program_cache[num_wires] = (prog, groups)
    return _program_cache[num_wires]


# ------------------------------------------------------------------------------------------------ witness fillers (numpy, vectorised)
def _gsum(rows):
    acc = rows[0]
    for r in rows[1:]:
        acc = G.add(acc, r)
    return acc


def _col(vals):
    return np.array([int(v) % P for v in vals], dtype=np.uint64)[:, None]


_MDS_MAT = np.array([[PT.MDS_CIRC[(c - r) % 12] + (PT.MDS_DIAG[r] if c == r else 0) for c in range(12)] for r in range(12)], dtype=np.uint64)


def _mds_np(s):
    """MDS layer of a (12, m) block: the constants are < 2^6, so low and high halves are accumulated exactly in 64 bits"""
    lo, hi = s & np.uint64(0xFFFFFFFF), s >> np.uint64(32)
    a_lo, a_hi = _MDS_MAT.dot(lo), _MDS_MAT.dot(hi)                       # < 2^42 each
    # a_lo + a_hi 2^32 = a_lo + (a_hi & M32) 2^32 + (a_hi >> 32) 2^64,  2^64 = 2^32 - 1 (mod p)
    mid = (a_hi & np.uint64(0xFFFFFFFF)) << np.uint64(32)
    top = (a_hi >> np.uint64(32)) * G.EPS
    mid = np.where(mid >= G.P, mid - G.P, mid)
    return G.add(G.add(a_lo, mid), top)


def poseidon_rows(inp, swap):
    """PoseidonGate witness of m rows: inp (12, m), swap (m,) in {0, 1} -> (135, m) wire block"""
    T = PT.tables()
    RC = T["RC"]
    m = inp.shape[1]
    w = np.zeros((135, m), dtype=np.uint64)
    w[PG_IN:PG_IN + 12] = inp
    w[PG_SWAP] = swap
    delta = G.mul(np.broadcast_to(swap, (4, m)), G.sub(inp[4:8], inp[0:4]))
    w[PG_DELTA:PG_DELTA + 4] = delta
    s = np.concatenate([G.add(inp[0:4], delta), G.sub(inp[4:8], delta), inp[8:12]])
    for r in range(4):
        s = G.add(s, _col(RC[12 * r:12 * r + 12]))
        if r:
            w[PG_FULL0 + 12 * (r - 1):PG_FULL0 + 12 * r] = s
        s = _mds_np(G.pow7(s))
    s = G.add(s, _col(T["PR_FIRST"]))
    init = np.array(T["PR_INIT"], dtype=np.uint64)
    rest = np.stack([_gsum(list(G.mul(init[j][:, None], s[1:]))) for j in range(11)])
    s0 = s[0]
    for i in range(22):
        w[PG_PARTIAL + i] = s0
        x = G.pow7(s0)
        if T["PR_K"][i]:
            x = G.add(x, G.scalar(T["PR_K"][i], x))
        d = G.add(G.mul(x, G.scalar(T["M00"], x)), _gsum(list(G.mul(_col(T["PR_VHAT"][i]), rest))))
        rest = G.add(rest, G.mul(_col(T["PR_W"][i]), np.broadcast_to(x, rest.shape)))
        s0 = d
    s = np.concatenate([s0[None, :], rest])
    for r in range(4):
        s = G.add(s, _col(RC[12 * (26 + r):12 * (26 + r) + 12]))
        w[PG_FULL1 + 12 * r:PG_FULL1 + 12 * (r + 1)] = s
        s = _mds_np(G.pow7(s))
    w[PG_OUT:PG_OUT + 12] = s
    return w


def arithmetic_rows(rng, m, c0, c1):
    """ArithmeticGate{20}: output_j = c0 m0 m1 + c1 addend; op j's first multiplicand is op j-1's output (a copy constraint)"""
    w = G.rand(rng, (135, m))
    for j in range(ARITH_OPS):
        if j:
            w[4 * j] = w[4 * j - 1]
        w[4 * j + 3] = G.add(G.mul(G.mul(w[4 * j], w[4 * j + 1]), c0), G.mul(w[4 * j + 2], c1))
    return w


def _ext_mul_np(a0, a1, b0, b1):
    return G.add(G.mul(a0, b0), G.mul(G.mul(a1, b1), G.scalar(7, a0))), G.add(G.mul(a0, b1), G.mul(a1, b0))


def arithmetic_ext_rows(rng, m, c0, c1):
    w = G.rand(rng, (135, m))
    for i in range(ARITH_EXT_OPS):
        o = 8 * i
        p0, p1 = _ext_mul_np(w[o], w[o + 1], w[o + 2], w[o + 3])
        w[o + 6] = G.add(G.mul(p0, c0), G.mul(w[o + 4], c1))
        w[o + 7] = G.add(G.mul(p1, c0), G.mul(w[o + 5], c1))
    return w


def base_sum_rows(rng, m):
    w = G.rand(rng, (135, m))
    bits = rng.integers(0, 2, size=(BASESUM_LIMBS, m), dtype=np.uint64)
    w[1:1 + BASESUM_LIMBS] = bits
    w[0] = (bits << np.arange(BASESUM_LIMBS, dtype=np.uint64)[:, None]).sum(axis=0, dtype=np.uint64)      # < 2^63 < p
    return w


def reducing_rows(rng, m):
    w = G.rand(rng, (135, m))
    a0, a1 = w[4], w[5]
    for i in range(REDUCING_COEFFS):
        a0, a1 = _ext_mul_np(a0, a1, w[2], w[3])
        a0 = G.add(a0, w[6 + i])
        c0_, c1_ = reducing_acc_wires(i)
        w[c0_], w[c1_] = a0, a1
    return w


def constant_rows(rng, m, k0, k1):
    w = G.rand(rng, (135, m))
    w[0], w[1] = k0, k1
    return w


# ------------------------------------------------------------------------------------------------ checking aid (python ints)
def eval_program_row(prog_instrs, consts, wires_row, consts_row, pih, num_wires, n_regs=N_REGS):
    """The gate program at one row over python ints -> {term index: value}; every value is 0 on a satisfying witness."""
    regs = [0] * n_regs
    out = {}

    def val(o):
        k, i = o >> 24, o & 0xFFFFFF
        if k == OPND_REG:
            return regs[i]
        if k == OPND_LOCAL:
            return int(wires_row[i]) if i < num_wires else int(consts_row[i - num_wires])
        if k == OPND_CONST:
            return consts[i]
        if k == OPND_PUBLIC:
            return int(pih[i])
        raise ValueError(o)
    for op, d, a, b in prog_instrs:
        if op == OP_ADD:
            regs[d] = (val(a) + val(b)) % P
        elif op == OP_SUB:
            regs[d] = (val(a) - val(b)) % P
        elif op == OP_MUL:
            regs[d] = (val(a) * val(b)) % P
        else:
            out[d] = (out.get(d, 0) + val(a)) % P
    return out
