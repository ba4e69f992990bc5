"""Synthetic degree-3 AIRs shaped like the zkEVM tables.

The real tables' `Stark::eval_packed_generic` bodies live in evm_arithmetization 0.1.3
(/root/reference/Cargo.lock:1675-1677), whose source is absent; BASELINE.json asks for
"synthetic fixed-degree trace tables".  An AIR is plain data (include/zkgl.h: zk_air): a straight-line
constraint program plus logUp/CTL-shaped helper-column descriptors.
"""
import ctypes as C

from ._abi import (Air, HelperDesc, Instr, OP_ADD, OP_EMIT, OP_EMIT_FIRST, OP_EMIT_LAST, OP_EMIT_TRANSITION, OP_MUL,
                   OP_SUB, OPND_CHALLENGE, OPND_CONST, OPND_LOCAL, OPND_NEXT, OPND_PUBLIC, OPND_REG, opnd, u64)


class AirProgram:
    """Owns the ctypes arrays behind a zk_air; `.air` is the struct, `.ref` a pointer to it."""

    def __init__(self, n_trace_cols, n_public_inputs, helpers, constants, instrs, n_regs, num_challenges=2):
        self.n_trace_cols = n_trace_cols
        self.n_helpers = len(helpers)
        self.n_aux_cols = num_challenges * (len(helpers) + 1) if helpers else 0
        self._helpers = (HelperDesc * max(1, len(helpers)))(*[HelperDesc(*h) for h in helpers])
        self._constants = (u64 * max(1, len(constants)))(*constants)
        self._instrs = (Instr * max(1, len(instrs)))(*[Instr(*i) for i in instrs])
        self.air = Air(n_trace_cols, n_public_inputs, len(helpers), len(constants), n_regs, len(instrs), 3, 0,
                       C.cast(self._helpers, C.POINTER(HelperDesc)), C.cast(self._constants, C.POINTER(u64)),
                       C.cast(self._instrs, C.POINTER(Instr)))
        self.ref = C.pointer(self.air)
        self.n_instrs = len(instrs)


def cpu_like_air(n_cols=93, n_helpers=19, num_challenges=2):
    """Column triples (a,b,c) with a'=a*b+c, b'=b^2+a, c'=a*b*c+1; leftover columns count up by one;
    boundary constraints on the first/last row; n_helpers helper columns per challenge plus a running-sum Z column.
    Constraint emission order is part of the spec (alpha-Horner accumulation)."""
    ins = []
    R = lambda i: opnd(OPND_REG, i)
    L = lambda i: opnd(OPND_LOCAL, i)
    N = lambda i: opnd(OPND_NEXT, i)
    K = lambda i: opnd(OPND_CONST, i)
    PUB = lambda i: opnd(OPND_PUBLIC, i)
    CH = lambda i: opnd(OPND_CHALLENGE, i)
    constants = [1]
    T = n_cols // 3
    for t in range(T):
        a, b, c = 3 * t, 3 * t + 1, 3 * t + 2
        ins += [(OP_MUL, 0, L(a), L(b)),            # r0 = a*b
                (OP_ADD, 1, R(0), L(c)),            # r1 = a*b + c
                (OP_SUB, 1, N(a), R(1)),
                (OP_EMIT_TRANSITION, 0, R(1), 0),
                (OP_MUL, 1, L(b), L(b)),
                (OP_ADD, 1, R(1), L(a)),
                (OP_SUB, 1, N(b), R(1)),
                (OP_EMIT_TRANSITION, 0, R(1), 0),
                (OP_MUL, 1, R(0), L(c)),
                (OP_ADD, 1, R(1), K(0)),
                (OP_SUB, 1, N(c), R(1)),
                (OP_EMIT_TRANSITION, 0, R(1), 0)]
    for c in range(3 * T, n_cols):
        ins += [(OP_ADD, 0, L(c), K(0)), (OP_SUB, 0, N(c), R(0)), (OP_EMIT_TRANSITION, 0, R(0), 0)]
    ins += [(OP_SUB, 0, L(0), PUB(0)), (OP_EMIT_FIRST, 0, R(0), 0)]
    if n_cols > 1:
        ins += [(OP_SUB, 0, L(1), PUB(1)), (OP_EMIT_FIRST, 0, R(0), 0)]
    ins += [(OP_SUB, 0, L(0), PUB(2)), (OP_EMIT_LAST, 0, R(0), 0)]
    helpers = [((4 * m) % n_cols, (4 * m + 1) % n_cols, (4 * m + 2) % n_cols, (4 * m + 3) % n_cols)
               for m in range(n_helpers)]
    H = n_helpers
    n_regs = 6
    if H:
        for j in range(num_challenges):
            beta, gamma = CH(2 * j), CH(2 * j + 1)
            # r5 accumulates sum of helpers of this challenge
            for m, (a0, b0, a1, b1) in enumerate(helpers):
                h = L(n_cols + j * H + m)
                ins += [(OP_MUL, 0, beta, L(b0)), (OP_ADD, 0, R(0), L(a0)), (OP_ADD, 0, R(0), gamma),   # x1
                        (OP_MUL, 1, beta, L(b1)), (OP_ADD, 1, R(1), L(a1)), (OP_ADD, 1, R(1), gamma),   # x2
                        (OP_MUL, 2, R(0), R(1)), (OP_MUL, 2, R(2), h),                                   # h*x1*x2
                        (OP_SUB, 2, R(2), R(0)), (OP_SUB, 2, R(2), R(1)),
                        (OP_EMIT, 0, R(2), 0)]
                ins += [(OP_ADD, 5, R(5), h)] if m else [(OP_ADD, 5, h, opnd(OPND_CONST, 1))]
            z = n_cols + num_challenges * H + j
            ins += [(OP_SUB, 0, L(z), R(5)), (OP_EMIT_LAST, 0, R(0), 0),
                    (OP_SUB, 0, L(z), N(z)), (OP_SUB, 0, R(0), R(5)), (OP_EMIT_TRANSITION, 0, R(0), 0)]
        constants.append(0)
    return AirProgram(n_cols, 3, helpers, constants, ins, n_regs, num_challenges)


# widths recalled for the seven zkEVM tables: Arithmetic, BytePacking, Cpu, Keccak, KeccakSponge, Logic, Memory
ZKEVM_TABLE_WIDTHS = (165, 71, 93, 2431, 430, 523, 21)
ZKEVM_CFG4_LOG_SIZES = (15, 9, 12, 14, 9, 12, 17)
