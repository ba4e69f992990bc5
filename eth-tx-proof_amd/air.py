"""Synthetic degree-3 AIRs shaped like the zkEVM tables.

The real tables' `Stark::eval_packed_generic` bodies live in evm_arithmetization 0.1.3
(/root/reference/Cargo.lock:1675-1677), whose source is absent; BASELINE.json asks for
"synthetic fixed-degree trace tables".  An AIR is plain data (include/zkgl.h: zk_air): a straight-line
constraint program plus logUp/CTL-shaped helper-column descriptors.
"""
import ctypes as C

from ._abi import (Air, CTL_MAX_EXTRA, HelperDesc, HelperExt, Instr, Lookup, LookupCol, NO_FILTER, OP_ADD, OP_EMIT, OP_EMIT_FIRST, OP_EMIT_LAST,
                   OP_EMIT_TRANSITION, OP_MUL, OP_SUB, OPND_CHALLENGE, OPND_CONST, OPND_LOCAL, OPND_NEXT, OPND_PUBLIC, OPND_REG,
                   opnd, u64)


class AirProgram:
    """Owns the ctypes arrays behind a zk_air; `.air` is the struct, `.ref` a pointer to it."""

    def __init__(self, n_trace_cols, n_public_inputs, helpers, constants, instrs, n_regs, num_challenges=2, lookup=None, helpers_ext=None):
        """lookup: None or (looked [(col, filter_col or None)], table_col, freq_col) -- a logUp argument inside the table.
        helpers_ext: None or one (filter0, filter1, extra0, extra1) per helper: filter columns (None: constant 1) and the further columns
        of each pair's column set (zk_helper_ext: CTL helpers with filters and column sets of more than two columns)"""
        self.n_trace_cols = n_trace_cols
        self.n_helpers = len(helpers)
        self.n_aux_cols = num_challenges * (len(helpers) + 1) if helpers else 0
        self.n_lookup_aux = 0
        self._lookup = None
        if lookup is not None:
            looked, tcol, fcol = lookup
            self._lookup_cols = (LookupCol * len(looked))(*[LookupCol(c, NO_FILTER if f is None else f) for c, f in looked])
            self._lookup = Lookup(len(looked), tcol, fcol, 0, C.cast(self._lookup_cols, C.POINTER(LookupCol)))
            self.n_lookup_aux = num_challenges * ((len(looked) + 1) // 2 + 1)
            self.n_aux_cols += self.n_lookup_aux
        self._helpers = (HelperDesc * max(1, len(helpers)))(*[HelperDesc(*h) for h in helpers])
        self._constants = (u64 * max(1, len(constants)))(*constants)
        self._instrs = (Instr * max(1, len(instrs)))(*[Instr(*i) for i in instrs])
        self.helpers_ext = helpers_ext
        self._helpers_ext = None
        if helpers_ext is not None:
            assert len(helpers_ext) == len(helpers)
            self._helpers_ext = (HelperExt * max(1, len(helpers_ext)))()
            for m, (f0, f1, e0, e1) in enumerate(helpers_ext):
                x = self._helpers_ext[m]
                x.filter[0], x.filter[1] = (NO_FILTER if f0 is None else f0), (NO_FILTER if f1 is None else f1)
                x.n_extra[0], x.n_extra[1] = len(e0), len(e1)
                assert len(e0) <= CTL_MAX_EXTRA and len(e1) <= CTL_MAX_EXTRA
                for i, c in enumerate(e0):
                    x.extra[0][i] = c
                for i, c in enumerate(e1):
                    x.extra[1][i] = c
        self.air = Air(n_trace_cols, n_public_inputs, len(helpers), len(constants), n_regs, len(instrs), 3, 0,
                       C.cast(self._helpers, C.POINTER(HelperDesc)), C.cast(self._constants, C.POINTER(u64)),
                       C.cast(self._instrs, C.POINTER(Instr)),
                       C.pointer(self._lookup) if self._lookup is not None else C.POINTER(Lookup)(),
                       C.cast(self._helpers_ext, C.POINTER(HelperExt)) if self._helpers_ext is not None else C.POINTER(HelperExt)())
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


def lookup_air(n_looked=5, filtered=(1, 3), log_n=8, seed=1, num_challenges=2):
    """A table with a logUp lookup argument (starky::lookup) and one CTL-shaped helper, plus a satisfying trace.
    Columns: 0 table t (distinct values), 1 multiplicity m, 2 .. 2+n_looked-1 looked columns f_i (values drawn from t), then one 0/1
    filter column per index in `filtered`, then four free columns for the CTL-shaped helper.  Constraints (emission order = spec):
      per challenge j and batch b of two looked columns:  h (f_a + beta)(f_b + beta) - filt_a (f_b + beta) - filt_b (f_a + beta)
      Z first row = 0;  every row (wrap-around included):  (Z' - Z)(t + beta) - (sum_b h_b)(t + beta) + m
    followed by the CTL-shaped helper and running-sum constraints of cpu_like_air.  Returns (AirProgram, trace uint64 (cols, n), pub)."""
    import numpy as np
    P = 0xFFFFFFFF00000001
    n = 1 << log_n
    rng = np.random.default_rng(seed)
    filt_of = {i: 2 + n_looked + k for k, i in enumerate(filtered)}
    free0 = 2 + n_looked + len(filtered)
    n_cols = free0 + 4
    t = (np.arange(n, dtype=np.uint64) * np.uint64(7) + np.uint64(1000 + seed))
    trace = np.zeros((n_cols, n), dtype=np.uint64)
    trace[0] = t
    mult = np.zeros(n, dtype=np.uint64)
    for i in range(n_looked):
        idx = rng.integers(0, n, size=n)
        trace[2 + i] = t[idx]
        if i in filt_of:
            f = rng.integers(0, 2, size=n).astype(np.uint64)
            trace[filt_of[i]] = f
            np.add.at(mult, idx[f == 1], np.uint64(1))
        else:
            np.add.at(mult, idx, np.uint64(1))
    trace[1] = mult
    a = rng.integers(0, 1 << 62, size=(4, n), dtype=np.uint64)
    trace[free0:free0 + 4] = a
    looked = [(2 + i, filt_of.get(i)) for i in range(n_looked)]
    LH = (n_looked + 1) // 2
    helpers = [(free0, free0 + 1, free0 + 2, free0 + 3)]
    H = 1
    A_lk = num_challenges * (LH + 1)
    R = lambda i: opnd(OPND_REG, i)
    L = lambda i: opnd(OPND_LOCAL, i)
    N = lambda i: opnd(OPND_NEXT, i)
    K = lambda i: opnd(OPND_CONST, i)
    CH = lambda i: opnd(OPND_CHALLENGE, i)
    ins = []
    for j in range(num_challenges):
        beta = CH(2 * j)
        base = n_cols + j * (LH + 1)
        for b in range(LH):
            h = L(base + b)
            ca = looked[2 * b]
            ins += [(OP_ADD, 0, L(ca[0]), beta)]                                   # r0 = f_a + beta
            if 2 * b + 1 < n_looked:
                cb = looked[2 * b + 1]
                ins += [(OP_ADD, 1, L(cb[0]), beta),                               # r1 = f_b + beta
                        (OP_MUL, 2, R(0), R(1)), (OP_MUL, 2, R(2), h)]            # r2 = h (f_a + beta)(f_b + beta)
                ins += [(OP_MUL, 3, L(ca[1]), R(1))] if ca[1] is not None else [(OP_ADD, 3, R(1), K(1))]     # filt_a (f_b + beta)
                ins += [(OP_MUL, 4, L(cb[1]), R(0))] if cb[1] is not None else [(OP_ADD, 4, R(0), K(1))]     # filt_b (f_a + beta)
                ins += [(OP_SUB, 2, R(2), R(3)), (OP_SUB, 2, R(2), R(4)), (OP_EMIT, 0, R(2), 0)]
            else:
                ins += [(OP_MUL, 2, R(0), h)]
                ins += [(OP_SUB, 2, R(2), L(ca[1]))] if ca[1] is not None else [(OP_SUB, 2, R(2), K(0))]
                ins += [(OP_EMIT, 0, R(2), 0)]
            ins += [(OP_ADD, 5, R(5), h)] if b else [(OP_ADD, 5, h, K(1))]           # r5 = sum of the helpers
        z = base + LH
        ins += [(OP_EMIT_FIRST, 0, L(z), 0),
                (OP_ADD, 0, L(0), beta),                                             # t + beta
                (OP_SUB, 1, N(z), L(z)), (OP_MUL, 1, R(1), R(0)),
                (OP_MUL, 2, R(5), R(0)), (OP_SUB, 1, R(1), R(2)), (OP_ADD, 1, R(1), L(1)),
                (OP_EMIT, 0, R(1), 0)]
    for j in range(num_challenges):           # the CTL-shaped helper and its running sum, as in cpu_like_air
        beta, gamma = CH(2 * j), CH(2 * j + 1)
        a0, b0, a1, b1 = helpers[0]
        h = L(n_cols + A_lk + j * H)
        ins += [(OP_MUL, 0, beta, L(b0)), (OP_ADD, 0, R(0), L(a0)), (OP_ADD, 0, R(0), gamma),
                (OP_MUL, 1, beta, L(b1)), (OP_ADD, 1, R(1), L(a1)), (OP_ADD, 1, R(1), gamma),
                (OP_MUL, 2, R(0), R(1)), (OP_MUL, 2, R(2), h), (OP_SUB, 2, R(2), R(0)), (OP_SUB, 2, R(2), R(1)), (OP_EMIT, 0, R(2), 0)]
        z = n_cols + A_lk + num_challenges * H + j
        ins += [(OP_SUB, 0, L(z), h), (OP_EMIT_LAST, 0, R(0), 0),
                (OP_SUB, 0, L(z), N(z)), (OP_SUB, 0, R(0), h), (OP_EMIT_TRANSITION, 0, R(0), 0)]
    prog = AirProgram(n_cols, 0, helpers, [1, 0], ins, 6, num_challenges, lookup=(looked, 0, 1))
    return prog, np.ascontiguousarray(trace % np.uint64(P)), np.zeros(0, dtype=np.uint64)


def ctl_air(n_sets=6, log_n=7, seed=1, num_challenges=2):
    """A table whose auxiliary columns are cross-table-lookup helpers in upstream's full form (starky::cross_table_lookup::partial_sums,
    SURVEY.md A9): every helper sums  filter / combine(column set)  over two (column set, filter) pairs -- column sets of 2 to 5 columns,
    some pairs filtered by a 0/1 column, some not -- and Z is the suffix sum of all helpers.  Constraints (emission order = spec), per
    challenge j and helper m with x_k = combine_j(set k):   h x_0 x_1 - f_0 x_1 - f_1 x_0,   then  Z = sum_m h  on the last row and
    Z - Z' - sum_m h  on every transition.  Returns (AirProgram, trace uint64 (cols, n), public inputs (none))."""
    import numpy as np
    P = 0xFFFFFFFF00000001
    n = 1 << log_n
    rng = np.random.default_rng(seed)
    n_val, n_flt = 10, 3                    # value columns 0..9, then three 0/1 filter columns
    n_cols = n_val + n_flt
    trace = rng.integers(0, 1 << 62, size=(n_cols, n), dtype=np.uint64)
    trace[n_val:] = rng.integers(0, 2, size=(n_flt, n), dtype=np.uint64)
    sets = []                               # (columns, filter or None)
    for k in range(n_sets):
        ln = 2 + k % 4
        cols = [int(c) for c in rng.choice(n_val, size=ln, replace=False)]
        sets.append((cols, (n_val + k % n_flt) if k % 3 != 2 else None))
    assert n_sets % 2 == 0
    helpers, ext = [], []
    for m in range(n_sets // 2):
        (c0, f0), (c1, f1) = sets[2 * m], sets[2 * m + 1]
        helpers.append((c0[0], c0[1], c1[0], c1[1]))
        ext.append((f0, f1, c0[2:], c1[2:]))
    H = len(helpers)
    R = lambda i: opnd(OPND_REG, i)
    L = lambda i: opnd(OPND_LOCAL, i)
    N = lambda i: opnd(OPND_NEXT, i)
    K = lambda i: opnd(OPND_CONST, i)
    CH = lambda i: opnd(OPND_CHALLENGE, i)
    ins = []
    for j in range(num_challenges):
        beta, gamma = CH(2 * j), CH(2 * j + 1)

        def combine(dst, cols):             # Horner from the last column: ((c_last beta + ..) beta + c_0) + gamma
            ins.append((OP_ADD, dst, L(cols[-1]), K(1)))
            for c in reversed(cols[:-1]):
                ins.extend([(OP_MUL, dst, R(dst), beta), (OP_ADD, dst, R(dst), L(c))])
            ins.append((OP_ADD, dst, R(dst), gamma))
        for m in range(H):
            (c0, f0), (c1, f1) = sets[2 * m], sets[2 * m + 1]
            h = L(n_cols + j * H + m)
            combine(0, c0)
            combine(1, c1)
            ins += [(OP_MUL, 2, R(0), R(1)), (OP_MUL, 2, R(2), h)]
            ins += [(OP_MUL, 3, L(f0), R(1))] if f0 is not None else [(OP_ADD, 3, R(1), K(1))]
            ins += [(OP_MUL, 4, L(f1), R(0))] if f1 is not None else [(OP_ADD, 4, R(0), K(1))]
            ins += [(OP_SUB, 2, R(2), R(3)), (OP_SUB, 2, R(2), R(4)), (OP_EMIT, 0, R(2), 0)]
            ins += [(OP_ADD, 5, R(5), h)] if m else [(OP_ADD, 5, h, K(1))]
        z = n_cols + num_challenges * H + j
        ins += [(OP_SUB, 0, L(z), R(5)), (OP_EMIT_LAST, 0, R(0), 0),
                (OP_SUB, 0, L(z), N(z)), (OP_SUB, 0, R(0), R(5)), (OP_EMIT_TRANSITION, 0, R(0), 0)]
    prog = AirProgram(n_cols, 0, helpers, [1, 0], ins, 6, num_challenges, helpers_ext=ext)
    return prog, np.ascontiguousarray(trace % np.uint64(P)), np.zeros(0, dtype=np.uint64)
