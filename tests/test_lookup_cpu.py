"""logUp lookup argument inside a table (starky::lookup) on the oracle: helper columns with filters, the forward
running sum with the multiplicity column, their constraints (emitted by air.lookup_air) -- the oracle's proof verifies, and a
trace whose looked values are not covered by the table with the claimed multiplicities is rejected."""
import numpy as np
import pytest


def _cfg():
    import eth_tx_proof_amd as E
    return E.standard_fast_config()


@pytest.mark.parametrize("n_looked,filtered", [(5, (1, 3)), (4, ()), (1, (0,))])
def test_lookup_columns_and_proof_on_the_oracle(orc, n_looked, filtered):
    from eth_tx_proof_amd import air
    P = 0xFFFFFFFF00000001
    cfg = _cfg()
    prog, tr, pub = air.lookup_air(n_looked, filtered, log_n=6, seed=3)
    ch = np.array([11, 22, 33, 44], dtype=np.uint64)
    rc, aux = orc.aux_columns(prog, ch, tr)
    assert rc == 0 and aux.shape[0] == prog.n_aux_cols
    LH = (n_looked + 1) // 2
    for j in range(2):
        z = aux[j * (LH + 1) + LH]
        assert z[0] == 0
        # the running sum closes: Z[n-1] + delta[n-1] == 0  <=>  total of the deltas is zero
        beta = int(ch[2 * j])
        total = 0
        for r in range(tr.shape[1]):
            d = sum(int(aux[j * (LH + 1) + b][r]) for b in range(LH)) - int(tr[1][r]) * pow((int(tr[0][r]) + beta) % P, P - 2, P)
            total = (total + d) % P
        assert total == 0
    rc, proof = orc.prove(cfg, [(prog, 6, tr, pub)])
    assert rc == 0 and orc.verify(cfg, [(prog, 6, None, pub)], proof) == 0
    bad = tr.copy(); bad[1, 5] += np.uint64(1)                        # a wrong multiplicity
    rc, p2 = orc.prove(cfg, [(prog, 6, bad, pub)])
    assert rc == 0 and orc.verify(cfg, [(prog, 6, None, pub)], p2) != 0
    bad = tr.copy(); bad[2, 9] = np.uint64(5)                         # a looked value that is not in the table
    if not filtered or 0 not in filtered or bad[2 + n_looked + list(filtered).index(0), 9] == 1:
        rc, p3 = orc.prove(cfg, [(prog, 6, bad, pub)])
        assert rc == 0 and orc.verify(cfg, [(prog, 6, None, pub)], p3) != 0


def test_lookup_needs_ctl_columns_and_valid_indices(zk):
    import ctypes as C
    from eth_tx_proof_amd import air
    L = zk.lib()
    prog, tr, pub = air.lookup_air(3, (), log_n=4)
    assert L.zk_air_validate(prog.ref, C.c_uint32(2)) == 0
    bad = air.AirProgram(4, 0, [], [1], [], 1, lookup=([(2, None)], 0, 1))          # lookup without CTL-shaped columns
    assert L.zk_air_validate(bad.ref, C.c_uint32(2)) == -1
    bad = air.AirProgram(4, 0, [(0, 1, 2, 3)], [1], [], 1, lookup=([(9, None)], 0, 1))   # looked column past the trace
    assert L.zk_air_validate(bad.ref, C.c_uint32(2)) == -1


def test_ctl_helpers_with_filters_and_column_sets(orc):
    """CTL helper columns in upstream's full form (cross_table_lookup::partial_sums, SURVEY.md A9: sum over two (column set, filter)
    pairs of filter / combine(cols), column sets of 2..5 columns): the oracle's columns satisfy their defining relation row by row, the
    whole proof verifies, equals the pure-Python restatement word for word, and a flipped filter bit is caught by the verifier."""
    import pyspec
    from eth_tx_proof_amd import air
    from eth_tx_proof_amd._abi import StarkConfig
    P = 0xFFFFFFFF00000001
    prog, tr, pub = air.ctl_air(n_sets=6, log_n=5, seed=3)
    assert any(f0 is not None or f1 is not None for f0, f1, _, _ in prog.helpers_ext) and any(len(e0) > 0 for _, _, e0, _ in prog.helpers_ext)
    ch = np.array([11, 22, 33, 44], dtype=np.uint64)
    rc, aux = orc.aux_columns(prog, ch, tr)
    assert rc == 0
    H, n = 3, 32
    for j in range(2):
        beta, gamma = int(ch[2 * j]), int(ch[2 * j + 1])
        for m, ((a0, b0, a1, b1), (f0, f1, e0, e1)) in enumerate(zip([(h.a0, h.b0, h.a1, h.b1) for h in prog._helpers[:H]], prog.helpers_ext)):
            for r in (0, 7, 31):
                x = [(sum(int(tr[c, r]) * pow(beta, i, P) for i, c in enumerate(cs)) + gamma) % P for cs in ([a0, b0] + list(e0), [a1, b1] + list(e1))]
                f = [int(tr[q, r]) if q is not None else 1 for q in (f0, f1)]
                assert (int(aux[j * H + m, r]) * x[0] * x[1] - f[0] * x[1] - f[1] * x[0]) % P == 0
    cfg = StarkConfig(100, 2, 1, 2, 3, 4, 5, 5)
    rc, ref = orc.prove(cfg, [(prog, 5, tr, pub)])
    assert rc == 0 and orc.verify(cfg, [(prog, 5, None, pub)], ref) == 0
    pcfg = dict(num_challenges=2, rate_bits=1, cap_height=2, proof_of_work_bits=3, fri_arity_bits=4, fri_final_poly_bits=5, num_query_rounds=5)
    table = dict(log_n=5, trace=[[int(x) for x in c] for c in tr], pub=[], instrs=[(i.op, i.dst, i.a, i.b) for i in prog._instrs[:prog.n_instrs]],
                 constants=[int(x) for x in prog._constants], n_regs=prog.air.n_regs, helpers=[(h.a0, h.b0, h.a1, h.b1) for h in prog._helpers[:H]],
                 helpers_ext=prog.helpers_ext)
    got = pyspec.prove_tables(pcfg, [table])
    assert len(got) == ref.size and not [i for i, (a, b) in enumerate(zip(got, ref)) if a != int(b)]
    # the refinement matters: the same helpers read as plain pairs (filters and extra columns ignored) give other columns, and those
    # do not satisfy the constraints of the filtered form
    plain = air.AirProgram(prog.n_trace_cols, 0, [(h.a0, h.b0, h.a1, h.b1) for h in prog._helpers[:H]], [1, 0],
                           [(i.op, i.dst, i.a, i.b) for i in prog._instrs[:prog.n_instrs]], 6)
    assert not (orc.aux_columns(plain, ch, tr)[1] == aux).all()
    rc, bad = orc.prove(cfg, [(plain, 5, tr, pub)])             # columns of the plain form do not satisfy the filtered constraints
    assert rc != 0 or orc.verify(cfg, [(plain, 5, None, pub)], bad) != 0
