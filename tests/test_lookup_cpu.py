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
