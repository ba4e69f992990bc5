"""GPU parity of the finer-grained C-ABI pieces (one upstream item each) against the oracle's counterpart."""
import numpy as np
import pytest

from conftest import field_rand

pytestmark = pytest.mark.gpu


def _cfg():
    import eth_tx_proof_amd as E
    return E.standard_fast_config()


@pytest.mark.parametrize("n_cols,n_helpers,log_n", [(6, 2, 6), (93, 19, 9), (12, 9, 12)])
def test_aux_columns(ctx, zk, orc, n_cols, n_helpers, log_n):
    from eth_tx_proof_amd import air
    prog = air.cpu_like_air(n_cols, n_helpers)
    tr, _ = orc.synth_trace(n_cols, log_n, 77)
    ch = field_rand(np.random.default_rng(1), 4)
    got = zk.aux_columns(ctx, prog, ch, tr)
    rc, ref = orc.aux_columns(prog, ch, tr)
    assert rc == 0 and (got == ref).all()
    # Z columns are suffix sums of the helper columns: Z[last] == sum of helpers at the last row
    H = n_helpers
    P = 0xFFFFFFFF00000001
    for j in range(2):
        assert int(got[2 * H + j, -1]) == sum(int(got[j * H + m, -1]) for m in range(H)) % P


@pytest.mark.parametrize("n_cols,n_helpers,log_n", [(6, 2, 7), (93, 19, 9), (7, 0, 8)])
def test_quotient_polys(ctx, zk, orc, n_cols, n_helpers, log_n):
    from eth_tx_proof_amd import air
    cfg = _cfg()
    rng = np.random.default_rng(log_n)
    prog = air.cpu_like_air(n_cols, n_helpers)
    tr, pub = orc.synth_trace(n_cols, log_n, 5)
    ch, al = field_rand(rng, 4), field_rand(rng, 2)
    tb, ob = ctx.batch_from_values(tr), orc.Batch(tr)
    ab = oab = None
    if n_helpers:
        rc, aux = orc.aux_columns(prog, ch, tr)
        assert rc == 0
        ab, oab = ctx.batch_from_values(aux), orc.Batch(aux)
    got = zk.quotient(ctx, cfg, prog, tb, ab, pub, ch, al)
    ref = orc.quotient(cfg, prog, ob, oab, pub, ch, al)
    assert (got == ref).all()
    # a satisfied AIR has a quotient of degree < 2n - 1: the top coefficient of each pair of chunks vanishes
    assert int(got[1, -1]) == 0 and int(got[3, -1]) == 0


def test_pow_grind_smallest_witness(ctx, zk, orc):
    rng = np.random.default_rng(9)
    for pos, bits in ((0, 8), (3, 12), (7, 16)):
        st = field_rand(rng, 12)
        w = zk.pow_grind(ctx, st, pos, bits)
        assert w == orc.pow_grind(st, pos, bits)
        s2 = st.copy(); s2[pos] = w
        resp = int(orc.poseidon(s2)[7])
        assert resp < (1 << (64 - bits))


@pytest.mark.parametrize("n_helpers,log_n", [(2, 8), (0, 6), (3, 11)])
def test_fri_prove_openings(ctx, zk, orc, n_helpers, log_n):
    """same oracles, same challenger state, same zeta -> identical FriProof words and identical challenger afterwards"""
    from eth_tx_proof_amd import air
    cfg = _cfg()
    rng = np.random.default_rng(log_n)
    n_cols = 9
    prog = air.cpu_like_air(n_cols, n_helpers)
    tr, pub = orc.synth_trace(n_cols, log_n, 5)
    ch4, al = field_rand(rng, 4), field_rand(rng, 2)
    tb, ob = ctx.batch_from_values(tr), orc.Batch(tr)
    ab = oab = None
    if n_helpers:
        rc, aux = orc.aux_columns(prog, ch4, tr)
        ab, oab = ctx.batch_from_values(aux), orc.Batch(aux)
    q = orc.quotient(cfg, prog, ob, oab, pub, ch4, al)
    qb, oqb = ctx.batch_from_coeffs(q), orc.Batch(q, from_coeffs=True)
    gch, och = ctx.challenger(), orc.Challenger()
    seed = field_rand(rng, 11)
    gch.observe(seed); och.observe(seed)
    zeta = field_rand(rng, 2)
    got = zk.fri_prove_openings(ctx, cfg, tb, ab, qb, gch, zeta)
    rc, ref = orc.fri_prove(cfg, ob, oab, oqb, och, zeta)
    assert rc == 0 and got.size == ref.size and (got == ref).all()
    assert (gch.get(3) == och.get(3)).all()
