"""GPU parity: HIP kernels (through the C ABI) vs the CPU oracle on the same seeded inputs -- bit exact."""
import json
import os

import numpy as np
import pytest

from conftest import EDGE, P, field_rand

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
G = 14293326489335486720


def test_poseidon_kats_gpu(ctx):
    g = json.load(open(os.path.join(GOLD, "vectors.json")))
    inp = np.array([[int(x, 16) for x in k["input"]] for k in g["poseidon_kats"]], dtype=np.uint64)
    out = ctx.poseidon(inp)
    for row, k in zip(out, g["poseidon_kats"]):
        assert ["%016x" % int(x) for x in row] == k["output"]


def test_poseidon_random_and_edge(ctx, orc):
    rng = np.random.default_rng(10)
    s = field_rand(rng, (4099, 12))
    s[:8, :] = EDGE[:, None]            # includes non-canonical encodings
    s[8, :] = np.uint64(P - 1)
    assert (ctx.poseidon(s) == orc.poseidon_batch(s)).all()


@pytest.mark.parametrize("row_len", [1, 3, 4, 5, 8, 9, 16, 21, 93, 135])
def test_hash_no_pad(ctx, orc, row_len):
    rng = np.random.default_rng(row_len)
    rows = field_rand(rng, (777, row_len))
    assert (ctx.hash_rows(rows) == orc.hash_rows(rows)).all()


def test_two_to_one(ctx, orc):
    rng = np.random.default_rng(12)
    l, r = field_rand(rng, (300, 4)), field_rand(rng, (300, 4))
    assert (ctx.two_to_one(l, r) == orc.two_to_one(l, r)).all()


@pytest.mark.parametrize("L,ll,ch", [(16, 7, 2), (8, 3, 0), (8, 9, 3), (1024, 32, 4), (64, 12, 6), (4096, 5, 4)])
def test_merkle_build(ctx, orc, L, ll, ch):
    rng = np.random.default_rng(L + ll)
    leaves = field_rand(rng, (L, ll))
    cap, dig = ctx.merkle_build(leaves, ch)
    ocap, odig = orc.merkle_build(leaves, ch)
    assert (cap == ocap).all() and (dig == odig).all()


@pytest.mark.parametrize("log_n", [1, 2, 3, 5, 8, 9, 10, 11, 12, 13, 16, 18, 19, 20, 21])
def test_ntt_forward_inverse_coset(ctx, orc, log_n):
    rng = np.random.default_rng(100 + log_n)
    ncols = 3 if log_n <= 16 else 2
    v = field_rand(rng, (ncols, 1 << log_n))
    if log_n >= 3:
        v[0, :8] = EDGE
    f = ctx.ntt(v)
    assert (f == orc.fft(v)).all()
    assert (ctx.ntt(f, inverse=True) == v % np.uint64(P)).all()
    cf = ctx.ntt(v, shift=G)
    assert (cf == orc.fft(v, shift=G)).all()
    assert (ctx.ntt(cf, inverse=True, shift=G) == v % np.uint64(P)).all()


@pytest.mark.parametrize("log_n", [22, 25, 26])
def test_ntt_tall_transforms(ctx, orc, log_n):
    """Transforms beyond 2^24 (the reference's default table ranges reach 2^25 for Cpu and 2^28 for Memory,
    /root/reference/common/src/prover_state/circuit.rs:76-86): three digit passes through the register-radix kernel -- the middle pass
    reads ONE PERIOD of its inter-pass factors (a table of 2^(W+S) entries), the 1/n of the inverse is a constant, not a table.  Up to round 4
    these sizes fell to the LDS radix-2 kernel.  One column against the oracle's transform, and the round trip."""
    rng = np.random.default_rng(300 + log_n)
    v = field_rand(rng, (1, 1 << log_n))
    v[0, :8] = EDGE
    f = ctx.ntt(v)
    assert (f == orc.fft(v)).all()
    assert (ctx.ntt(f, inverse=True) == v % np.uint64(P)).all()
    if log_n <= 25:
        cf = ctx.ntt(v, shift=G)
        assert (ctx.ntt(cf, inverse=True, shift=G) == v % np.uint64(P)).all()
        if log_n <= 22:
            assert (cf == orc.fft(v, shift=G)).all()


@pytest.mark.parametrize("log_n,ncols", [(22, 3), (25, 2)])
def test_commit_of_tall_columns_register_radix_equals_lds_radix2_and_oracle(zk, orc, monkeypatch, log_n, ncols):
    """from_values at three digit passes (inverse DIF, then the coset DIT passes with the permuted last store): the register-radix passes
    against the LDS radix-2 kernel (ZK_NTT_SLOW=1: every pass the old way) -- same cap, same opened rows -- and at 2^22 against the
    oracle's commitment."""
    rng = np.random.default_rng(17 * log_n)
    v = field_rand(rng, (ncols, 1 << log_n))
    c = zk.Context(0)
    try:
        b = c.batch_from_values(v, rate_bits=1, cap_height=4)
        cap = b.cap.copy()
        idx = [0, 1, 12345, (1 << log_n) + 77, (2 << log_n) - 1]
        rows = [b.prove(i).copy() for i in idx]
        b.free()
        monkeypatch.setenv("ZK_NTT_SLOW", "1")
        b2 = c.batch_from_values(v, rate_bits=1, cap_height=4)
        assert (b2.cap == cap).all()
        for i, r in zip(idx, rows):
            assert (b2.prove(i) == r).all()
        b2.free()
        monkeypatch.delenv("ZK_NTT_SLOW")
        if log_n <= 22:
            o = orc.Batch(v, rate_bits=1, cap_height=4)
            assert (o.cap == cap).all()
    finally:
        c.close()


@pytest.mark.parametrize("log_n,r", [(4, 1), (6, 3), (10, 1), (12, 2), (14, 3), (16, 1), (5, 5), (3, 0)])
def test_ntt_zero_factor(ctx, orc, log_n, r):
    """zk_ntt_batch_opts: upstream's `zero_factor` of fft_with_options / coset_fft_with_options / ifft_with_options -- the transform of
    the input with everything past the first n >> r entries TAKEN AS zero (whatever the caller left there), equal to the oracle's full
    transform of the zero-padded input; plain, coset and inverse."""
    rng = np.random.default_rng(7 * log_n + r)
    n = 1 << log_n
    v = field_rand(rng, (3, n))                      # the tail is garbage on purpose: it must be ignored
    z = v.copy()
    z[:, n >> r:] = 0
    assert (ctx.ntt(v, zero_factor=r) == orc.fft(z)).all()
    assert (ctx.ntt(v, shift=G, zero_factor=r) == orc.fft(z, shift=G)).all()
    assert (ctx.ntt(v, inverse=True, zero_factor=r) == orc.fft(z, inverse=True)).all()
    assert (ctx.ntt(z, zero_factor=r) == ctx.ntt(z)).all()
    with pytest.raises(Exception):
        ctx.ntt(v, zero_factor=log_n + 1)


def test_ntt_golden(ctx):
    g = json.load(open(os.path.join(GOLD, "vectors.json")))
    v = np.array([g["fft_1_to_8"]["input"]], dtype=np.uint64)
    assert ctx.ntt(v)[0].tolist() == g["fft_1_to_8"]["output"]


def test_ntt_linearity_full_size(ctx):
    """size-independent property at BASELINE size 2^20: NTT(a + b) == NTT(a) + NTT(b), and round trip"""
    rng = np.random.default_rng(5)
    a, b = field_rand(rng, (1, 1 << 20)), field_rand(rng, (1, 1 << 20))
    s = ((a.astype(object) + b.astype(object)) % P).astype(np.uint64)
    fa, fb, fs = ctx.ntt(a), ctx.ntt(b), ctx.ntt(s)
    assert (((fa.astype(object) + fb.astype(object)) % P).astype(np.uint64) == fs).all()
    assert (ctx.ntt(fs, inverse=True) == s).all()


@pytest.mark.parametrize("log_n,ncols,cap_h", [(3, 1, 0), (4, 5, 2), (6, 9, 4), (10, 3, 4), (11, 8, 4), (12, 17, 4), (13, 93, 4), (7, 135, 4), (5, 4, 4), (9, 2, 4),
                                               (18, 3, 4), (19, 2, 4)])       # 9 + 9 and 10 + 9 bits: the W = 9 ROW passes with shift twiddles (round 5)
def test_batch_from_values(ctx, orc, log_n, ncols, cap_h):
    rng = np.random.default_rng(log_n * 100 + ncols)
    vals = field_rand(rng, (ncols, 1 << log_n))
    if log_n >= 3:
        vals[0, :8] = EDGE
    b = ctx.batch_from_values(vals, rate_bits=1, cap_height=cap_h)
    o = orc.Batch(vals, rate_bits=1, cap_height=cap_h)
    assert (b.cap == o.cap).all()
    assert (b.digests() == o.digests).all()
    for c in (0, ncols - 1):
        assert (b.coeffs(c) == o.coeffs[c]).all()
    lv = o.leaves
    L = 2 << log_n
    for leaf in (0, 1, L // 2 + 1, L - 1):
        assert (b.leaf(leaf) == lv[leaf]).all()
        assert (b.prove(leaf) == o.prove(leaf)).all()
    import pyspec
    idx = np.array([0, 1, 5 % L, L - 1], dtype=np.uint32)
    rows = b.lde_rows(idx, step=1)
    for i, r in zip(idx, rows):
        assert (r == lv[pyspec.bitrev(int(i), log_n + 1)]).all()
    z = field_rand(rng, 2)
    assert (b.eval_at(z) == o.eval_at(z)).all()
    b.free()


@pytest.mark.parametrize("log_n,ncols,rate_bits,cap_h", [(5, 3, 0, 2), (6, 5, 2, 4), (8, 135, 3, 4), (12, 9, 3, 4), (3, 2, 4, 0), (4, 4, 3, 4)])
def test_batch_rate_bits(ctx, orc, log_n, ncols, rate_bits, cap_h):
    """blow-ups other than the STARK config's 2: 1 (no LDE), 4, 8 (standard_recursion_config), 16"""
    rng = np.random.default_rng(log_n * 10 + rate_bits)
    vals = field_rand(rng, (ncols, 1 << log_n))
    b = ctx.batch_from_values(vals, rate_bits=rate_bits, cap_height=cap_h)
    o = orc.Batch(vals, rate_bits=rate_bits, cap_height=cap_h)
    assert (b.cap == o.cap).all()
    assert (b.digests() == o.digests).all()
    L = 1 << (log_n + rate_bits)
    for leaf in (0, 1, L // 2, L - 1):
        assert (b.leaf(leaf) == o.leaves[leaf]).all()
        assert (b.prove(leaf) == o.prove(leaf)).all()
    z = field_rand(rng, 2)
    assert (b.eval_at(z) == o.eval_at(z)).all()
    b.free()


@pytest.mark.parametrize("log_n,ncols", [(4, 4), (8, 4), (12, 4), (10, 7)])
def test_batch_from_coeffs(ctx, orc, log_n, ncols):
    rng = np.random.default_rng(log_n + ncols)
    co = field_rand(rng, (ncols, 1 << log_n))
    b = ctx.batch_from_coeffs(co, rate_bits=1, cap_height=4 if log_n >= 4 else 0)
    o = orc.Batch(co, rate_bits=1, cap_height=4 if log_n >= 4 else 0, from_coeffs=True)
    assert (b.cap == o.cap).all() and (b.digests() == o.digests).all()
    assert (b.coeffs(1) == co[1]).all()
    b.free()


def test_commit_properties_full_size(ctx, orc):
    """BASELINE cfg2 shape (2^20 rows): checked through size-independent properties -- coefficients round-trip to the
    input, sampled leaves equal a direct coset evaluation, sampled Merkle paths verify against the cap."""
    rng = np.random.default_rng(77)
    ncols, log_n = 8, 20
    vals = field_rand(rng, (ncols, 1 << log_n))
    b = ctx.batch_from_values(vals, rate_bits=1, cap_height=4)
    co = b.coeffs(3)
    assert (ctx.ntt(co[None, :])[0] == vals[3]).all()
    z = field_rand(rng, 2)
    ev = b.eval_at(z)
    import pyspec
    L = 2 << log_n
    wL = pyspec.root_of_unity(log_n + 1)
    for leaf in (0, 12345, L - 1):
        x = G * pow(wL, pyspec.bitrev(leaf, log_n + 1), P) % P
        row = b.leaf(leaf)
        # evaluate column 3 at the base-field point x with the same kernel used for openings
        assert int(b.eval_at(np.array([x, 0], dtype=np.uint64))[3][0]) == int(row[3])
        assert orc.merkle_verify(row, leaf, b.prove(leaf), b.cap)
    b.free()


def test_challenger(ctx, orc):
    c, o = ctx.challenger(), orc.Challenger()
    rng = np.random.default_rng(3)
    for n_obs, n_get in ((3, 2), (8, 1), (13, 9), (0, 3), (1, 0), (16, 84)):
        xs = field_rand(rng, n_obs)
        if n_obs:
            c.observe(xs); o.observe(xs)
        if n_get:
            assert (c.get(n_get) == o.get(n_get)).all()
    assert (c.compact() == o.compact()).all()


@pytest.mark.parametrize("mfma_bits,force", [("3", "0"), ("15", "0"), ("3", "1")])
def test_challenger_long_sponge_every_permutation_kernel(zk, orc, monkeypatch, mfma_bits, force):
    """The transcript's permutation: wave-wide (poseidon_wave.cuh, default), 16-lane (bits 2|3 of ZK_POSEIDON_MFMA) and the wave-wide
    one forced through its exact-redo branch -- thousands of absorbed words (the bulk path of TOP_OBSERVE: what a 2 431-column opening
    set costs), ragged lengths around the rate, values that need the rare borrow repair, then 100 draws: same as the oracle."""
    import json, os
    monkeypatch.setenv("ZK_POSEIDON_MFMA", mfma_bits)
    monkeypatch.setenv("ZK_FORCE_EXACT_REDO", force)
    rc = [int(x, 16) for x in json.load(open(os.path.join(os.path.dirname(__file__), "golden", "poseidon_rc.json")))["round_constants_hex"]]
    c2 = zk.Context(0)
    try:
        c, o = c2.challenger(), orc.Challenger()
        rng = np.random.default_rng(77)
        for n_obs, n_get in ((5000, 3), (7, 1), (8, 0), (9, 2), (1, 1), (0, 2), (2431 * 4 + 3, 100)):
            xs = field_rand(rng, n_obs)
            for k in range(0, n_obs, 13):
                xs[k] = ((1 << 63) - rc[k % 8]) % P         # s-box input 2^63 in the first round when it lands on word k % 8
            if n_obs:
                c.observe(xs); o.observe(xs)
            if n_get:
                assert (c.get(n_get) == o.get(n_get)).all()
        assert (c.compact() == o.compact()).all()
        # Merkle trees whose upper levels run one parent per wave / per 16 lanes
        rows = field_rand(rng, (4096, 5))
        cap, dig = c2.merkle_build(rows, 1)
        ocap, odig = orc.merkle_build(rows, 1)
        assert (cap == ocap).all() and (dig == odig).all()
    finally:
        c2.close()


def test_rare_borrow_inputs_take_the_exact_redo(ctx, orc):
    """poseidon.cuh skips the 1-in-2^32 borrow repair of each reduction and redoes the wave exactly when one was needed.
    An S-box input of 2^63 needs it (x^2 = 2^126: high word 2^30 is subtracted from a reduced 0): rows whose j-th element is
    2^63 - RC[j] hit that in the first round; a later absorbed block and Merkle nodes over such digests are covered too."""
    import json, os
    rc = [int(x, 16) for x in json.load(open(os.path.join(os.path.dirname(__file__), "golden", "poseidon_rc.json")))["round_constants_hex"]]
    rng = np.random.default_rng(2026)
    rows = field_rand(rng, (300, 21))
    for r in range(0, 300, 7):
        j = (r // 7) % 8
        rows[r, j] = ((1 << 63) - rc[j]) % P
    assert (ctx.hash_rows(rows) == orc.hash_rows(rows)).all()
    cap, dig = ctx.merkle_build(rows[:256], 2)
    ocap, odig = orc.merkle_build(rows[:256], 2)
    assert (cap == ocap).all() and (dig == odig).all()
    cols = np.ascontiguousarray(rows[:256].T)           # 21 columns x 256 rows: leaves of a from_coeffs commit are LDE values,
    b = ctx.batch_from_values(cols, rate_bits=1, cap_height=2)   # so here only the comparison with the oracle matters
    o = orc.Batch(cols, rate_bits=1, cap_height=2)
    assert (b.cap == o.cap).all()
    b.free()


def test_forced_exact_redo_matches(zk, orc, monkeypatch):
    """ZK_FORCE_EXACT_REDO=1 sends every wave of the hashing kernels through the exact-redo branch: same digests, same proof"""
    from eth_tx_proof_amd import air
    import eth_tx_proof_amd as E
    monkeypatch.setenv("ZK_FORCE_EXACT_REDO", "1")
    c2 = zk.Context(0)
    try:
        rng = np.random.default_rng(8)
        vals = field_rand(rng, (17, 1 << 10))
        b = c2.batch_from_values(vals, rate_bits=1, cap_height=4)
        o = orc.Batch(vals, rate_bits=1, cap_height=4)
        assert (b.cap == o.cap).all() and (b.digests() == o.digests).all()
        b.free()
        rows = field_rand(rng, (100, 13))
        assert (c2.hash_rows(rows) == orc.hash_rows(rows)).all()
        cfg = E.standard_fast_config()
        prog = air.cpu_like_air(9, 2)
        tr, pub = orc.synth_trace(9, 8, 3)
        proof = zk.prove_tables(c2, cfg, [(prog, 8, tr, pub)])
        rc, ref = orc.prove(cfg, [(prog, 8, tr, pub)])
        assert rc == 0 and (proof == ref).all()
    finally:
        c2.close()


@pytest.mark.gpu
@pytest.mark.parametrize("log_n,ncols,rate_bits", [(14, 21, 1), (13, 93, 2), (15, 8, 1), (14, 40, 1), (14, 5, 1), (13, 135, 1), (15, 9, 0), (14, 16, 0)])
def test_matrix_core_poseidon_equals_vector_alu_and_oracle(zk, orc, monkeypatch, log_n, ncols, rate_bits):
    """Big commits hash their leaves and their wide Merkle levels with the matrix-core permutation (poseidon_mfma.cuh,
    >= 16384 leaves / parents); ZK_POSEIDON_MFMA=0 keeps the one-thread-per-permutation kernels.  Both must give the oracle's
    cap and digests, also when every wave is sent through the exact-redo branch; rows that need the rare borrow repair
    (S-box input 2^63 in the first round) are planted in the values of a rate_bits-0 commit, where leaves are the values."""
    rng = np.random.default_rng(log_n * 100 + ncols)
    vals = field_rand(rng, (ncols, 1 << log_n))
    o = orc.Batch(vals, rate_bits=rate_bits, cap_height=4)
    # 19 = 3 | 16: pair layout (64 leaves per wave) also below 65536 leaves; 11 = 3 | 8: no lane-cooperative small kernels
    for mode, redo in (("3", "0"), ("0", "0"), ("3", "1"), ("1", "0"), ("2", "0"), ("19", "0"), ("19", "1"), ("11", "0")):
        monkeypatch.setenv("ZK_POSEIDON_MFMA", mode)
        monkeypatch.setenv("ZK_FORCE_EXACT_REDO", redo)
        c = zk.Context(0)
        try:
            b = c.batch_from_values(vals, rate_bits=rate_bits, cap_height=4)
            assert (b.cap == o.cap).all(), (mode, redo)
            assert (b.digests() == o.digests).all(), (mode, redo)
            for leaf in (0, 1, 12345 % (1 << (log_n + rate_bits)), (1 << (log_n + rate_bits)) - 1):
                assert (b.prove(leaf) == o.prove(leaf)).all()
            b.free()
        finally:
            c.close()


@pytest.mark.gpu
def test_matrix_core_poseidon_rare_repair_rows(zk, orc, monkeypatch):
    """leaves that really need the skipped borrow repair, hashed by the matrix-core kernel (rate_bits 0: leaves = the values)"""
    import json, os
    rc = [int(x, 16) for x in json.load(open(os.path.join(os.path.dirname(__file__), "golden", "poseidon_rc.json")))["round_constants_hex"]]
    rng = np.random.default_rng(77)
    log_n, ncols = 14, 19
    # a rate_bits-0 commit stores LDE = NTT(iNTT(values)) = values, in bit-reversed row order
    vals = field_rand(rng, (ncols, 1 << log_n))
    for r in range(5, 1 << log_n, 97):
        j = r % 8
        vals[j, r] = ((1 << 63) - rc[j]) % P
    o = orc.Batch(vals, rate_bits=0, cap_height=4)
    for mode in ("3", "0"):
        monkeypatch.setenv("ZK_POSEIDON_MFMA", mode)
        c = zk.Context(0)
        try:
            b = c.batch_from_values(vals, rate_bits=0, cap_height=4)
            assert (b.cap == o.cap).all() and (b.digests() == o.digests).all(), mode
            b.free()
        finally:
            c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("log_n,ncols,rate_bits", [(9, 430, 1), (8, 71, 1), (12, 93, 1), (11, 523, 1), (5, 9, 1), (10, 5, 0), (6, 135, 3)])
def test_small_commit_kernels_equal_thread_per_leaf_and_oracle(zk, orc, monkeypatch, log_n, ncols, rate_bits):
    """Commits of fewer than 16384 leaves: a wave per leaf (up to 1024 leaves), 16 lanes per leaf (up to 8192), one thread per leaf
    (ZK_POSEIDON_MFMA bit 3) -- with and without the forced exact redo -- all give the oracle's cap and digests."""
    rng = np.random.default_rng(1000 + log_n + ncols)
    vals = field_rand(rng, (ncols, 1 << log_n))
    o = orc.Batch(vals, rate_bits=rate_bits, cap_height=min(4, log_n + rate_bits))
    for mode, redo in (("3", "0"), ("11", "0"), ("3", "1")):
        monkeypatch.setenv("ZK_POSEIDON_MFMA", mode)
        monkeypatch.setenv("ZK_FORCE_EXACT_REDO", redo)
        c2 = zk.Context(0)
        try:
            b = c2.batch_from_values(vals, rate_bits=rate_bits, cap_height=min(4, log_n + rate_bits))
            assert (b.cap == o.cap).all(), (mode, redo)
            assert (b.digests() == o.digests).all(), (mode, redo)
            b.free()
        finally:
            c2.close()


def test_packed_trace_upload_widens_to_the_u64_layout(ctx):
    """zk_trace_upload_packed: columns of 1 / 2 / 4 / 8 bytes per row cross PCIe narrow and come out as the usual n_cols x n uint64 trace
    (8-byte columns canonicalised like every trace that enters); odd column orders exercise the alignment rule of the packed buffer."""
    rng = np.random.default_rng(11)
    n = 1 << 9
    cols = [rng.integers(0, 2, n).astype(np.uint8),                       # a flag
            rng.integers(0, 1 << 16, n).astype(np.uint16),
            rng.integers(0, 256, n).astype(np.uint8),
            rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32),
            rng.integers(0, 1 << 63, n, dtype=np.uint64) * np.uint64(2) + np.uint64(1),     # full width, some values >= p
            rng.integers(0, 256, n).astype(np.uint8),
            rng.integers(0, 1 << 63, n, dtype=np.uint64)]
    cols[4][:3] = [0xFFFFFFFF00000001, 0xFFFFFFFFFFFFFFFF, 0xFFFFFFFF00000000]
    d = ctx.upload_packed(cols)
    got = ctx.download(d, (len(cols), n))
    ctx.dev_free(d)
    P = np.uint64(0xFFFFFFFF00000001)
    for j, c in enumerate(cols):
        want = c.astype(np.uint64)
        want = np.where(want >= P, want - P, want)
        assert (got[j] == want).all(), j


def test_packed_upload_from_pinned_memory_then_a_proof_straight_away(zk, orc):
    """zk_trace_upload_packed returns before its copies have run (pinned source).  The column offset table used to sit in the
    per-proof staging buffer, which the next zk_prove_tables on the context resets and overwrites: on a FRESH context the table is at
    staging offset 0, right where the proof's header words go.  Here: fresh context, pinned packed source, no synchronize, prove at
    once through the C ABI -- the proof must be the oracle's proof of the widened trace."""
    import ctypes as C
    import eth_tx_proof_amd as E
    from eth_tx_proof_amd import air
    L = zk.lib()
    log_n, n_cols = 10, 9
    tr, pub = orc.synth_trace(n_cols, log_n, 5)
    cfg = E.standard_fast_config()
    prog = air.cpu_like_air(n_cols, 2)
    rc, ref = orc.prove(cfg, [(prog, log_n, tr, pub)])
    assert rc == 0
    for _ in range(3):
        c = zk.Context(0)
        try:
            src = zk.pinned_empty(c, tr.shape)
            src[...] = tr
            widths = np.full(n_cols, 8, dtype=np.uint8)
            p = C.c_void_p()
            c._chk(L.zk_trace_upload_packed(c.h, src.ctypes.data_as(C.c_void_p), C.c_uint32(log_n), C.c_uint32(n_cols),
                                            widths.ctypes.data_as(C.POINTER(C.c_uint8)), C.byref(p)), "zk_trace_upload_packed")
            proof = zk.prove_tables(c, cfg, [(prog, log_n, p.value, pub)])          # no synchronize in between
            assert proof.size == ref.size and (proof == ref).all()
            c.dev_free(p.value)
            del src
        finally:
            c.close()


@pytest.mark.gpu
def test_fault_recorder_names_the_owner_of_an_address(zk, capfd):
    """csrc/fault_recorder.hip: the report the library's handler writes on a GPU memory fault, produced here WITHOUT a fault
    (zk_debug_fault_report) for an address inside a block this context holds: the runtime's view (a live allocation), the library's own
    record (tag, offset, still held) and the context's last launches must all be there."""
    import ctypes as C
    if os.environ.get("ZK_ALLOC_EFENCE"):
        pytest.skip("the debug allocators place blocks through the virtual-memory API: other tags, another runtime pointer type")
    c = zk.Context(0)
    try:
        nbytes = 3 << 20                             # a driver allocation of its own (2 MiB and more are never sub-allocated)
        d = c.dev_alloc(nbytes)
        c.upload(d, np.arange(1024, dtype=np.uint64))
        c.two_to_one(np.zeros((4, 4), dtype=np.uint64), np.ones((4, 4), dtype=np.uint64))     # any launch: something for the launch ring
        addr = int(d) + 4096 + 8
        f = zk.lib().zk_debug_fault_report
        f.argtypes, f.restype = [C.c_uint64], C.c_int32
        capfd.readouterr()
        assert f(addr) == 0, "the handler is not registered"
        err = capfd.readouterr().err
        assert "[zk-fault] GPU memory fault at 0x%x" % addr in err
        assert "runtime allocation" in err and "offset 4104" in err
        assert "library record: address inside dev-block" in err and "still held" in err
        assert "last launches" in err and "[" in err.split("last launches")[1]
        c.dev_free(d)
    finally:
        c.close()


def test_ntt_persistent_tile_loop_gives_the_same_commitment(orc):
    """ZK_NTT_PERSIST (opt-in, measured slower: DESIGN.md section 8): a fixed grid of workgroups walking the tiles of a pass -- the tile loop
    is in every register-radix kernel, so it must be exact when switched on: a 2^17 x 5 commit (two passes each way, 80 tiles per pass over a
    grid of 16) in a child process against the oracle's cap and opened rows, forced exact redo included."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import eth_tx_proof_amd
from eth_tx_proof_amd import lib as zl
import orc
rng = np.random.default_rng(9)
v = rng.integers(0, 0xFFFFFFFF00000001, size=(5, 1 << 17), dtype=np.uint64)
o = orc.Batch(v, rate_bits=1, cap_height=4)
c = zl.Context(0)
b = c.batch_from_values(v, rate_bits=1, cap_height=4)
assert (b.cap == o.cap).all()
for i in (0, 77, (1 << 18) - 1):
    assert (b.prove(i) == o.prove(i)).all()
f = c.ntt(v[:2]); assert (f == orc.fft(v[:2])).all() and (c.ntt(f, inverse=True) == v[:2]).all()
print("PERSIST_OK")
''' % (root, os.path.join(root, "oracle"))
    for extra in ({}, {"ZK_FORCE_EXACT_REDO": "1"}):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, ZK_NTT_PERSIST="16", **extra), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "PERSIST_OK" in r.stdout, r.stderr[-3000:]
