"""GPU parity for the full unit proof: proof words from the HIP prover == proof words from the CPU oracle (bit exact),
and the oracle's independent verifier accepts them."""
import numpy as np
import pytest

from conftest import P, field_rand

pytestmark = pytest.mark.gpu


def _cfg():
    import eth_tx_proof_amd as E
    return E.standard_fast_config()


@pytest.mark.parametrize("n_cols,n_helpers,log_n", [(6, 2, 6), (7, 0, 5), (5, 1, 10), (93, 19, 8), (9, 3, 12), (21, 4, 13), (1, 0, 4), (12, 9, 9)])
def test_proof_bit_exact_vs_oracle(ctx, zk, orc, n_cols, n_helpers, log_n):
    from eth_tx_proof_amd import air
    cfg = _cfg()
    prog = air.cpu_like_air(n_cols, n_helpers)
    tr, pub = orc.synth_trace(n_cols, log_n, 0x5EED0000 + log_n)
    proof = zk.prove_tables(ctx, cfg, [(prog, log_n, tr, pub)])
    rc, ref = orc.prove(cfg, [(prog, log_n, tr, pub)])
    assert rc == 0
    assert proof.size == ref.size == orc.proof_size_words(cfg, [(prog, log_n, tr, pub)])
    if not (proof == ref).all():
        bad = np.nonzero(proof != ref)[0]
        pytest.fail("first differing word %d of %d (%d differ)" % (bad[0], proof.size, bad.size))
    assert (proof[2:] < np.uint64(P)).all()
    assert orc.verify(cfg, [(prog, log_n, None, pub)], proof) == 0


SWEEP = [(1, 0, 3), (2, 0, 3), (3, 1, 4), (8, 2, 5), (16, 4, 6), (17, 4, 7), (40, 10, 9), (135, 19, 7), (200, 19, 8), (33, 8, 11), (64, 16, 12),
         (4, 1, 15), (9, 2, 16), (5, 0, 17), (165, 19, 10), (71, 17, 9), (21, 5, 14), (430, 19, 6), (523, 19, 5), (2431, 19, 4)]


@pytest.mark.parametrize("n_cols,n_helpers,log_n", SWEEP)
def test_proof_shape_sweep_bit_exact(ctx, zk, orc, n_cols, n_helpers, log_n):
    """Shapes around every edge the prover has: the smallest tree (2^3 rows: the LDE is exactly the cap), one column, widths that are
    not a multiple of the sponge rate, no auxiliary columns, the zkEVM table widths (up to 2 431 columns), tall narrow tables whose
    Merkle and FRI layers cross every kernel-selection threshold.  Whole proof equal to the oracle's, and the verifier accepts it."""
    from eth_tx_proof_amd import air
    cfg = _cfg()
    prog = air.cpu_like_air(n_cols, n_helpers)
    tr, pub = orc.synth_trace(n_cols, log_n, 0xABC0000 + 31 * n_cols + log_n)
    proof = zk.prove_tables(ctx, cfg, [(prog, log_n, tr, pub)])
    rc, ref = orc.prove(cfg, [(prog, log_n, tr, pub)])
    assert rc == 0 and proof.size == ref.size
    if not (proof == ref).all():
        bad = np.nonzero(proof != ref)[0]
        pytest.fail("first differing word %d of %d (%d differ)" % (bad[0], proof.size, bad.size))
    assert orc.verify(cfg, [(prog, log_n, None, pub)], proof) == 0


CFG_SWEEP = [(100, 1, 1, 0, 0, 2, 2, 5), (100, 3, 1, 2, 10, 3, 4, 20), (100, 2, 1, 6, 8, 5, 3, 40), (100, 4, 1, 4, 12, 4, 5, 10), (100, 2, 1, 8, 4, 6, 1, 7),
             (100, 2, 1, 3, 20, 4, 0, 3), (100, 2, 1, 4, 16, 4, 5, 84), (100, 2, 1, 5, 1, 2, 6, 9)]


@pytest.mark.parametrize("c", CFG_SWEEP)
def test_stark_config_sweep_bit_exact_or_refused_alike(ctx, zk, orc, c):
    """StarkConfig away from standard_fast_config: 1 to 4 challenges, cap heights 0 to 8, no proof of work up to 20 bits, FRI arities
    4 to 64, final polynomials of 1 to 64 coefficients, 3 to 84 query rounds.  Whatever the oracle does with a (config, table) pair
    the GPU does too: the same proof words, or both refuse.  (The synthetic AIR's auxiliary columns are laid out for two challenges,
    so tables with helper columns appear under two challenges only.)"""
    from eth_tx_proof_amd import air
    from eth_tx_proof_amd._abi import StarkConfig
    cfg = StarkConfig(*c)
    shapes = [(12, 0, 11), (5, 0, 6)] + ([(7, 2, 8), (21, 4, 10)] if cfg.num_challenges == 2 else [])
    for n_cols, n_helpers, log_n in shapes:
        prog = air.cpu_like_air(n_cols, n_helpers)
        tr, pub = orc.synth_trace(n_cols, log_n, 0xC0F + log_n)
        rc, ref = orc.prove(cfg, [(prog, log_n, tr, pub)])
        if rc != 0:
            with pytest.raises(zk.ZkError):
                zk.prove_tables(ctx, cfg, [(prog, log_n, tr, pub)])
            continue
        proof = zk.prove_tables(ctx, cfg, [(prog, log_n, tr, pub)])
        assert proof.size == ref.size and (proof == ref).all(), (c, n_cols, n_helpers, log_n)
        assert orc.verify(cfg, [(prog, log_n, None, pub)], proof) == 0


def test_multi_table_bit_exact(ctx, zk, orc):
    """cfg4 shape at reduced sizes: several tables, one challenger, tables proved in order"""
    from eth_tx_proof_amd import air
    cfg = _cfg()
    items = []
    for t, (w, lg, h) in enumerate(((165, 9, 8), (71, 5, 3), (93, 8, 19), (21, 11, 2))):
        prog = air.cpu_like_air(w, h)
        tr, pub = orc.synth_trace(w, lg, 0x5EED0000 + 1000 * t)
        items.append((prog, lg, tr, pub))
    proof = zk.prove_tables(ctx, cfg, items)
    rc, ref = orc.prove(cfg, items)
    assert rc == 0 and (proof == ref).all()
    assert orc.verify(cfg, [(p, lg, None, pub) for p, lg, _, pub in items], proof) == 0


def test_synth_trace_dev_matches_oracle(ctx, zk, orc):
    for n_cols, log_n in ((93, 10), (7, 6), (2, 5)):
        d, pub = zk.synth_trace_dev(ctx, n_cols, log_n, 1234)
        got = ctx.download(d, (n_cols, 1 << log_n))
        ref, rpub = orc.synth_trace(n_cols, log_n, 1234)
        assert (got == ref).all() and (pub == rpub).all()
        ctx.dev_free(d)


def test_device_resident_trace_and_determinism(ctx, zk, orc):
    from eth_tx_proof_amd import air
    cfg = _cfg()
    prog = air.cpu_like_air(93, 19)
    d, pub = zk.synth_trace_dev(ctx, 93, 11, 99)
    p1 = zk.prove_tables(ctx, cfg, [(prog, 11, d, pub)]).copy()
    p2 = zk.prove_tables(ctx, cfg, [(prog, 11, d, pub)]).copy()
    assert (p1 == p2).all()
    assert orc.verify(cfg, [(prog, 11, None, pub)], p1) == 0
    tr = ctx.download(d, (93, 1 << 11))
    rc, ref = orc.prove(cfg, [(prog, 11, tr, pub)])
    assert rc == 0 and (ref == p1).all()
    ctx.dev_free(d)


def test_unsatisfied_trace_proof_is_rejected(ctx, zk, orc):
    from eth_tx_proof_amd import air
    cfg = _cfg()
    prog = air.cpu_like_air(6, 2)
    tr, pub = orc.synth_trace(6, 8, 11)
    tr[2, 17] ^= np.uint64(1)
    proof = zk.prove_tables(ctx, cfg, [(prog, 8, tr, pub)])
    assert orc.verify(cfg, [(prog, 8, None, pub)], proof) != 0


def test_full_size_unit_proof_verifies(ctx, zk, orc):
    """BASELINE cfg3 at a size the verifier handles in seconds is 2^16; the 2^20 shape is exercised by bench.py and checked
    there through the verifier as well (size-independent property: the oracle's verifier accepts the proof)."""
    from eth_tx_proof_amd import air
    cfg = _cfg()
    prog = air.cpu_like_air(93, 19)
    d, pub = zk.synth_trace_dev(ctx, 93, 16, 7)
    proof = zk.prove_tables(ctx, cfg, [(prog, 16, d, pub)])
    assert orc.verify(cfg, [(prog, 16, None, pub)], proof) == 0
    ctx.dev_free(d)


def test_error_paths(ctx, zk):
    from eth_tx_proof_amd import air
    import eth_tx_proof_amd as E
    cfg = E.standard_fast_config()
    cfg.rate_bits = 2
    prog = air.cpu_like_air(6, 0)
    tr = np.zeros((6, 64), dtype=np.uint64)
    with pytest.raises(zk.ZkError) as e:
        zk.prove_tables(ctx, cfg, [(prog, 6, tr, np.zeros(3, dtype=np.uint64))])
    assert e.value.code == -1


def test_cfg4_seven_table_shapes(ctx, zk, orc):
    """BASELINE configs[3]: seven tables at log sizes 15/9/12/14/9/12/17 with the recalled zkEVM widths
    (165/71/93/2431/430/523/21), one shared challenger; accepted by the oracle's verifier."""
    from eth_tx_proof_amd import air
    cfg = _cfg()
    items, ptrs = [], []
    for t, (w, lg) in enumerate(zip(air.ZKEVM_TABLE_WIDTHS, air.ZKEVM_CFG4_LOG_SIZES)):
        prog = air.cpu_like_air(w, min(19, w // 4))
        d, pub = zk.synth_trace_dev(ctx, w, lg, 0x5EED0000 + 1000 * t)
        items.append((prog, lg, d, pub)); ptrs.append(d)
    proof = zk.prove_tables(ctx, cfg, items)
    assert orc.verify(cfg, [(p, lg, None, pub) for p, lg, _, pub in items], proof) == 0
    bad = proof.copy(); bad[proof.size // 3] ^= np.uint64(1)
    assert orc.verify(cfg, [(p, lg, None, pub) for p, lg, _, pub in items], bad) != 0
    for d in ptrs:
        ctx.dev_free(d)


def test_concurrent_contexts_are_independent(zk, orc):
    """paladin `--num-workers N` shape: N host threads, one zk_ctx (one HIP stream) each, proving different units at the
    same time; every proof is bit-identical to the oracle's."""
    import threading
    from eth_tx_proof_amd import air
    cfg = _cfg()
    prog = air.cpu_like_air(12, 3)
    jobs = [orc.synth_trace(12, 9 + (k % 2), 100 + k) for k in range(4)]
    outs, errs = [None] * 4, []

    def work(k):
        try:
            c = zk.Context(0)
            tr, pub = jobs[k]
            for _ in range(3):
                outs[k] = zk.prove_tables(c, cfg, [(prog, 9 + (k % 2), tr, pub)]).copy()
            c.close()
        except Exception as e:
            errs.append(e)
    th = [threading.Thread(target=work, args=(k,)) for k in range(4)]
    [t.start() for t in th]; [t.join() for t in th]
    assert not errs, errs
    for k in range(4):
        tr, pub = jobs[k]
        rc, ref = orc.prove(cfg, [(prog, 9 + (k % 2), tr, pub)])
        assert rc == 0 and (outs[k] == ref).all()


def test_proof_buffer_too_small_and_repeatability(ctx, zk, orc):
    import ctypes as C
    from eth_tx_proof_amd import air
    cfg = _cfg()
    prog = air.cpu_like_air(6, 2)
    tr, pub = orc.synth_trace(6, 7, 3)
    n = zk.proof_size_words(cfg, [(prog, 7, tr, pub)])
    small = np.empty(n - 1, dtype=np.uint64)
    with pytest.raises(zk.ZkError) as e:
        zk.prove_tables(ctx, cfg, [(prog, 7, tr, pub)], out=small)
    assert e.value.code == -9
    # the context stays usable after an error, and 30 proofs in a row give the same words (no state leaks between calls)
    first = zk.prove_tables(ctx, cfg, [(prog, 7, tr, pub)]).copy()
    for _ in range(30):
        assert (zk.prove_tables(ctx, cfg, [(prog, 7, tr, pub)]) == first).all()


def test_baseline_size_unit_proof_verifies_and_openings_are_consistent(ctx, zk, orc):
    """BASELINE cfg3 at full size (2^20 rows x 93 + 40 columns): too large for the oracle's prover inside a test, so the
    proof is checked through size-independent properties -- the oracle's verifier accepts it (constraints at zeta, FRI
    low-degree test, every Merkle path), and a single-word corruption anywhere is rejected."""
    from eth_tx_proof_amd import air
    cfg = _cfg()
    prog = air.cpu_like_air(93, 19)
    d, pub = zk.synth_trace_dev(ctx, 93, 20, 0x5EED0000)
    proof = zk.prove_tables(ctx, cfg, [(prog, 20, d, pub)])
    assert proof.size == zk.proof_size_words(cfg, [(prog, 20, d, pub)])
    assert (proof[2:] < np.uint64(P)).all()
    assert orc.verify(cfg, [(prog, 20, None, pub)], proof) == 0
    rng = np.random.default_rng(20)
    for pos in rng.integers(5, proof.size, size=6):
        bad = proof.copy(); bad[pos] ^= np.uint64(1)
        assert orc.verify(cfg, [(prog, 20, None, pub)], bad) != 0
    ctx.dev_free(d)


def test_mid_size_bit_exact(ctx, zk, orc):
    """2^14 rows x 93 columns: the largest size at which the oracle's prover still finishes in seconds."""
    from eth_tx_proof_amd import air
    cfg = _cfg()
    prog = air.cpu_like_air(93, 19)
    tr, pub = orc.synth_trace(93, 14, 4242)
    proof = zk.prove_tables(ctx, cfg, [(prog, 14, tr, pub)])
    rc, ref = orc.prove(cfg, [(prog, 14, tr, pub)])
    assert rc == 0 and (proof == ref).all()


@pytest.mark.parametrize("log_n", [18, 20, 21])
def test_baseline_shape_bit_exact_vs_oracle_up_to_full_size(ctx, zk, orc, log_n):
    """BASELINE configs[2] itself -- 93 trace + 40 aux columns, standard_fast_config -- at 2^18 rows, at the FULL 2^20 rows and one size
    up (2^21: the NTT passes change shape): the GPU proof (trace generated in HBM) equals the oracle's proof of the same trace
    (generated on the host from the same seed) word for word: every cap, opening, FRI layer, the proof-of-work witness and all 84
    query rounds.  (2^23, north_star's target, is beyond the oracle's prover in a test: verifier-checked below.)"""
    from eth_tx_proof_amd import air
    cfg = _cfg()
    prog = air.cpu_like_air(93, 19)
    seed = 0x5EED0000 + log_n
    d, pub = zk.synth_trace_dev(ctx, 93, log_n, seed)
    proof = zk.prove_tables(ctx, cfg, [(prog, log_n, d, pub)])
    ctx.dev_free(d)
    tr, opub = orc.synth_trace(93, log_n, seed)
    assert (opub == pub).all()
    rc, ref = orc.prove(cfg, [(prog, log_n, tr, opub)])
    del tr
    assert rc == 0 and proof.size == ref.size
    if not (proof == ref).all():
        bad = np.nonzero(proof != ref)[0]
        pytest.fail("first differing word %d of %d (%d differ)" % (bad[0], proof.size, bad.size))


def test_2_17_rows_narrow_table_bit_exact_both_permutation_paths(zk, orc, monkeypatch):
    """2^17 rows x 9 columns: the first FRI layer has 2^14 leaves, so the matrix-core FRI-leaf kernel, leaf kernel and Merkle-level kernel
    all take part (and, with ZK_POSEIDON_MFMA=0, none of them); both proofs must equal the oracle's word for word."""
    from eth_tx_proof_amd import air
    cfg = _cfg()
    prog = air.cpu_like_air(9, 3)
    tr, pub = orc.synth_trace(9, 17, 99)
    rc, ref = orc.prove(cfg, [(prog, 17, tr, pub)])
    assert rc == 0
    for mode in ("3", "0"):
        monkeypatch.setenv("ZK_POSEIDON_MFMA", mode)
        c = zk.Context(0)
        try:
            proof = zk.prove_tables(c, cfg, [(prog, 17, tr, pub)])
            assert (proof == ref).all(), mode
        finally:
            c.close()


def test_four_streams_full_size_units_equal_their_single_stream_proofs(zk, orc):
    """bench.py's shape as a test: four contexts (= four HIP streams), one host thread each, proving 2^20-row units concurrently, twice.
    Every proof must equal the proof of the same unit made alone, and verify.  (An earlier generation of the compiled constraint
    kernels ended in GPU memory faults in exactly this setting and nowhere in the test-suite: DESIGN.md section 9.)"""
    import threading
    from eth_tx_proof_amd import air
    cfg = _cfg()
    prog = air.cpu_like_air(93, 19)
    log_n, T = 20, 4
    ctxs = [zk.Context(0) for _ in range(T)]
    try:
        units = [zk.synth_trace_dev(ctxs[i], 93, log_n, 0x5EED0000 + 1000 * i) for i in range(T)]
        alone = [zk.prove_tables(ctxs[0], cfg, [(prog, log_n, units[i][0], units[i][1])]).copy() for i in range(T)]
        for c in ctxs:
            c.synchronize()
        got, errs = {}, []

        def work(i, r):
            try:
                got[(i, r)] = zk.prove_tables(ctxs[i], cfg, [(prog, log_n, units[i][0], units[i][1])]).copy()
            except Exception as e:      # surfaced after join
                errs.append(e)
        for r in range(2):
            th = [threading.Thread(target=work, args=(i, r)) for i in range(T)]
            for t in th:
                t.start()
            for t in th:
                t.join()
        assert not errs, errs
        for (i, r), p in got.items():
            assert (p == alone[i]).all(), (i, r)
        assert orc.verify(cfg, [(prog, log_n, None, units[0][1])], alone[0]) == 0
    finally:
        for c in ctxs:
            c.close()


def test_north_star_target_2_23_rows_unit_proof_verifies(zk, orc):
    """north_star's target configuration: ONE 2^23-row x 93 (+40) column unit proof on one GPU (about 45 GB resident).  Far beyond
    what the oracle's prover can do inside a test, so parity is checked through the size-independent properties: the oracle's
    verifier (constraints at zeta, FRI low-degree test over 2^24 points, every Merkle path) accepts the proof, rejects it after a
    single flipped word anywhere, and a second run gives the same words (determinism of the smallest PoW witness)."""
    from eth_tx_proof_amd import air
    cfg = _cfg()
    prog = air.cpu_like_air(93, 19)
    c = zk.Context(0)
    try:
        d, pub = zk.synth_trace_dev(c, 93, 23, 0x5EED0023)
        items = [(prog, 23, d, pub)]
        proof = zk.prove_tables(c, cfg, items).copy()
        assert proof.size == zk.proof_size_words(cfg, items)
        assert (proof[2:] < np.uint64(P)).all()
        assert orc.verify(cfg, [(prog, 23, None, pub)], proof) == 0
        rng = np.random.default_rng(23)
        for pos in rng.integers(5, proof.size, size=4):
            bad = proof.copy(); bad[pos] ^= np.uint64(1)
            assert orc.verify(cfg, [(prog, 23, None, pub)], bad) != 0
        assert (zk.prove_tables(c, cfg, items) == proof).all()
    finally:
        c.close()


def test_host_resident_traces_pinned_pageable_and_prefetched_give_the_same_proof(zk, orc):
    """The PCIe-inclusive entry: a host-resident trace in pinned memory (zk_host_alloc: one asynchronous copy), in pageable memory
    (staged through the context's two pinned chunks; 2^17 x 93 x 8 B = 97 MB spans several 32 MB chunks) and prefetched on the
    copy stream (zk_trace_upload_async) must all give the proof of the device-resident trace, which the oracle's verifier accepts."""
    from eth_tx_proof_amd import air
    cfg = _cfg()
    prog = air.cpu_like_air(93, 19)
    log_n = 17
    c = zk.Context(0)
    try:
        d, pub = zk.synth_trace_dev(c, 93, log_n, 0x5EED0017)
        ref = zk.prove_tables(c, cfg, [(prog, log_n, d, pub)]).copy()
        assert orc.verify(cfg, [(prog, log_n, None, pub)], ref) == 0
        pageable = c.download(d, (93, 1 << log_n))
        pinned = zk.pinned_empty(c, (93, 1 << log_n))
        pinned[...] = pageable
        assert (zk.prove_tables(c, cfg, [(prog, log_n, pageable, pub)]) == ref).all()
        assert (zk.prove_tables(c, cfg, [(prog, log_n, pinned, pub)]) == ref).all()
        # single-worker double buffering: upload unit k+1 while unit k is proved
        d2 = zk.trace_upload_async(c, pinned)
        p_a = zk.prove_tables(c, cfg, [(prog, log_n, pageable, pub)]).copy()
        d3 = zk.trace_upload_async(c, pageable)
        p_b = zk.prove_tables(c, cfg, [(prog, log_n, d2, pub)]).copy()
        p_c = zk.prove_tables(c, cfg, [(prog, log_n, d3, pub)]).copy()
        assert (p_a == ref).all() and (p_b == ref).all() and (p_c == ref).all()
        c.dev_free(d2); c.dev_free(d3); c.dev_free(d)
        del pinned
    finally:
        c.close()


@pytest.mark.parametrize("n_looked,filtered,log_n", [(5, (1, 3), 7), (4, (), 10), (1, (0,), 6), (7, (0, 2, 6), 13)])
def test_lookup_argument_columns_and_proof_bit_exact(ctx, zk, orc, n_looked, filtered, log_n):
    """starky::lookup on the GPU: helper columns with filters (batches of two, batched inversion), the forward
    running sum with the multiplicity column, in front of the CTL-shaped columns: zk_aux_columns == the oracle's, and the whole
    unit proof (interpreter and, at 2^13 rows, the compiled constraint kernel path) equals the oracle's word for word."""
    from eth_tx_proof_amd import air
    cfg = _cfg()
    prog, tr, pub = air.lookup_air(n_looked, filtered, log_n=log_n, seed=log_n)
    ch = np.array([5, 6, 7, 8], dtype=np.uint64)
    rc, ref_aux = orc.aux_columns(prog, ch, tr)
    assert rc == 0
    assert (zk.aux_columns(ctx, prog, ch, tr) == ref_aux).all()
    proof = zk.prove_tables(ctx, cfg, [(prog, log_n, tr, pub)])
    rc, ref = orc.prove(cfg, [(prog, log_n, tr, pub)])
    assert rc == 0 and proof.size == ref.size and (proof == ref).all()
    assert orc.verify(cfg, [(prog, log_n, None, pub)], proof) == 0
    bad = tr.copy(); bad[1, 3] += np.uint64(1)
    assert orc.verify(cfg, [(prog, log_n, None, pub)], zk.prove_tables(ctx, cfg, [(prog, log_n, bad, pub)])) != 0


def test_cfg4_all_seven_tables_at_their_real_sizes_bit_exact(ctx, zk, orc):
    """BASELINE configs[3] in full: Arithmetic (165 x 2^15), BytePacking (71 x 2^9), Cpu (93 x 2^12), Keccak (2431 x 2^14),
    KeccakSponge (430 x 2^9), Logic (523 x 2^12), Memory (21 x 2^17) -- recalled zkEVM widths, all seven on one challenger in table
    order: the GPU proof equals the oracle's word for word, and the oracle's verifier accepts it."""
    from eth_tx_proof_amd import air
    cfg = _cfg()
    items = []
    for t, (w, lg) in enumerate(zip(air.ZKEVM_TABLE_WIDTHS, air.ZKEVM_CFG4_LOG_SIZES)):
        prog = air.cpu_like_air(w, min(19, w // 4))
        tr, pub = orc.synth_trace(w, lg, 0x5EED0000 + 1000 * t)
        items.append((prog, lg, tr, pub))
    proof = zk.prove_tables(ctx, cfg, items)
    rc, ref = orc.prove(cfg, items)
    assert rc == 0 and proof.size == ref.size
    if not (proof == ref).all():
        bad = np.nonzero(proof != ref)[0]
        pytest.fail("first differing word %d of %d" % (bad[0], proof.size))
    assert orc.verify(cfg, [(p, lg, None, pub) for p, lg, _, pub in items], proof) == 0


def test_cfg4_with_prepared_constraint_kernels_bit_exact(zk, orc):
    """The seven-table unit again, this time with the prover state prepared (zk_prover_state_prepare: hiprtc compiles the constraint
    kernels of the three wide tables -- 430, 523 and 2 431 columns, sequences of 3 / 4 / 14 call-free kernels, about a minute and a half
    when nothing is cached; the prover-state directory of the tree (eth-tx-proof_amd/prover_state) is used when it holds them -- the
    narrow ones were built with the library): every table's quotient comes from compiled code, and the proof is still the
    oracle's, word for word."""
    from eth_tx_proof_amd import air
    cfg = _cfg()
    c2 = zk.Context(0)
    try:
        items = []
        for t, (w, lg) in enumerate(zip(air.ZKEVM_TABLE_WIDTHS, air.ZKEVM_CFG4_LOG_SIZES)):
            prog = air.cpu_like_air(w, min(19, w // 4))
            tr, pub = orc.synth_trace(w, lg, 0x5EED0000 + 1000 * t)
            items.append((prog, lg, tr, pub))
        import os
        state_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "eth-tx-proof_amd", "prover_state")
        if os.path.isdir(state_dir):
            zk.prover_state_cache_dir(state_dir)
        try:
            zk.prover_state_prepare(c2, cfg, items)
        finally:
            zk.prover_state_cache_dir(None)
        assert sum(zk.prover_state_stats()) >= 3            # compiled now, or loaded from the prover-state directory
        c2.profile(True); c2.profile_reset()
        proof = zk.prove_tables(c2, cfg, items)
        assert c2.profile_report()["quotient_eval"][0] == 7
        rc, ref = orc.prove(cfg, items)
        assert rc == 0 and proof.size == ref.size and (proof == ref).all()
    finally:
        c2.close()


def test_host_side_challenger_gives_the_same_proofs(zk, orc):
    """zk_ctx_set_host_transcript: the duplex sponge on a host thread (observed words by asynchronous D2H copies, challenges back behind a
    stream-side wait, no host synchronisation of the proving thread) -- the same proof words as the device-resident transcript and as the
    oracle: a STARK unit with lookups + CTL helpers, a two-table unit, the challenger API itself, and the switch flipped back and forth
    between proofs of one context."""
    from eth_tx_proof_amd import air
    cfg = _cfg()
    c2 = zk.Context(0)
    try:
        prog = air.cpu_like_air(21, 5)
        tr, pub = orc.synth_trace(21, 10, 77)
        prog2 = air.cpu_like_air(9, 2)
        tr2, pub2 = orc.synth_trace(9, 8, 78)
        items = [(prog, 10, tr, pub), (prog2, 8, tr2, pub2)]
        rc, ref = orc.prove(cfg, items)
        assert rc == 0
        dev = zk.prove_tables(c2, cfg, items)
        assert (dev == ref).all()
        for rep in range(2):
            c2.set_host_transcript(True)
            host = zk.prove_tables(c2, cfg, items)
            assert host.size == ref.size and (host == ref).all(), "host-side challenger: another proof"
            # the challenger object of the C ABI on the host path: a long observe (several rate-sized chunks + a ragged tail), squeezes in between
            ch, och = c2.challenger(), orc.Challenger()
            xs = np.array([(i * 0x9E3779B97F4A7C15) % 0xFFFFFFFF00000001 for i in range(1, 1 + 8 * 37 + 5)], dtype=np.uint64)
            ch.observe(xs[:100]); och.observe(xs[:100])
            assert (ch.get(3) == och.get(3)).all()
            ch.observe(xs[100:]); och.observe(xs[100:])
            assert (ch.get(9) == och.get(9)).all() and (ch.compact() == och.compact()).all()
            c2.set_host_transcript(False)
            assert (zk.prove_tables(c2, cfg, items) == ref).all()
        # more observed words than the 16 MiB pinned ring of the mailbox holds (2.2 M words in 4 096-word calls): the ring is drained and
        # started over in the middle of the stream of jobs, the sponge must not notice
        c2.set_host_transcript(True)
        rng = np.random.default_rng(4)
        big = rng.integers(0, 0xFFFFFFFF00000001, size=(1 << 21) + (1 << 17) + 3, dtype=np.uint64)
        ch, och = c2.challenger(), orc.Challenger()
        ch.observe(big); och.observe(big)
        assert (ch.get(5) == och.get(5)).all() and (ch.compact() == och.compact()).all()
        c2.set_host_transcript(False)
    finally:
        c2.close()


@pytest.mark.gpu
def test_a_challenger_that_outlives_the_switch_keeps_its_sponge(zk, orc):
    """zk_ctx_set_host_transcript moves the sponge of every live zk_challenger between its two homes (DevChallenger block <->
    host map), in every buffer state: pending inputs, unread outputs, right after a permutation.  One oracle challenger is the reference
    for a handle that crosses the switch six times, with proofs of the same context in between; a second handle is created on the host side
    and freed there (its map entry must go with it)."""
    from eth_tx_proof_amd import air
    cfg = _cfg()
    c2 = zk.Context(0)
    try:
        prog = air.cpu_like_air(9, 2)
        tr, pub = orc.synth_trace(9, 8, 5)
        items = [(prog, 8, tr, pub)]
        rc, ref = orc.prove(cfg, items)
        assert rc == 0
        xs = np.array([(i * 0xD1B54A32D192ED03) % 0xFFFFFFFF00000001 for i in range(1, 400)], dtype=np.uint64)
        ch, och = c2.challenger(), orc.Challenger()
        at = 0
        def step(n_obs, n_get):
            nonlocal at
            if n_obs:
                ch.observe(xs[at:at + n_obs]); och.observe(xs[at:at + n_obs]); at += n_obs
            if n_get:
                assert (ch.get(n_get) == och.get(n_get)).all()
        step(5, 0)                                   # pending inputs cross to the host
        c2.set_host_transcript(True)
        step(3, 2)                                   # completes the chunk there; 6 outputs left unread
        c2.set_host_transcript(False)
        step(0, 3)                                   # ... and are read on the device
        assert (zk.prove_tables(c2, cfg, items) == ref).all()
        step(11, 0)
        c2.set_host_transcript(True)
        assert (zk.prove_tables(c2, cfg, items) == ref).all()
        other = c2.challenger(); other.observe(xs[:9]); oo = orc.Challenger(); oo.observe(xs[:9])
        assert (other.get(2) == oo.get(2)).all()
        other.free()
        step(0, 9)                                   # more than one buffer of challenges
        c2.set_host_transcript(False)
        step(8, 1)
        c2.set_host_transcript(True)
        c2.set_host_transcript(False)
        step(1, 1)
        assert (ch.compact() == och.compact()).all()
        c2.set_host_transcript(True)
        step(2, 2)
        assert (ch.compact() == och.compact()).all()
        c2.set_host_transcript(False)
    finally:
        c2.close()


@pytest.mark.gpu
def test_every_block_across_a_4gib_boundary_gives_the_same_proofs(zk, orc, monkeypatch):
    """The layout behind the GPU memory faults of rounds 1-4: a device block that straddles a 4 GiB boundary, read or
    written behind the boundary by a kernel whose pointer arithmetic had lost its carry (an asm block overwrote SCC between s_add_u32 and
    s_addc_u32).  ZK_ALLOC_EFENCE=2 places EVERY block of a context across such a boundary: transforms, commitments, a STARK unit with
    lookups through the build-time constraint kernel, a unit through the interpreter, and a Plonk proof with the compiled recursion gate
    program must still be the oracle's, bit for bit.  (A regression here is a GPU memory fault, not a wrong word: the library's handler
    reports it and ends the process.)"""
    import eth_tx_proof_amd as E
    from eth_tx_proof_amd import air, plonk
    monkeypatch.setenv("ZK_ALLOC_EFENCE", "2")
    c2 = zk.Context(0)
    try:
        rng = np.random.default_rng(44)
        for log_n, ncols in ((12, 5), (16, 9), (18, 3)):              # one pass, two passes, coset-merged and plain LDE shapes
            vals = field_rand(rng, (ncols, 1 << log_n))
            assert (c2.ntt(vals) == orc.fft(vals)).all() and (c2.ntt(vals, inverse=True) == orc.fft(vals, inverse=True)).all()
            b, o = c2.batch_from_values(vals, rate_bits=1, cap_height=4), orc.Batch(vals, rate_bits=1, cap_height=4)
            assert (b.cap == o.cap).all() and (b.digests() == o.digests).all()
            b.free()
        cfg = E.standard_fast_config()
        for w, h, lg, seed in ((93, 19, 12, 5), (21, 5, 10, 6)):      # 93 x 19 helpers: a kernel built with the library
            prog = air.cpu_like_air(w, h)
            tr, pub = orc.synth_trace(w, lg, seed)
            rc, ref = orc.prove(cfg, [(prog, lg, tr, pub)])
            assert rc == 0 and (zk.prove_tables(c2, cfg, [(prog, lg, tr, pub)]) == ref).all()
        rcfg = E.standard_recursion_config()
        circ, wires = plonk.synthetic_circuit(degree_bits=12, seed=3)
        pubs = np.array([12, 3, 99], dtype=np.uint64)
        ost, st = orc.PlonkState(rcfg, circ), zk.PlonkState(c2, rcfg, circ)
        try:
            rc, ref = orc.prove_plonk(ost, wires, pubs)
            assert rc == 0 and (zk.prove_plonk(st, wires, pubs, ctx=c2) == ref).all()
        finally:
            st.free()
    finally:
        c2.close()
