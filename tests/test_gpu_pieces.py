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


@pytest.mark.parametrize("n_sets,log_n", [(2, 4), (6, 7), (10, 11), (16, 13)])
def test_ctl_helpers_with_filters_and_column_sets(ctx, zk, orc, n_sets, log_n):
    """CTL helper columns in upstream's full form -- sum over two (column set, filter) pairs of filter / combine(cols), column sets of
    2..5 columns (zk_helper_ext) -- on the GPU: the columns equal the oracle's, and the whole proof of the table (constraints over those
    columns, the interpreter and -- at 2^13 rows -- nothing else changes) equals the oracle's word for word and verifies."""
    from eth_tx_proof_amd import air
    prog, tr, pub = air.ctl_air(n_sets=n_sets, log_n=log_n, seed=n_sets)
    ch = field_rand(np.random.default_rng(2), 4)
    got = zk.aux_columns(ctx, prog, ch, tr)
    rc, ref = orc.aux_columns(prog, ch, tr)
    assert rc == 0 and (got == ref).all()
    cfg = _cfg()
    proof = zk.prove_tables(ctx, cfg, [(prog, log_n, tr, pub)])
    rc, want = orc.prove(cfg, [(prog, log_n, tr, pub)])
    assert rc == 0 and proof.size == want.size and (proof == want).all()
    assert orc.verify(cfg, [(prog, log_n, None, pub)], proof) == 0


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


def _stark_instance(n_cols, n_aux, n_q, nch, zeta, log_n):
    """the FriInstanceInfo a starky table has: (zeta: trace|aux|quot) (g*zeta: trace|aux) (1: ctl Zs)"""
    import pyspec
    P = 0xFFFFFFFF00000001
    g = pyspec.root_of_unity(log_n)
    o_q = 2 if n_aux else 1
    b0 = [(0, c) for c in range(n_cols)] + [(1, c) for c in range(n_aux)] + [(o_q, c) for c in range(n_q)]
    b1 = [(0, c) for c in range(n_cols)] + [(1, c) for c in range(n_aux)]
    out = [(zeta, b0), ((int(zeta[0]) * g % P, int(zeta[1]) * g % P), b1)]
    if n_aux:
        out.append(((1, 0), [(1, n_aux - nch + j) for j in range(nch)]))
    return out


@pytest.mark.parametrize("n_helpers,log_n", [(2, 8), (0, 7)])
def test_fri_instance_equals_stark_entry(ctx, zk, orc, n_helpers, log_n):
    """zk_fri_prove_instance on the instance a STARK table has == zk_fri_prove_openings == the oracle's generic restatement"""
    from eth_tx_proof_amd import air
    cfg = _cfg()
    rng = np.random.default_rng(40 + log_n)
    prog = air.cpu_like_air(9, n_helpers)
    tr, pub = orc.synth_trace(9, log_n, 5)
    ch4, al = field_rand(rng, 4), field_rand(rng, 2)
    tb, ob = ctx.batch_from_values(tr), orc.Batch(tr)
    ab = oab = None
    if n_helpers:
        rc, aux = orc.aux_columns(prog, ch4, tr)
        ab, oab = ctx.batch_from_values(aux), orc.Batch(aux)
    q = orc.quotient(cfg, prog, ob, oab, pub, ch4, al)
    qb, oqb = ctx.batch_from_coeffs(q), orc.Batch(q, from_coeffs=True)
    zeta = field_rand(rng, 2)
    inst = _stark_instance(9, prog.n_aux_cols if n_helpers else 0, q.shape[0], 2, zeta, log_n)
    seed = field_rand(rng, 5)
    outs = []
    for which in range(3):
        if which < 2:
            c = ctx.challenger(); c.observe(seed)
            gor = [b for b in (tb, ab, qb) if b is not None]
            outs.append(zk.fri_prove_instance(ctx, cfg, gor, inst, c) if which == 0 else zk.fri_prove_openings(ctx, cfg, tb, ab, qb, c, zeta))
        else:
            c = orc.Challenger(); c.observe(seed)
            rc, ref = orc.fri_prove_instance(cfg, [b for b in (ob, oab, oqb) if b is not None], inst, c)
            assert rc == 0
            outs.append(ref)
    assert outs[0].size == outs[1].size == outs[2].size
    assert (outs[0] == outs[1]).all() and (outs[0] == outs[2]).all()


@pytest.mark.parametrize("log_n,widths", [(6, (5, 9, 4, 8)), (9, (20, 135, 20, 16)), (12, (4, 135, 20, 16))])
def test_fri_instance_recursion_shape(ctx, zk, orc, log_n, widths):
    """the shape of a plonky2 recursion proof's opening: four oracles (constants+sigmas, wires, Zs+partial
    products, quotient chunks) committed at rate_bits 3, everything opened at zeta, the Z polynomials also at g*zeta;
    standard_recursion_config FRI (28 queries, arity 16, PoW 16).  Bit-exact against the oracle, same challenger after."""
    import eth_tx_proof_amd as E
    import pyspec
    P = 0xFFFFFFFF00000001
    cfg = E.standard_recursion_config()
    rng = np.random.default_rng(log_n)
    gb, ob = [], []
    for k, w in enumerate(widths):
        v = field_rand(rng, (w, 1 << log_n))
        if k == 3:
            gb.append(ctx.batch_from_coeffs(v, rate_bits=3, cap_height=4)); ob.append(orc.Batch(v, rate_bits=3, cap_height=4, from_coeffs=True))
        else:
            gb.append(ctx.batch_from_values(v, rate_bits=3, cap_height=4)); ob.append(orc.Batch(v, rate_bits=3, cap_height=4))
        assert (gb[-1].cap == ob[-1].cap).all()
    zeta = field_rand(rng, 2)
    g = pyspec.root_of_unity(log_n)
    inst = [(zeta, [(o, c) for o, w in enumerate(widths) for c in range(w)]),
            ((int(zeta[0]) * g % P, int(zeta[1]) * g % P), [(2, c) for c in range(2)])]
    seed = field_rand(rng, 7)
    gch, och = ctx.challenger(), orc.Challenger()
    gch.observe(seed); och.observe(seed)
    got = zk.fri_prove_instance(ctx, cfg, gb, inst, gch)
    rc, ref = orc.fri_prove_instance(cfg, ob, inst, och)
    assert rc == 0 and got.size == ref.size and (got == ref).all()
    assert (gch.get(4) == och.get(4)).all()
    for b in gb:
        b.free()


def test_fri_instance_error_paths(ctx, zk):
    import eth_tx_proof_amd as E
    cfg = E.standard_fast_config()
    rng = np.random.default_rng(1)
    b = ctx.batch_from_values(field_rand(rng, (3, 64)))
    c = ctx.challenger()
    with pytest.raises(zk.ZkError) as e:      # polynomial index out of range
        zk.fri_prove_instance(ctx, cfg, [b], [((5, 6), [(0, 3)])], c)
    assert e.value.code == -1
    with pytest.raises(zk.ZkError) as e:      # empty batch
        zk.fri_prove_instance(ctx, cfg, [b], [((5, 6), [])], c)
    assert e.value.code == -1
    with pytest.raises(zk.ZkError) as e:      # base-field point on the LDE coset g*H
        zk.fri_prove_instance(ctx, cfg, [b], [((E.COSET_SHIFT, 0), [(0, 0)])], c)
    assert e.value.code == -8
    # a valid call still works afterwards
    assert zk.fri_prove_instance(ctx, cfg, [b], [((5, 6), [(0, 0), (0, 2)])], c).size > 0


@pytest.mark.parametrize("n_cols,n_helpers,log_n", [(6, 2, 7), (93, 19, 9), (7, 0, 8), (21, 5, 10), (430, 19, 6)])
def test_quotient_jit_equals_interpreter_and_oracle(zk, orc, monkeypatch, n_cols, n_helpers, log_n):
    """the constraint program compiled with hiprtc (ZK_QUOTIENT_JIT=2: always) and the bytecode interpreter (=0) produce the
    oracle's quotient polynomials bit for bit.  (430 columns = 2 191 instructions: the generator cuts programs of that length into
    a sequence of call-free kernels that pass accumulators and live registers on through a state buffer.)"""
    from eth_tx_proof_amd import air
    cfg = _cfg()
    rng = np.random.default_rng(log_n)
    prog = air.cpu_like_air(n_cols, n_helpers)
    tr, pub = orc.synth_trace(n_cols, log_n, 5)
    ch, al = field_rand(rng, 4), field_rand(rng, 2)
    ob = orc.Batch(tr)
    oab = aux = None
    if n_helpers:
        rc, aux = orc.aux_columns(prog, ch, tr)
        assert rc == 0
        oab = orc.Batch(aux)
    ref = orc.quotient(cfg, prog, ob, oab, pub, ch, al)
    # "2+forced redo": every slice flagged -> all redone by the interpreter fix-up; "0+prepared": kernels compiled ahead of
    # time by zk_prover_state_prepare (min size lowered so that these small tables qualify); "-": never (the default)
    # "2+forced redo": every wave takes the exact-redo branch of the compiled kernel
    for mode in ("2", "0", "2+forced redo", "0+prepared", "-", "aot"):
        monkeypatch.setenv("ZK_QUOTIENT_JIT", "-1" if mode in ("-", "aot") else mode[0])
        monkeypatch.setenv("ZK_QUOTIENT_JIT_MIN_LOG", "1")
        monkeypatch.setenv("ZK_FORCE_EXACT_REDO", "1" if "forced" in mode else "0")
        monkeypatch.setenv("ZK_QUOTIENT_AOT", "1" if mode == "aot" else "0")     # "aot": kernels compiled with the library (93 x 19 shape)
        c2 = zk.Context(0)
        try:
            if "prepared" in mode:
                zk.prover_state_prepare(c2, cfg, [(prog, log_n, tr, pub)])
            tb = c2.batch_from_values(tr)
            ab = c2.batch_from_values(aux) if n_helpers else None
            got = zk.quotient(c2, cfg, prog, tb, ab, pub, ch, al)
            assert (got == ref).all(), mode
        finally:
            c2.close()


def test_prover_state_persists_compiled_kernels_on_disk(tmp_path):
    """The reference keeps its prover state in ./prover_state_<digest> and loads it at the next start
    (common/src/prover_state/persistence.rs:40-88).  Here: a first process compiles the constraint kernel of an AIR that was not
    built with the library (hiprtc) and writes prover_state_<digest>.gfx950.co; a second process loads it without compiling and
    produces the same proof; a truncated file is ignored and rebuilt."""
    import json, os, subprocess, sys, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    worker = textwrap.dedent("""
        import json, sys, hashlib
        sys.path.insert(0, %r)
        import numpy as np
        import eth_tx_proof_amd as E
        from eth_tx_proof_amd import air, lib as zl
        zl.prover_state_cache_dir(sys.argv[1])
        c = zl.Context(0)
        cfg = E.standard_fast_config()
        prog = air.cpu_like_air(11, 3)               # not one of the shapes compiled with the library
        d, pub = zl.synth_trace_dev(c, 11, 17, 5)
        zl.prover_state_prepare(c, cfg, [(prog, 17, d, pub)])
        p = zl.prove_tables(c, cfg, [(prog, 17, d, pub)])
        comp, load = zl.prover_state_stats()
        print(json.dumps({"compiled": comp, "loaded": load, "sha": hashlib.sha256(p.tobytes()).hexdigest()}))
        c.close()
    """) % root
    script = tmp_path / "w.py"
    script.write_text(worker)
    cache = tmp_path / "state"
    cache.mkdir()

    def run():
        out = subprocess.run([sys.executable, str(script), str(cache)], capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        return json.loads(out.stdout.strip().splitlines()[-1])
    first = run()
    files = [f for f in os.listdir(cache) if f.startswith("prover_state_") and f.endswith(".gfx950.co")]
    assert first["compiled"] == 1 and first["loaded"] == 0 and len(files) == 1
    second = run()
    assert second["compiled"] == 0 and second["loaded"] == 1 and second["sha"] == first["sha"]
    path = os.path.join(cache, files[0])
    blob = open(path, "rb").read()
    open(path, "wb").write(blob[: len(blob) // 2])          # damaged file: ignored, rebuilt, rewritten
    third = run()
    assert third["compiled"] == 1 and third["loaded"] == 0 and third["sha"] == first["sha"]
    assert open(path, "rb").read() == blob


@pytest.mark.parametrize("log_n,widths,recursion", [(9, (20, 135, 20, 16), True), (12, (7, 5, 4), False)])
def test_fri_piece_by_piece_equals_the_whole(ctx, zk, orc, log_n, widths, recursion):
    """SURVEY.md 8b's finer-grained FRI exports: zk_fri_combine == the oracle's combination for the same alpha, and the commit
    phase driven layer by layer (zk_fri_fold_commit_layer), its final polynomial and every query round (zk_fri_query) reproduce,
    word for word, what the one-call prover zk_fri_prove_instance (== the oracle, tested above) puts into the FriProof."""
    import eth_tx_proof_amd as E
    import pyspec
    P = 0xFFFFFFFF00000001
    cfg = E.standard_recursion_config() if recursion else E.standard_fast_config()
    rng = np.random.default_rng(100 + log_n)
    gb, ob = [], []
    for w in widths:
        v = field_rand(rng, (w, 1 << log_n))
        gb.append(ctx.batch_from_values(v, rate_bits=cfg.rate_bits, cap_height=cfg.cap_height))
        ob.append(orc.Batch(v, rate_bits=cfg.rate_bits, cap_height=cfg.cap_height))
    zeta = field_rand(rng, 2)
    g = pyspec.root_of_unity(log_n)
    inst = [(zeta, [(o, c) for o, w in enumerate(widths) for c in range(w)]),
            ((int(zeta[0]) * g % P, int(zeta[1]) * g % P), [(len(widths) - 1, c) for c in range(2)])]
    seed = field_rand(rng, 7)
    whole_ch = ctx.challenger(); whole_ch.observe(seed)
    whole = zk.fri_prove_instance(ctx, cfg, gb, inst, whole_ch)
    # ---- the same proof in pieces
    ch = ctx.challenger(); ch.observe(seed)
    alpha = ch.get(2)
    vals = zk.fri_combine(ctx, cfg, gb, inst, alpha)
    assert (vals == orc.fri_combine(cfg, ob, inst, alpha)).all()
    fc = zk.FriCommit(ctx, cfg, log_n, vals)
    nl, done = fc.layers()
    capw = 4 << cfg.cap_height
    pos = 0
    assert whole[pos] == nl; pos += 1
    for l in range(nl):
        cap, beta = fc.fold_commit_layer(ch)
        assert (cap.ravel() == whole[pos:pos + capw]).all(), l
        pos += capw
    with pytest.raises(zk.ZkError):
        fc.fold_commit_layer(ch)                       # no layer left
    fin = fc.finish(ch)
    assert whole[pos] == fin.size // 2; pos += 1
    assert (fin == whole[pos:pos + fin.size]).all(); pos += fin.size
    witness = whole[pos]; pos += 1
    state = ch.compact()                              # nothing pending after the final polynomial's last full block? compact is exact either way
    # replay the transcript tail on the oracle's challenger to get the query indices (same transcript as the whole proof)
    och = orc.Challenger(); och.observe(seed); och.get(2)
    p2 = 1
    for l in range(nl):
        och.observe(whole[p2:p2 + capw]); och.get(2); p2 += capw
    och.observe(fin); och.observe(np.array([witness], dtype=np.uint64)); och.get(1)
    L = 1 << (log_n + cfg.rate_bits)
    idx = [int(x) % L for x in och.get(cfg.num_query_rounds)]
    per = (whole.size - pos) // cfg.num_query_rounds
    for q, x in enumerate(idx[:10]):
        got = fc.query(gb, x)
        assert got.size == per and (got == whole[pos + q * per: pos + (q + 1) * per]).all(), q
    with pytest.raises(zk.ZkError):
        fc.query(gb, L)                                # outside the domain
    fc.free()
    for b in gb:
        b.free()
