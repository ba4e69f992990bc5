"""Pure-Python restatement of the upstream primitives (TEST INFRASTRUCTURE ONLY).

This file is part of the oracle: only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import it.  The product never does.

Source of truth: SURVEY.md Appendix A, which restates the un-vendored crates
pinned by /root/reference/Cargo.lock:3441-3443 (plonky2 0.2.2),
:3466-3468 (plonky2_field 0.2.2) and :4529-4531 (starky 0.4.0), reached from
the reference at ops/src/lib.rs:52 (generate_txn_proof).

Parity status
  * externally anchored: Goldilocks constants, the 360 Poseidon round
    constants (sha256 below), the four upstream Poseidon `test_vectors` KATs.
  * everything above the permutation (sponge mode, Merkle layout, challenger,
    FRI) is a recalled spec: "parity unpinned" (SURVEY.md section 8c).

Pure-Python loops: use for small cases only.
"""
import hashlib
import struct

P = 0xFFFFFFFF00000001
EPS = 0xFFFFFFFF
MULTIPLICATIVE_GROUP_GENERATOR = 14293326489335486720  # = coset shift  (A1)
POWER_OF_TWO_GENERATOR = 7277203076849721926           # order 2^32     (A1)
TWO_ADICITY = 32
W_EXT = 7                                              # ext2: x^2 = 7   (A1)

# ---------------------------------------------------------------- field (A1)

def add(a, b): return (a + b) % P
def sub(a, b): return (a - b) % P
def mul(a, b): return (a * b) % P
def inv(a): return pow(a, P - 2, P)


def root_of_unity(k):
    """primitive_root_of_unity(k) = POWER_OF_TWO_GENERATOR^(2^(32-k))  (A1)"""
    assert 0 <= k <= TWO_ADICITY
    return pow(POWER_OF_TWO_GENERATOR, 1 << (TWO_ADICITY - k), P)


def ext_mul(a, b):
    """(a0+a1 x)(b0+b1 x) with x^2 = 7  (A1)"""
    return ((a[0] * b[0] + W_EXT * a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def ext_add(a, b): return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)
def ext_sub(a, b): return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)


def ext_inv(a):
    # (a0 + a1 x)^-1 = (a0 - a1 x) / (a0^2 - 7 a1^2)
    d = inv((a[0] * a[0] - W_EXT * a[1] * a[1]) % P)
    return (a[0] * d % P, (P - a[1]) * d % P)


def ext_pow(a, e):
    r = (1, 0)
    while e:
        if e & 1:
            r = ext_mul(r, a)
        a = ext_mul(a, a)
        e >>= 1
    return r

# ------------------------------------------------ ChaCha8Rng(seed_from_u64(0)) (A2)


def _pcg32_seed_bytes(state):
    """rand_core::SeedableRng::seed_from_u64: 8 PCG32 outputs -> 32-byte key."""
    MUL, INC = 6364136223846793005, 11634580027462260723
    out = b""
    for _ in range(8):
        state = (state * MUL + INC) & 0xFFFFFFFFFFFFFFFF
        xs = (((state >> 18) ^ state) >> 27) & 0xFFFFFFFF
        rot = state >> 59
        x = ((xs >> rot) | (xs << ((32 - rot) & 31))) & 0xFFFFFFFF
        out += struct.pack("<I", x)
    return out


def _chacha_block(key_words, counter, rounds):
    def rotl(v, n): return ((v << n) | (v >> (32 - n))) & 0xFFFFFFFF
    st = [0x61707865, 0x3320646E, 0x79622D32, 0x6B206574] + list(key_words) + [
        counter & 0xFFFFFFFF, (counter >> 32) & 0xFFFFFFFF, 0, 0]
    x = list(st)

    def qr(a, b, c, d):
        x[a] = (x[a] + x[b]) & 0xFFFFFFFF; x[d] = rotl(x[d] ^ x[a], 16)
        x[c] = (x[c] + x[d]) & 0xFFFFFFFF; x[b] = rotl(x[b] ^ x[c], 12)
        x[a] = (x[a] + x[b]) & 0xFFFFFFFF; x[d] = rotl(x[d] ^ x[a], 8)
        x[c] = (x[c] + x[d]) & 0xFFFFFFFF; x[b] = rotl(x[b] ^ x[c], 7)
    for _ in range(rounds // 2):
        qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15)
        qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14)
    return [(x[i] + st[i]) & 0xFFFFFFFF for i in range(16)]


def chacha8_u64_stream(seed_u64):
    key = struct.unpack("<8I", _pcg32_seed_bytes(seed_u64))
    counter = 0
    while True:
        blk = _chacha_block(key, counter, 8)
        counter += 1
        for i in range(0, 16, 2):
            yield blk[i] | (blk[i + 1] << 32)


def poseidon_round_constants():
    """360 x gen_range(0..p) from ChaCha8Rng::seed_from_u64(0)  (A2)."""
    out = []
    s = chacha8_u64_stream(0)
    while len(out) < 360:
        v = next(s)
        m = v * P
        if (m & 0xFFFFFFFFFFFFFFFF) <= P - 1:
            out.append(m >> 64)
    return out


RC_SHA256 = "d2fcbb5be293c50ab4b1ddcd9c81005b12d689816a54c91a054f97f6588a20a8"

# ----------------------------------------------------------- Poseidon (A2)
WIDTH, RATE, CAP = 12, 8, 4
N_FULL_HALF, N_PARTIAL = 4, 22
MDS_CIRC = [17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20]
MDS_DIAG = [8] + [0] * 11
_RC = None


def rc():
    global _RC
    if _RC is None:
        _RC = poseidon_round_constants()
        h = hashlib.sha256(b"".join(struct.pack("<Q", c) for c in _RC)).hexdigest()
        assert h == RC_SHA256, h
    return _RC


def mds_matrix():
    """M[r][c] such that out[r] = sum_c M[r][c] s[c]  (A2)."""
    return [[(MDS_CIRC[(c - r) % 12] + (MDS_DIAG[r] if r == c else 0)) for c in range(12)]
            for r in range(12)]


def _mds(s):
    return [(sum(s[(i + r) % 12] * MDS_CIRC[i] for i in range(12)) + s[r] * MDS_DIAG[r]) % P
            for r in range(12)]


def poseidon(state):
    """Naive permutation: 4 full + 22 partial + 4 full, x^7  (A2)."""
    RC = rc()
    s = [x % P for x in state]
    r = 0
    for phase, n in ((0, N_FULL_HALF), (1, N_PARTIAL), (0, N_FULL_HALF)):
        for _ in range(n):
            s = [(s[i] + RC[12 * r + i]) % P for i in range(12)]
            if phase == 0:
                s = [pow(x, 7, P) for x in s]
            else:
                s[0] = pow(s[0], 7, P)
            s = _mds(s)
            r += 1
    return s


KATS = [
    ([0] * 12,
     "3c18a9786cb0b359 c4055e3364a246c3 7953db0ab48808f4 c71603f33a1144ca d7709673896996dc 46a84e87642f44ed "
     "d032648251ee0b3c 1c687363b207df62 df8565563e8045fe 40f5b37ff4254dae d070f637b431067c 1792b1c4342109d7"),
    (list(range(12)),
     "d64e1e3efc5b8e9e 53666633020aaa47 d40285597c6a8825 613a4f81e81231d2 414754bfebd051f0 cb1f8980294a023f "
     "6eb2a9e4d54a9d0f 1902bc3af467e056 f045d5eafdc6021f e4150f77caaa3be5 c9bfd01d39b50cce 5c0a27fcb0e1459b"),
    ([P - 1] * 12,
     "be0085cfc57a8357 d95af71847d05c09 cf55a13d33c1c953 95803a74f4530e82 fcd99eb30a135df1 e095905e913a3029 "
     "de0392461b42919b 7d3260e24e81d031 10d3d0465d9deaa0 a87571083dfc2a47 e18263681e9958f8 e28e96f1ae5e60d3"),
    ([int(x, 16) for x in
      "8ccbbbea4fe5d2b7 c2af59ee9ec49970 90f7e1a9e658446a dcc0630a3ab8b1b8 7ff8256bca20588c 5d99a7ca0c44ecfb "
      "48452b17a70fbee3 eb09d654690b6c88 4a55d3a39c676a88 c0407a38d2285139 a234bac9356386d1 e1633f2bad98a52f".split()],
     "a89280105650c4ec ab542d53860d12ed 5704148e9ccab94f d3a826d4b62da9f5 8a7a6ca87892574f c7017e1cad1a674e "
     "1f06668922318e34 a3b203bc8102676f fcc781b0ce382bf2 934c69ff3ed14ba5 504688a5996e8f13 401f3f2ed524a2ba"),
]

# ------------------------------------------------------- sponge / compression (A4)


def hash_no_pad(x):
    st = [0] * 12
    for off in range(0, len(x), RATE):
        chunk = x[off:off + RATE]
        st[:len(chunk)] = [c % P for c in chunk]
        st = poseidon(st)
    return st[:4]


def hash_or_noop(x):
    if len(x) <= 4:
        return [c % P for c in x] + [0] * (4 - len(x))
    return hash_no_pad(x)


def two_to_one(l, r):
    return poseidon(list(l) + list(r) + [0] * 4)[:4]

# ------------------------------------------------------------- FFT / LDE (A3)


def bitrev(i, bits):
    r = 0
    for _ in range(bits):
        r = (r << 1) | (i & 1)
        i >>= 1
    return r


def fft(c):
    n = len(c)
    k = n.bit_length() - 1
    assert 1 << k == n
    a = [c[bitrev(i, k)] % P for i in range(n)]
    m = 1
    while m < n:
        w_m = root_of_unity((2 * m).bit_length() - 1)
        for s in range(0, n, 2 * m):
            w = 1
            for j in range(m):
                u, v = a[s + j], a[s + j + m] * w % P
                a[s + j], a[s + j + m] = (u + v) % P, (u - v) % P
                w = w * w_m % P
        m *= 2
    return a


def ifft(v):
    n = len(v)
    ninv = inv(n)
    f = fft(v)
    return [f[(n - i) % n] * ninv % P for i in range(n)]


def coset_fft(c, shift):
    s, out = 1, []
    for x in c:
        out.append(x * s % P)
        s = s * shift % P
    return fft(out)


def coset_ifft(v, shift):
    c = ifft(v)
    sinv, s, out = inv(shift), 1, []
    for x in c:
        out.append(x * s % P)
        s = s * sinv % P
    return out


def lde_values(coeffs, rate_bits):
    """PolynomialCoeffs::lde(rate_bits).coset_fft(g)  (A3)"""
    n = len(coeffs)
    return coset_fft(list(coeffs) + [0] * (n * ((1 << rate_bits) - 1)), MULTIPLICATIVE_GROUP_GENERATOR)

# -------------------------------------------------------------- Merkle (A5)


def merkle_tree(leaves, cap_height):
    """Returns (cap, digests) with plonky2's recursive digest layout (A5)."""
    L = len(leaves)
    lg = L.bit_length() - 1
    assert 1 << lg == L and cap_height <= lg
    hashes = [hash_or_noop(l) for l in leaves]
    if cap_height == lg:
        return hashes, []
    ncap = 1 << cap_height
    sub = L // ncap

    def build(hs):
        # returns (root, buffer) ; buffer = left_buf | left_digest | right_digest | right_buf
        if len(hs) == 1:
            return hs[0], []
        h = len(hs) // 2
        lroot, lbuf = build(hs[:h])
        rroot, rbuf = build(hs[h:])
        return two_to_one(lroot, rroot), lbuf + [lroot, rroot] + rbuf
    cap, digests = [], []
    for i in range(ncap):
        root, buf = build(hashes[i * sub:(i + 1) * sub])
        cap.append(root)
        digests += buf
    return cap, digests


def merkle_prove(leaves, cap_height, idx):
    """Sibling path from the leaf level up to (not including) the cap level (A5)."""
    L = len(leaves)
    lg = L.bit_length() - 1
    level = [hash_or_noop(l) for l in leaves]
    sibs = []
    for _ in range(lg - cap_height):
        sibs.append(level[idx ^ 1])
        level = [two_to_one(level[2 * i], level[2 * i + 1]) for i in range(len(level) // 2)]
        idx >>= 1
    return sibs


def merkle_verify(leaf, idx, sibs, cap):
    cur = hash_or_noop(leaf)
    for s in sibs:
        cur = two_to_one(s, cur) if idx & 1 else two_to_one(cur, s)
        idx >>= 1
    return cur == list(cap[idx])

# ---------------------------------------------------------- Challenger (A6)


class Challenger:
    def __init__(self):
        self.state = [0] * 12
        self.inp = []
        self.out = []

    def observe(self, x):
        self.out = []
        self.inp.append(x % P)
        if len(self.inp) == RATE:
            self._duplex()

    def observe_many(self, xs):
        for x in xs:
            self.observe(x)

    def observe_cap(self, cap):
        for h in cap:
            self.observe_many(h)

    def observe_ext(self, e):
        self.observe(e[0]); self.observe(e[1])

    def _duplex(self):
        self.state[:len(self.inp)] = self.inp
        self.inp = []
        self.state = poseidon(self.state)
        self.out = list(self.state[:RATE])

    def get(self):
        if self.inp or not self.out:
            self._duplex()
        return self.out.pop()

    def get_n(self, n):
        return [self.get() for _ in range(n)]

    def get_ext(self):
        a = self.get(); b = self.get()
        return (a, b)

    def compact(self):
        if self.inp:
            self._duplex()
        self.out = []
        return list(self.state)


def fri_reduction_arity_bits(degree_bits, rate_bits=1, cap_height=4, arity_bits=4, final_poly_bits=5):
    """ConstantArityBits(4,5) schedule (A8)."""
    out, d = [], degree_bits
    while d > final_poly_bits and d + rate_bits - arity_bits >= cap_height:
        out.append(arity_bits)
        d -= arity_bits
    return out


# ---------------------------------------------------------------- a whole unit proof, in plain Python (tiny sizes only)
# Second, independent restatement of what oracle/stark.c does (starky prove_with_commitment + plonky2 fri_proof):
# same inputs, same proof-word layout (include/zkgl.h), written from the appendix with Python integers and lists, sharing no code
# with the C oracle -- tests/test_oracle_cpu.py compares the two word for word at 2^4 .. 2^6 rows.  It narrows the single-author
# risk of stark.c; parity with upstream stays unpinned (no vector exists).
def poly_eval_ext(coeffs, z):
    acc = (0, 0)
    for c in reversed(coeffs):
        acc = ext_add(ext_mul(acc, z), (c % P, 0))
    return acc


def commit_values(cols, rate_bits, cap_height):
    """PolynomialBatch::from_values: (coeffs per column, leaves (rows, bit-reversed order), cap, digests-by-level)"""
    coeffs = [ifft(c) for c in cols]
    return commit_coeffs(coeffs, rate_bits, cap_height)


def commit_coeffs(coeffs, rate_bits, cap_height):
    n = len(coeffs[0])
    lg = (n << rate_bits).bit_length() - 1
    ldes = [coset_fft(c + [0] * ((n << rate_bits) - n), MULTIPLICATIVE_GROUP_GENERATOR) for c in coeffs]
    leaves = [[ldes[c][bitrev(r, lg)] for c in range(len(coeffs))] for r in range(n << rate_bits)]
    cap, levels = _tree_levels([hash_or_noop(l) for l in leaves], cap_height)
    return {"coeffs": coeffs, "leaves": leaves, "cap": cap, "levels": levels}


def _tree_levels(digests, cap_height):
    levels = [digests]
    while len(levels[-1]) > (1 << cap_height):
        cur = levels[-1]
        levels.append([two_to_one(cur[2 * i], cur[2 * i + 1]) for i in range(len(cur) // 2)])
    return levels[-1], levels


def _path(levels, idx):
    out = []
    for k in range(len(levels) - 1):
        out += levels[k][(idx >> k) ^ 1]
    return out


def _aux_columns(helpers, nch, chal, trace, n, helpers_ext=None):
    """CTL helper columns and Z; helpers_ext[m] = (filter0, filter1, extra0, extra1) refines helper m: filter / combine(column set)"""
    H = len(helpers)
    cols = [[0] * n for _ in range(nch * (H + 1))]
    for j in range(nch):
        beta, gamma = chal[2 * j], chal[2 * j + 1]
        for m, (a0, b0, a1, b1) in enumerate(helpers):
            f0, f1, e0, e1 = helpers_ext[m] if helpers_ext else (None, None, [], [])
            s0, s1 = [a0, b0] + list(e0), [a1, b1] + list(e1)
            for r in range(n):
                d0 = (sum(trace[c][r] * pow(beta, i, P) for i, c in enumerate(s0)) + gamma) % P
                d1 = (sum(trace[c][r] * pow(beta, i, P) for i, c in enumerate(s1)) + gamma) % P
                v0 = trace[f0][r] if f0 is not None else 1
                v1 = trace[f1][r] if f1 is not None else 1
                cols[j * H + m][r] = (v0 * inv(d0) + v1 * inv(d1)) % P
        run = 0
        for r in range(n - 1, -1, -1):
            run = (run + sum(cols[j * H + m][r] for m in range(H))) % P
            cols[nch * H + j][r] = run
    return cols


def _eval_program(instrs, consts, n_regs, local, nxt, pub, chal, alphas, z_last, l_first, l_last):
    regs = [0] * max(1, n_regs)
    acc = [0] * len(alphas)

    def val(o):
        k, i = o >> 24, o & 0xFFFFFF
        if k == 0:
            return regs[i]
        if k == 1:
            return local[i]
        if k == 2:
            return nxt[i]
        return (consts[i] if k == 3 else pub[i] if k == 4 else chal[i]) % P
    for op, dst, a, b in instrs:
        x = val(a)
        if op <= 2:
            y = val(b)
            regs[dst & 0xFFFFFF] = (x + y) % P if op == 0 else (x - y) % P if op == 1 else (x * y) % P
        else:
            c = x if op == 3 else x * (z_last if op == 4 else l_first if op == 5 else l_last) % P
            acc = [(a_ * al + c) % P for a_, al in zip(acc, alphas)]
    return acc


def prove_tables(cfg, tables):
    """cfg: dict(num_challenges, rate_bits, cap_height, proof_of_work_bits, fri_arity_bits, fri_final_poly_bits, num_query_rounds);
    tables: list of dict(log_n, trace (cols x n ints), pub, instrs [(op,dst,a,b)], constants, n_regs, helpers [(a0,b0,a1,b1)]).
    Returns the proof words (list of ints) in the layout of zk_prove_tables."""
    nch, rb, ch_h = cfg["num_challenges"], cfg["rate_bits"], cfg["cap_height"]
    assert rb == 1
    g = MULTIPLICATIVE_GROUP_GENERATOR
    out = [0x31465250474C4B5A, len(tables)]
    ch = Challenger()
    tb = []
    for t in tables:
        b = commit_values([list(c) for c in t["trace"]], rb, ch_h)
        tb.append(b)
        for h in b["cap"]:
            ch.observe_many(h)
    for t in tables:
        ch.observe_many([x % P for x in t["pub"]])
    chal = ch.get_n(2 * nch)
    for t, b in zip(tables, tb):
        k, n = t["log_n"], 1 << t["log_n"]
        L, lgL = n << rb, t["log_n"] + rb
        C, H = len(t["trace"]), len(t["helpers"])
        A = nch * (H + 1) if H else 0
        ch.compact()
        out += [k, C, A] + [w for h in b["cap"] for w in h]
        ab = None
        if A:
            ab = commit_values(_aux_columns(t["helpers"], nch, chal, t["trace"], n, t.get("helpers_ext")), rb, ch_h)
            for h in ab["cap"]:
                ch.observe_many(h)
            out += [w for h in ab["cap"] for w in h]
        alphas = ch.get_n(nch)
        # ---- quotient: constraints on the coset g*<w_L>, / Z_H, coset_ifft, chunks of n
        w_L, w_n = root_of_unity(lgL), root_of_unity(k)
        sel_f, sel_l = [0] * n, [0] * n
        sel_f[0], sel_l[n - 1] = 1, 1
        lag_f = coset_fft(ifft(sel_f) + [0] * (L - n), g)
        lag_l = coset_fft(ifft(sel_l) + [0] * (L - n), g)
        gn = pow(g, n, P)
        zh_inv = [inv((gn * pow(root_of_unity(rb), i, P) - 1) % P) for i in range(1 << rb)]
        last = inv(w_n)
        qv = [[0] * L for _ in range(nch)]
        for i in range(L):
            rl, rn = bitrev(i, lgL), bitrev((i + (1 << rb)) % L, lgL)
            local = b["leaves"][rl] + (ab["leaves"][rl] if A else [])
            nxt = b["leaves"][rn] + (ab["leaves"][rn] if A else [])
            x = g * pow(w_L, i, P) % P
            acc = _eval_program(t["instrs"], t["constants"], t["n_regs"], local, nxt, t["pub"], chal, alphas, (x - last) % P, lag_f[i], lag_l[i])
            for j in range(nch):
                qv[j][i] = acc[j] * zh_inv[i % (1 << rb)] % P
        qco = []
        for j in range(nch):
            c = coset_ifft(qv[j], g)
            qco += [c[s * n:(s + 1) * n] for s in range(1 << rb)]
        qb = commit_coeffs(qco, rb, ch_h)
        for h in qb["cap"]:
            ch.observe_many(h)
        out += [w for h in qb["cap"] for w in h]
        zeta = ch.get_ext()
        assert ext_pow(zeta, n) != (1, 0)
        gz = ext_mul(zeta, (w_n, 0))
        Q = len(qco)
        o_local = [poly_eval_ext(c, zeta) for c in b["coeffs"]]
        o_next = [poly_eval_ext(c, gz) for c in b["coeffs"]]
        o_aux = [poly_eval_ext(c, zeta) for c in ab["coeffs"]] if A else []
        o_auxn = [poly_eval_ext(c, gz) for c in ab["coeffs"]] if A else []
        o_q = [poly_eval_ext(c, zeta) for c in qb["coeffs"]]
        zs = [sum(ab["coeffs"][A - nch + j]) % P for j in range(nch)] if A else []
        flat = lambda es: [w for e in es for w in e]
        out += flat(o_local) + flat(o_next) + flat(o_aux) + flat(o_auxn) + zs + flat(o_q)
        ch.observe_many(flat(o_local)); ch.observe_many(flat(o_aux)); ch.observe_many(flat(o_q))
        ch.observe_many(flat(o_next)); ch.observe_many(flat(o_auxn))
        for z in zs:
            ch.observe_many([z, 0])
        # ---- FRI: batches (zeta: trace|aux|quot) (g zeta: trace|aux) (1: Zs)
        oracles = [b] + ([ab] if A else []) + [qb]
        batches = [(zeta, b["coeffs"] + (ab["coeffs"] if A else []) + qb["coeffs"]), (gz, b["coeffs"] + (ab["coeffs"] if A else []))]
        if A:
            batches.append(((1, 0), ab["coeffs"][A - nch:]))
        out += _fri_prove(cfg, k, oracles, batches, ch)
    return out


def _fri_prove(cfg, k, oracles, batches, ch):
    rb, ch_h = cfg["rate_bits"], cfg["cap_height"]
    n, L, lgL = 1 << k, (1 << k) << rb, k + rb
    g = MULTIPLICATIVE_GROUP_GENERATOR
    alpha = ch.get_ext()
    final = [(0, 0)] * n
    for bi, (z, polys) in enumerate(batches):
        comp = []
        for i in range(n):
            acc = (0, 0)
            for p in reversed(polys):
                acc = ext_add(ext_mul(acc, alpha), (p[i] % P, 0))
            comp.append(acc)
        q, acc = [(0, 0)] * n, (0, 0)              # divide_by_linear: Horner from the top, drop the remainder, push a zero
        for i in range(n - 1, -1, -1):
            acc = ext_add(ext_mul(acc, z), comp[i])
            if i > 0:
                q[i - 1] = acc
        sh = ext_pow(alpha, len(polys))
        final = [ext_add(ext_mul(f, sh) if bi else (0, 0), qq) for f, qq in zip(final, q)]
    coeffs = final + [(0, 0)] * (L - n)

    def ext_coset_fft(cs, shift):
        re, im = coset_fft([c[0] for c in cs], shift), coset_fft([c[1] for c in cs], shift)
        return list(zip(re, im))
    values = ext_coset_fft(coeffs, g)
    ar = fri_reduction_arity_bits(k, rb, ch_h, cfg["fri_arity_bits"], cfg["fri_final_poly_bits"])
    out = [len(ar)]
    cur_lg, shift = lgL, g
    layers = []
    for a in ar:
        arity = 1 << a
        cur = 1 << cur_lg
        br = [values[bitrev(i, cur_lg)] for i in range(cur)]
        leaves = [[w for e in br[i * arity:(i + 1) * arity] for w in e] for i in range(cur // arity)]
        cap, levels = _tree_levels([hash_or_noop(l) for l in leaves], ch_h)
        for h in cap:
            ch.observe_many(h)
        out += [w for h in cap for w in h]
        layers.append((leaves, levels, a))
        beta = ch.get_ext()
        nxt = []
        for i in range(cur // arity):
            acc = (0, 0)
            for t in range(arity - 1, -1, -1):
                acc = ext_add(ext_mul(acc, beta), coeffs[arity * i + t])
            nxt.append(acc)
        coeffs = nxt
        shift = pow(shift, arity, P)
        cur_lg -= a
        values = ext_coset_fft(coeffs, shift)
    flen = (1 << cur_lg) >> rb
    out += [flen] + [w for e in coeffs[:flen] for w in e]
    ch.observe_many([w for e in coeffs[:flen] for w in e])
    # proof of work: smallest witness
    st = list(ch.state)
    for i, v in enumerate(ch.inp):
        st[i] = v
    pos, wit = len(ch.inp), 0
    while True:
        s2 = list(st); s2[pos] = wit
        r = poseidon(s2)[7]
        if (64 - r.bit_length()) >= cfg["proof_of_work_bits"]:
            break
        wit += 1
    ch.observe(wit)
    assert (64 - ch.get().bit_length()) >= cfg["proof_of_work_bits"]
    out.append(wit)
    for x in [c % L for c in ch.get_n(cfg["num_query_rounds"])]:
        for o in oracles:
            out += o["leaves"][x] + _path(o["levels"], x)
        for leaves, levels, a in layers:
            leaf = x >> a
            out += leaves[leaf] + _path(levels, leaf)
            x = leaf
    return out


# ------------------------------------------------------------------------------------------------ Plonk prover (second restatement)
HASH_PAD_MULTIPLE = 8      # RATE, restated from recall (WIDTH = 12 would be the other reading): include/zkgl.h ZK_HASH_PAD_MULTIPLE


def hash_pad(x):
    """Hasher::hash_pad: x | 1 | 0.. | 1, padded to a multiple of HASH_PAD_MULTIPLE, through hash_no_pad (pad10*1)"""
    p = [v % P for v in x] + [1]
    while (len(p) + 1) % HASH_PAD_MULTIPLE:
        p.append(0)
    p.append(1)
    return hash_no_pad(p)


def _gate_terms(instrs, consts, n_regs, local, pih):
    """gate constraint terms of one point: OP 3 (EMIT) writes term (number of EMITs so far), OP 7 (EMIT_SLOT) term `dst`; terms add up"""
    regs = [0] * max(1, n_regs)
    out, seq = {}, 0

    def val(o):
        k, i = o >> 24, o & 0xFFFFFF
        return regs[i] if k == 0 else local[i] if k == 1 else consts[i] % P if k == 3 else pih[i]
    for op, dst, a, b in instrs:
        x = val(a)
        if op <= 2:
            y = val(b)
            regs[dst & 0xFFFFFF] = (x + y) % P if op == 0 else (x - y) % P if op == 1 else (x * y) % P
        elif op == 3:
            out[seq] = (out.get(seq, 0) + x) % P
            seq += 1
        else:
            out[dst] = (out.get(dst, 0) + x) % P
    return out


def prove_plonk(cfg, circuit, wires, public_inputs):
    """plonky2 plonk::prover::prove_with_partition_witness over a circuit given as data, in plain Python, written from the published
    protocol separately from oracle/plonk.c.  cfg as in prove_tables; circuit: dict(degree_bits, num_wires, num_routed, qdf,
    constants (cols x n), sigmas (cols x n), instrs, gate_constants, n_regs, domain_separator); wires: cols x n.
    Returns the proof words in the layout of zk_prove_plonk."""
    nch, rb, ch_h = cfg["num_challenges"], cfg["rate_bits"], cfg["cap_height"]
    k, W, R, qdf = circuit["degree_bits"], circuit["num_wires"], circuit["num_routed"], circuit["qdf"]
    n, lgL = 1 << k, k + rb
    L = n << rb
    g = MULTIPLICATIVE_GROUP_GENERATOR
    consts, sigmas = circuit["constants"], circuit["sigmas"]
    NC = len(consts)
    assert qdf == 1 << rb and R % qdf == 0
    nchunks = R // qdf
    npp = nchunks - 1
    # ---- CircuitBuilder::build (prover side): constants | sigmas commitment and the circuit digest
    cs = commit_values([list(c) for c in consts] + [list(c) for c in sigmas], rb, ch_h)
    digest = hash_no_pad([w for h in cs["cap"] for w in h] + hash_pad(circuit.get("domain_separator", [])) + [k])
    out = [0x314B4C504C474B5A, k, W, R, NC]
    wb = commit_values([[int(x) % P for x in c] for c in wires], rb, ch_h)
    out += [w for h in wb["cap"] for w in h]
    pih = hash_no_pad([int(x) % P for x in public_inputs])
    ch = Challenger()
    ch.observe_many(digest)
    ch.observe_many(pih)
    for h in wb["cap"]:
        ch.observe_many(h)
    betas = ch.get_n(nch)
    gammas = ch.get_n(nch)
    # ---- wires_permutation_partial_products_and_zs
    w_n = root_of_unity(k)
    ks = [pow(g, j, P) for j in range(R)]
    zcols = [[0] * n for _ in range(nch)]
    ppcols = [[[0] * n for _ in range(npp)] for _ in range(nch)]
    for ci in range(nch):
        z, x = 1, 1
        for r in range(n):
            zcols[ci][r] = z
            acc = z
            for c in range(nchunks):
                num = den = 1
                for t in range(qdf):
                    j = c * qdf + t
                    wv = int(wires[j][r]) % P
                    num = num * ((wv + betas[ci] * ks[j] % P * x + gammas[ci]) % P) % P
                    den = den * ((wv + betas[ci] * (int(sigmas[j][r]) % P) + gammas[ci]) % P) % P
                acc = acc * num % P * inv(den) % P
                if c < npp:
                    ppcols[ci][c][r] = acc
            z = acc
            x = x * w_n % P
    zpp = zcols + [col for ci in range(nch) for col in ppcols[ci]]
    NZ = len(zpp)
    zb = commit_values(zpp, rb, ch_h)
    out += [w for h in zb["cap"] for w in h]
    for h in zb["cap"]:
        ch.observe_many(h)
    alphas = ch.get_n(nch)
    # ---- vanishing polynomial on the coset g<w_L>, divided by Z_H
    w_L = root_of_unity(lgL)
    sel = [0] * n
    sel[0] = 1
    l0 = coset_fft(ifft(sel) + [0] * (L - n), g)
    gn = pow(g, n, P)
    zh_inv = [inv((gn * pow(root_of_unity(rb), i, P) - 1) % P) for i in range(qdf)]
    qv = [[0] * L for _ in range(nch)]
    for i in range(L):
        rl, rn = bitrev(i, lgL), bitrev((i + qdf) % L, lgL)
        wr, zr, zn, cr = wb["leaves"][rl], zb["leaves"][rl], zb["leaves"][rn], cs["leaves"][rl]
        x = g * pow(w_L, i, P) % P
        terms = [l0[i] * (zr[ci] - 1) % P for ci in range(nch)]
        for ci in range(nch):
            for c in range(nchunks):
                num = den = 1
                for u in range(qdf):
                    j = c * qdf + u
                    num = num * ((wr[j] + betas[ci] * ks[j] % P * x + gammas[ci]) % P) % P
                    den = den * ((wr[j] + betas[ci] * cr[NC + j] + gammas[ci]) % P) % P
                prev = zr[ci] if c == 0 else zr[nch + ci * npp + c - 1]
                nxt = zn[ci] if c == nchunks - 1 else zr[nch + ci * npp + c]
                terms.append((prev * num - nxt * den) % P)
        gt = _gate_terms(circuit["instrs"], circuit["gate_constants"], circuit["n_regs"], wr + cr[:NC], pih)
        terms += [gt.get(t, 0) for t in range(max(gt) + 1 if gt else 0)]
        for j in range(nch):
            acc = 0
            for t in reversed(terms):
                acc = (acc * alphas[j] + t) % P
            qv[j][i] = acc * zh_inv[i % qdf] % P
    qco = []
    for j in range(nch):
        c = coset_ifft(qv[j], g)
        qco += [c[s * n:(s + 1) * n] for s in range(qdf)]
    qb = commit_coeffs(qco, rb, ch_h)
    out += [w for h in qb["cap"] for w in h]
    for h in qb["cap"]:
        ch.observe_many(h)
    zeta = ch.get_ext()
    assert ext_pow(zeta, n) != (1, 0)
    gz = ext_mul(zeta, (w_n, 0))
    ev = lambda b, z: [poly_eval_ext(c, z) for c in b["coeffs"]]
    o_cs, o_w, o_zpp, o_q = ev(cs, zeta), ev(wb, zeta), ev(zb, zeta), ev(qb, zeta)
    o_zn = [poly_eval_ext(c, gz) for c in zb["coeffs"][:nch]]
    flat = lambda es: [w for e in es for w in e]
    # proof order: constants, sigmas, wires, zs, zs_next, partial products, quotient; transcript: the zeta batch, then zs_next
    out += flat(o_cs) + flat(o_w) + flat(o_zpp[:nch]) + flat(o_zn) + flat(o_zpp[nch:]) + flat(o_q)
    ch.observe_many(flat(o_cs) + flat(o_w) + flat(o_zpp[:nch]) + flat(o_zpp[nch:]) + flat(o_q))
    ch.observe_many(flat(o_zn))
    batches = [(zeta, cs["coeffs"] + wb["coeffs"] + zb["coeffs"] + qb["coeffs"]), (gz, zb["coeffs"][:nch])]
    return out + _fri_prove(cfg, k, [cs, wb, zb, qb], batches, ch)
