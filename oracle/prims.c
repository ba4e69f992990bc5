/* ORACLE -- test infrastructure, not product code (see oracle.h for who may call this).
 * Poseidon, FFT, Merkle tree, Challenger, PolynomialBatch: plain-C restatement of the plonky2 0.2.2 /
 * plonky2_field 0.2.2 items named at each function (sources absent from /root/reference; spec = SURVEY.md A1-A6;
 * entry into this code from the reference: /root/reference/ops/src/lib.rs:52).
 */
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <omp.h>
#include "gl.h"
#include "oracle.h"
#include "poseidon_constants.h"

static int g_threads = 0;
void orc_set_threads(int32_t n) { g_threads = n; if (n > 0) omp_set_num_threads(n); }
int32_t orc_get_threads(void) { return g_threads > 0 ? g_threads : omp_get_max_threads(); }

uint64_t orc_gl_mul(uint64_t a, uint64_t b) { return gl_mul(a, b); }
uint64_t orc_gl_inv(uint64_t a) { return gl_inv(a); }
uint64_t orc_gl_root(uint32_t k) { return gl_root(k); }

/* ---------------------------------------------------------------- Poseidon
 * plonky2::hash::poseidon::Poseidon::poseidon (naive form: constant layer, S-box x^7, MDS layer; upstream's
 * fast partial rounds are an algebraic refactoring its own check_consistency test equates to this) -- A2. */
static inline uint64_t sbox7(uint64_t x) {
    uint64_t x2 = gl_mul(x, x), x3 = gl_mul(x2, x), x4 = gl_mul(x2, x2);
    return gl_mul(x3, x4);
}
static inline void mds_layer(uint64_t s[12]) {
    uint64_t o[12];
    for (int r = 0; r < 12; r++) {
        u128 acc = 0;
        for (int i = 0; i < 12; i++) acc += (u128)s[(i + r) % 12] * POSEIDON_MDS_CIRC[i];
        acc += (u128)s[r] * POSEIDON_MDS_DIAG[r];
        o[r] = gl_reduce128(acc);
    }
    memcpy(s, o, sizeof o);
}
void orc_poseidon(uint64_t s[12]) {
    for (int i = 0; i < 12; i++) s[i] = gl_canon(s[i]);
    int r = 0;
    for (int phase = 0; phase < 3; phase++) {
        int n = phase == 1 ? 22 : 4;
        for (int k = 0; k < n; k++, r++) {
            for (int i = 0; i < 12; i++) s[i] = gl_add(s[i], POSEIDON_RC[12 * r + i]);
            if (phase == 1) s[0] = sbox7(s[0]);
            else for (int i = 0; i < 12; i++) s[i] = sbox7(s[i]);
            mds_layer(s);
        }
    }
}
void orc_poseidon_batch(uint64_t *states, size_t n) {
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) orc_poseidon(states + 12 * i);
}
/* plonky2::hash::hashing::hash_n_to_m_no_pad: overwrite-mode sponge, rate 8, no padding -- A4 */
void orc_hash_no_pad(const uint64_t *in, size_t len, uint64_t out[4]) {
    uint64_t st[12] = {0};
    for (size_t off = 0; off < len; off += 8) {
        size_t m = len - off < 8 ? len - off : 8;
        for (size_t i = 0; i < m; i++) st[i] = gl_canon(in[off + i]);
        orc_poseidon(st);
    }
    memcpy(out, st, 4 * sizeof(uint64_t));
}
/* plonky2::hash::hashing::hash_or_noop -- A4 */
void orc_hash_or_noop(const uint64_t *in, size_t len, uint64_t out[4]) {
    if (len <= 4) {
        for (size_t i = 0; i < 4; i++) out[i] = i < len ? gl_canon(in[i]) : 0;
    } else orc_hash_no_pad(in, len, out);
}
/* PoseidonHash::two_to_one = compress(L,R) = perm(L | R | 0^4)[0..4] -- A4 */
void orc_two_to_one(const uint64_t l[4], const uint64_t r[4], uint64_t out[4]) {
    uint64_t st[12] = {l[0], l[1], l[2], l[3], r[0], r[1], r[2], r[3], 0, 0, 0, 0};
    orc_poseidon(st);
    memcpy(out, st, 4 * sizeof(uint64_t));
}
void orc_hash_rows(const uint64_t *rows, size_t n_rows, uint32_t row_len, uint64_t *out) {
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n_rows; i++) orc_hash_or_noop(rows + i * row_len, row_len, out + 4 * i);
}

/* ---------------------------------------------------------------- FFT
 * plonky2_field::fft::fft_classic: reverse_index_bits then radix-2 DIT layers, w = primitive_root_of_unity(lg n);
 * ifft_with_options: forward FFT, scale by n^-1, reverse indices 1..n-1; coset_fft: c[j] *= shift^j first;
 * coset_ifft: ifft then c[j] *= shift^-j -- A3. */
static void fft_one(uint64_t *a, uint32_t lg) {
    size_t n = (size_t)1 << lg;
    for (size_t i = 0; i < n; i++) {
        size_t j = bitrev32((uint32_t)i, lg);
        if (i < j) { uint64_t t = a[i]; a[i] = a[j]; a[j] = t; }
    }
    for (uint32_t s = 1; s <= lg; s++) {
        size_t m = (size_t)1 << (s - 1);
        uint64_t wm = gl_root(s);
        /* per-layer twiddle table */
        uint64_t *tw = (uint64_t *)malloc(m * sizeof(uint64_t));
        tw[0] = 1;
        for (size_t j = 1; j < m; j++) tw[j] = gl_mul(tw[j - 1], wm);
        for (size_t k = 0; k < n; k += 2 * m)
            for (size_t j = 0; j < m; j++) {
                uint64_t u = a[k + j], v = gl_mul(a[k + j + m], tw[j]);
                a[k + j] = gl_add(u, v);
                a[k + j + m] = gl_sub(u, v);
            }
        free(tw);
    }
}
static void fft_col(uint64_t *a, uint32_t lg, int inverse, uint64_t shift) {
    size_t n = (size_t)1 << lg;
    for (size_t i = 0; i < n; i++) a[i] = gl_canon(a[i]);
    if (!inverse) {
        if (shift > 1) { uint64_t s = 1; for (size_t j = 0; j < n; j++) { a[j] = gl_mul(a[j], s); s = gl_mul(s, shift); } }
        fft_one(a, lg);
    } else {
        fft_one(a, lg);
        uint64_t ninv = gl_inv((uint64_t)n % GL_P);
        for (size_t i = 0; i < n; i++) a[i] = gl_mul(a[i], ninv);
        for (size_t i = 1; i < n - i; i++) { uint64_t t = a[i]; a[i] = a[n - i]; a[n - i] = t; }
        if (shift > 1) { uint64_t si = gl_inv(shift), s = 1; for (size_t j = 0; j < n; j++) { a[j] = gl_mul(a[j], s); s = gl_mul(s, si); } }
    }
}
void orc_fft(uint64_t *data, uint32_t log_n, uint32_t n_cols, int32_t inverse, uint64_t shift) {
    size_t n = (size_t)1 << log_n;
#pragma omp parallel for schedule(dynamic)
    for (uint32_t c = 0; c < n_cols; c++) fft_col(data + (size_t)c * n, log_n, inverse, shift);
}

/* ---------------------------------------------------------------- Merkle tree
 * plonky2::hash::merkle_tree::MerkleTree::new / fill_digests_buf / fill_subtree -- A5.
 * digests: 2*(L - 2^h) hashes; per cap sub-tree the recursive layout left_buf | left_digest | right_digest | right_buf. */
static size_t layout_off(size_t m, uint32_t depth, size_t i) {
    /* position (in hashes) of node i at `depth` (children of the sub-tree root are depth 1) in a buffer over m leaves */
    if (depth == 1) return (m - 2) + i;
    size_t half_nodes = (size_t)1 << (depth - 1);
    if (i < half_nodes) return layout_off(m / 2, depth - 1, i);
    return m + layout_off(m / 2, depth - 1, i - half_nodes);
}
/* levels[0] = leaf hashes (L), levels[k] has L>>k hashes */
static uint64_t **build_levels(const uint64_t *leaves, size_t L, uint32_t leaf_len, uint32_t lg, uint32_t cap_height) {
    uint32_t nl = lg - cap_height + 1;
    uint64_t **lv = (uint64_t **)calloc(nl, sizeof(uint64_t *));
    lv[0] = (uint64_t *)malloc(L * 4 * sizeof(uint64_t));
    orc_hash_rows(leaves, L, leaf_len, lv[0]);
    for (uint32_t k = 1; k < nl; k++) {
        size_t cnt = L >> k;
        lv[k] = (uint64_t *)malloc(cnt * 4 * sizeof(uint64_t));
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < cnt; i++) orc_two_to_one(lv[k - 1] + 8 * i, lv[k - 1] + 8 * i + 4, lv[k] + 4 * i);
    }
    return lv;
}
static void free_levels(uint64_t **lv, uint32_t nl) { for (uint32_t k = 0; k < nl; k++) free(lv[k]); free(lv); }
static uint32_t ilog2(size_t x) { uint32_t l = 0; while (((size_t)1 << l) < x) l++; return l; }

int32_t orc_merkle_build(const uint64_t *leaves, size_t L, uint32_t leaf_len, uint32_t cap_height, uint64_t *cap_out, uint64_t *digests_out) {
    uint32_t lg = ilog2(L);
    if (((size_t)1 << lg) != L || cap_height > lg) return ZK_ERR_INVALID_ARG;
    uint32_t nl = lg - cap_height + 1;
    uint64_t **lv = build_levels(leaves, L, leaf_len, lg, cap_height);
    memcpy(cap_out, lv[nl - 1], ((size_t)4 << cap_height) * sizeof(uint64_t));
    if (digests_out && nl > 1) {
        size_t ncap = (size_t)1 << cap_height, m = L / ncap, sub_len = 2 * m - 2;
        for (uint32_t k = 0; k + 1 < nl; k++) {        /* level k sits at depth (nl-1-k) below the cap nodes */
            uint32_t depth = nl - 1 - k;
            size_t per_sub = (size_t)1 << depth;
#pragma omp parallel for schedule(static)
            for (size_t i = 0; i < (L >> k); i++) {
                size_t sub = i / per_sub, j = i % per_sub;
                memcpy(digests_out + 4 * (sub * sub_len + layout_off(m, depth, j)), lv[k] + 4 * i, 4 * sizeof(uint64_t));
            }
        }
    }
    free_levels(lv, nl);
    return ZK_OK;
}
/* tree object that keeps every level (used by the FRI commit-phase trees) */
orc_tree *orc_tree_new(const uint64_t *leaves, size_t L, uint32_t leaf_len, uint32_t cap_height) {
    uint32_t lg = ilog2(L);
    if (((size_t)1 << lg) != L || cap_height > lg) return NULL;
    orc_tree *t = (orc_tree *)calloc(1, sizeof *t);
    t->n_levels = lg - cap_height + 1; t->cap_height = cap_height; t->n_leaves = L;
    t->levels = build_levels(leaves, L, leaf_len, lg, cap_height);
    return t;
}
const uint64_t *orc_tree_cap(const orc_tree *t) { return t->levels[t->n_levels - 1]; }
void orc_tree_prove(const orc_tree *t, uint32_t idx, uint64_t *sib) {
    for (uint32_t k = 0; k + 1 < t->n_levels; k++) memcpy(sib + 4 * k, t->levels[k] + 4 * ((idx >> k) ^ 1), 4 * sizeof(uint64_t));
}
void orc_tree_free(orc_tree *t) { if (!t) return; free_levels(t->levels, t->n_levels); free(t); }

/* MerkleTree::prove: siblings from the leaf level up to (excluding) the cap level */
int32_t orc_merkle_prove(const uint64_t *leaves, size_t L, uint32_t leaf_len, uint32_t cap_height, uint32_t idx, uint64_t *sib) {
    uint32_t lg = ilog2(L);
    if (((size_t)1 << lg) != L || cap_height > lg || idx >= L) return ZK_ERR_INVALID_ARG;
    uint32_t nl = lg - cap_height + 1;
    uint64_t **lv = build_levels(leaves, L, leaf_len, lg, cap_height);
    for (uint32_t k = 0; k + 1 < nl; k++) { memcpy(sib + 4 * k, lv[k] + 4 * ((idx >> k) ^ 1), 4 * sizeof(uint64_t)); }
    free_levels(lv, nl);
    return ZK_OK;
}
/* plonky2::hash::merkle_proofs::verify_merkle_proof_to_cap */
int32_t orc_merkle_verify(const uint64_t *leaf, uint32_t leaf_len, uint32_t idx, const uint64_t *sib, uint32_t n_sib, const uint64_t *cap) {
    uint64_t cur[4], nxt[4];
    orc_hash_or_noop(leaf, leaf_len, cur);
    for (uint32_t k = 0; k < n_sib; k++) {
        if (idx & 1) orc_two_to_one(sib + 4 * k, cur, nxt); else orc_two_to_one(cur, sib + 4 * k, nxt);
        memcpy(cur, nxt, sizeof cur);
        idx >>= 1;
    }
    for (int i = 0; i < 4; i++) if (cur[i] != gl_canon(cap[4 * idx + i])) return 1;
    return 0;
}

/* ---------------------------------------------------------------- Challenger
 * plonky2::iop::challenger::Challenger<F, PoseidonHash> -- A6 */
void orc_challenger_init(orc_challenger *c) { memset(c, 0, sizeof *c); }
static void duplex(orc_challenger *c) {
    for (uint32_t i = 0; i < c->n_in; i++) c->state[i] = c->inbuf[i];
    c->n_in = 0;
    orc_poseidon(c->state);
    memcpy(c->outbuf, c->state, 8 * sizeof(uint64_t));
    c->n_out = 8;
}
void orc_challenger_observe(orc_challenger *c, const uint64_t *x, uint32_t n) {
    for (uint32_t i = 0; i < n; i++) {
        c->n_out = 0;
        c->inbuf[c->n_in++] = gl_canon(x[i]);
        if (c->n_in == 8) duplex(c);
    }
}
uint64_t orc_challenger_get(orc_challenger *c) {
    if (c->n_in || !c->n_out) duplex(c);
    return c->outbuf[--c->n_out];           /* pop from the back */
}
void orc_challenger_get_n(orc_challenger *c, uint32_t n, uint64_t *out) { for (uint32_t i = 0; i < n; i++) out[i] = orc_challenger_get(c); }
void orc_challenger_compact(orc_challenger *c, uint64_t out[12]) {
    if (c->n_in) duplex(c);
    c->n_out = 0;
    memcpy(out, c->state, 12 * sizeof(uint64_t));
}

/* ---------------------------------------------------------------- PolynomialBatch
 * plonky2::fri::oracle::PolynomialBatch::{from_values, from_coeffs, lde_values}: per column ifft -> lde(rate_bits)
 * -> coset_fft(g); transpose to rows; reverse_index_bits_in_place(rows); MerkleTree::new(rows, cap_height) -- A5 */
static orc_batch *batch_commit(uint64_t *coeffs, uint32_t log_n, uint32_t n_cols, uint32_t rate_bits, uint32_t cap_height) {
    size_t n = (size_t)1 << log_n, L = n << rate_bits;
    uint32_t lgL = log_n + rate_bits;
    orc_batch *b = (orc_batch *)calloc(1, sizeof *b);
    b->log_n = log_n; b->n_cols = n_cols; b->rate_bits = rate_bits; b->cap_height = cap_height;
    b->coeffs = coeffs;
    b->leaves = (uint64_t *)malloc(L * n_cols * sizeof(uint64_t));
#pragma omp parallel
    {
        uint64_t *tmp = (uint64_t *)malloc(L * sizeof(uint64_t));
#pragma omp for schedule(dynamic)
        for (uint32_t c = 0; c < n_cols; c++) {
            memcpy(tmp, coeffs + (size_t)c * n, n * sizeof(uint64_t));
            memset(tmp + n, 0, (L - n) * sizeof(uint64_t));
            fft_col(tmp, lgL, 0, GL_GEN);
            for (size_t r = 0; r < L; r++) b->leaves[(size_t)bitrev32((uint32_t)r, lgL) * n_cols + c] = tmp[r];
        }
        free(tmp);
    }
    b->cap = (uint64_t *)malloc(((size_t)4 << cap_height) * sizeof(uint64_t));
    size_t nd = 2 * (L - ((size_t)1 << cap_height));
    b->digests = (uint64_t *)malloc((nd ? nd : 1) * 4 * sizeof(uint64_t));
    orc_merkle_build(b->leaves, L, n_cols, cap_height, b->cap, b->digests);
    return b;
}
orc_batch *orc_batch_from_values(const uint64_t *cols, uint32_t log_n, uint32_t n_cols, uint32_t rate_bits, uint32_t cap_height) {
    size_t n = (size_t)1 << log_n;
    uint64_t *coeffs = (uint64_t *)malloc(n * n_cols * sizeof(uint64_t));
    memcpy(coeffs, cols, n * n_cols * sizeof(uint64_t));
    orc_fft(coeffs, log_n, n_cols, 1, 0);
    return batch_commit(coeffs, log_n, n_cols, rate_bits, cap_height);
}
orc_batch *orc_batch_from_coeffs(const uint64_t *cols, uint32_t log_n, uint32_t n_cols, uint32_t rate_bits, uint32_t cap_height) {
    size_t n = (size_t)1 << log_n;
    uint64_t *coeffs = (uint64_t *)malloc(n * n_cols * sizeof(uint64_t));
    for (size_t i = 0; i < n * n_cols; i++) coeffs[i] = gl_canon(cols[i]);
    return batch_commit(coeffs, log_n, n_cols, rate_bits, cap_height);
}
void orc_batch_free(orc_batch *b) { if (!b) return; free(b->coeffs); free(b->leaves); free(b->digests); free(b->cap); free(b); }
const uint64_t *orc_batch_cap(const orc_batch *b) { return b->cap; }
const uint64_t *orc_batch_leaves(const orc_batch *b) { return b->leaves; }
const uint64_t *orc_batch_coeffs(const orc_batch *b) { return b->coeffs; }
const uint64_t *orc_batch_digests(const orc_batch *b) { return b->digests; }
/* MerkleTree::prove, read from the stored digests buffer (exercises the recursive layout) */
void orc_batch_prove(const orc_batch *b, uint32_t idx, uint64_t *sib) {
    size_t L = (size_t)1 << (b->log_n + b->rate_bits);
    size_t ncap = (size_t)1 << b->cap_height, m = L / ncap, sub_len = 2 * m - 2;
    uint32_t nsib = b->log_n + b->rate_bits - b->cap_height;
    for (uint32_t k = 0; k < nsib; k++) {
        uint32_t depth = nsib - k;
        size_t node = (idx >> k) ^ 1, per_sub = (size_t)1 << depth;
        size_t sub = node / per_sub, j = node % per_sub;
        memcpy(sib + 4 * k, b->digests + 4 * (sub * sub_len + layout_off(m, depth, j)), 4 * sizeof(uint64_t));
    }
}
/* PolynomialCoeffs::to_extension().eval(point): Horner over the extension field */
void orc_batch_eval_at(const orc_batch *b, const uint64_t point[2], uint64_t *out) {
    size_t n = (size_t)1 << b->log_n;
    gl2 z = {gl_canon(point[0]), gl_canon(point[1])};
#pragma omp parallel for schedule(dynamic)
    for (uint32_t c = 0; c < b->n_cols; c++) {
        const uint64_t *p = b->coeffs + (size_t)c * n;
        gl2 acc = {0, 0};
        for (size_t i = n; i-- > 0;) { acc = gl2_mul(acc, z); acc.a = gl_add(acc.a, p[i]); }
        out[2 * c] = acc.a; out[2 * c + 1] = acc.b;
    }
}

/* ---------------------------------------------------------------- self test */
int32_t orc_selftest(void) {
    static const uint64_t edge[] = {0, 1, 2, 0xFFFFFFFFULL, 0x100000000ULL, 0x8000000000000000ULL, GL_P - 1, GL_P, GL_P + 1, ~0ULL};
    uint64_t x = 0x9E3779B97F4A7C15ULL;
    for (int i = 0; i < 10; i++) for (int j = 0; j < 10; j++)
        if (gl_mul(edge[i], edge[j]) != gl_mul_ref(edge[i], edge[j])) return 1;
    for (int i = 0; i < 100000; i++) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17; uint64_t a = x;
        x ^= x << 13; x ^= x >> 7; x ^= x << 17; uint64_t b = x;
        if (gl_mul(a, b) != gl_mul_ref(a, b)) return 2;
        if (gl_add(a, b) != (uint64_t)(((u128)a + b) % GL_P)) return 3;
        if (gl_sub(a, b) != (uint64_t)(((u128)(a % GL_P) + GL_P - (b % GL_P)) % GL_P)) return 4;
    }
    if (gl_pow(GL_POW2_GEN, 1ULL << 31) != GL_P - 1) return 5;
    if (gl_root(6) != 8) return 6;                 /* w_64 = 2^3 */
    return 0;
}
