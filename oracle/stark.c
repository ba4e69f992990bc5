/* ORACLE -- test infrastructure, not product code (see oracle.h for who may call this).
 * STARK prover: plain-C restatement of starky 0.4.0 `prover::prove_with_commitment` as driven by
 * evm_arithmetization 0.1.3 `prover::prove_with_traces`, and of plonky2 0.2.2 `fri::oracle::prove_openings`,
 * `fri::prover::{fri_proof, fri_committed_trees, fri_proof_of_work, fri_prover_query_rounds}`.
 * Sources absent from /root/reference (Cargo.lock:1675-1677, :3441-3443, :4529-4531); spec = SURVEY.md A6-A9;
 * entry from the reference: /root/reference/ops/src/lib.rs:52.   PARITY UNPINNED above the permutation.
 */
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <omp.h>
#include "gl.h"
#include "oracle.h"

void zk_stark_config_standard_fast_oracle(zk_stark_config *c) {
    c->security_bits = 100; c->num_challenges = 2; c->rate_bits = 1; c->cap_height = 4;
    c->proof_of_work_bits = 16; c->fri_arity_bits = 4; c->fri_final_poly_bits = 5; c->num_query_rounds = 84;
}

/* ------------------------------------------------------------ synthetic witness */
static uint64_t splitmix_field(uint64_t *st) {
    for (;;) {
        uint64_t z = (*st += 0x9E3779B97F4A7C15ULL);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
        z ^= z >> 31;
        if (z < GL_P) return z;
    }
}
void orc_synth_trace(uint32_t n_cols, uint32_t log_n, uint64_t seed, uint64_t *t, uint64_t pub[3]) {
    size_t n = (size_t)1 << log_n;
    uint32_t T = n_cols / 3;
    for (uint32_t c = 0; c < n_cols; c++) { uint64_t st = seed + c; t[(size_t)c * n] = splitmix_field(&st); }
#pragma omp parallel for schedule(dynamic)
    for (uint32_t k = 0; k < T + (n_cols - 3 * T); k++) {
        if (k < T) {
            uint64_t *a = t + (size_t)(3 * k) * n, *b = a + n, *c = b + n;
            for (size_t r = 0; r + 1 < n; r++) {
                uint64_t ab = gl_mul(a[r], b[r]);
                a[r + 1] = gl_add(ab, c[r]);
                b[r + 1] = gl_add(gl_mul(b[r], b[r]), a[r]);
                c[r + 1] = gl_add(gl_mul(ab, c[r]), 1);
            }
        } else {
            uint64_t *a = t + (size_t)(3 * T + (k - T)) * n;
            for (size_t r = 0; r + 1 < n; r++) a[r + 1] = gl_add(a[r], 1);
        }
    }
    pub[0] = t[0]; pub[1] = n_cols > 1 ? t[n] : 0; pub[2] = t[n - 1];
}

/* ------------------------------------------------------------ auxiliary columns (starky::lookup / CTL partial_sums, A9) */
static uint32_t n_aux_cols(const zk_air *air, uint32_t num_challenges) { return zk_air_n_aux(air, num_challenges); }

/* starky::lookup::lookup_helper_columns (A9): per challenge [helper_b .., Z];  helper_b = sum over a batch of two looked columns of
 * filter / (f + beta);  Z[0] = 0, Z[r+1] = Z[r] + sum_b helper_b[r] - m[r] / (t[r] + beta) */
static int lookup_columns(const zk_lookup *lk, uint32_t nch, const uint64_t *ch, const uint64_t *trace, size_t n, uint64_t *aux) {
    uint32_t LH = (lk->n_cols + 1) / 2;
    int bad = 0;
    for (uint32_t j = 0; j < nch; j++) {
        uint64_t beta = ch[2 * j];
        uint64_t *base = aux + (size_t)j * (LH + 1) * n, *Z = base + (size_t)LH * n;
        for (uint32_t b = 0; b < LH; b++)
            for (size_t r = 0; r < n; r++) {
                uint64_t h = 0;
                for (uint32_t i = 2 * b; i < 2 * b + 2 && i < lk->n_cols; i++) {
                    uint64_t d = gl_add(trace[(size_t)lk->cols[i].col * n + r], beta);
                    if (d == 0) bad = 1;
                    uint64_t f = lk->cols[i].filter == ZK_NO_FILTER ? 1 : gl_canon(trace[(size_t)lk->cols[i].filter * n + r]);
                    h = gl_add(h, gl_mul(f, gl_inv(d)));
                }
                base[(size_t)b * n + r] = h;
            }
        uint64_t run = 0;
        for (size_t r = 0; r < n; r++) {
            Z[r] = run;
            uint64_t d = gl_add(trace[(size_t)lk->table_col * n + r], beta);
            if (d == 0) bad = 1;
            uint64_t delta = gl_sub(0, gl_mul(gl_canon(trace[(size_t)lk->freq_col * n + r]), gl_inv(d)));
            for (uint32_t b = 0; b < LH; b++) delta = gl_add(delta, base[(size_t)b * n + r]);
            run = gl_add(run, delta);
        }
    }
    return bad;
}

int32_t orc_aux_columns(const zk_air *air, uint32_t nch, const uint64_t *ch, const uint64_t *trace, uint32_t log_n, uint64_t *aux) {
    size_t n = (size_t)1 << log_n;
    uint32_t H = air->n_helpers;
    int bad = 0;
    if (air->lookup && air->lookup->n_cols) {
        bad |= lookup_columns(air->lookup, nch, ch, trace, n, aux);
        aux += (size_t)nch * ((air->lookup->n_cols + 1) / 2 + 1) * n;         /* the CTL-shaped columns follow */
    }
    if (!H) return bad ? ZK_ERR_ZERO_INVERSE : ZK_OK;
    for (uint32_t j = 0; j < nch; j++) {
        uint64_t beta = ch[2 * j], gamma = ch[2 * j + 1];
#pragma omp parallel for schedule(dynamic)
        for (uint32_t m = 0; m < H; m++) {
            const zk_helper_desc *d = &air->helpers[m];
            uint64_t *h = aux + (size_t)(j * H + m) * n;
            /* batch_multiplicative_inverse (Montgomery trick) over the 2n denominators of this column */
            uint64_t *den = (uint64_t *)malloc(2 * n * sizeof(uint64_t)), *pre = (uint64_t *)malloc(2 * n * sizeof(uint64_t));
            /* partial_sums: combine(cols) = cols[0] + beta cols[1] + beta^2 cols[2] + .. + gamma for each of the two column sets */
            const zk_helper_ext *x = air->helpers_ext ? &air->helpers_ext[m] : NULL;
            for (size_t r = 0; r < n; r++)
                for (int k = 0; k < 2; k++) {
                    uint32_t cols[2 + ZK_CTL_MAX_EXTRA], nc = 0;
                    cols[nc++] = k ? d->a1 : d->a0; cols[nc++] = k ? d->b1 : d->b0;
                    if (x) for (uint32_t i = 0; i < x->n_extra[k]; i++) cols[nc++] = x->extra[k][i];
                    uint64_t acc = 0, bp = 1;
                    for (uint32_t i = 0; i < nc; i++) { acc = gl_add(acc, gl_mul(bp, trace[(size_t)cols[i] * n + r])); bp = gl_mul(bp, beta); }
                    den[2 * r + k] = gl_add(acc, gamma);
                }
            uint64_t acc = 1;
            for (size_t i = 0; i < 2 * n; i++) { pre[i] = acc; if (den[i] == 0) bad = 1; acc = gl_mul(acc, den[i]); }
            uint64_t inv = gl_inv(acc);
            for (size_t i = 2 * n; i-- > 0;) { uint64_t t = gl_mul(inv, pre[i]); inv = gl_mul(inv, den[i]); den[i] = t; }
            for (size_t r = 0; r < n; r++) {        /* sum of filter / combine over the two pairs (filter absent: 1) */
                uint64_t f0 = (x && x->filter[0] != ZK_NO_FILTER) ? trace[(size_t)x->filter[0] * n + r] : 1;
                uint64_t f1 = (x && x->filter[1] != ZK_NO_FILTER) ? trace[(size_t)x->filter[1] * n + r] : 1;
                h[r] = gl_add(gl_mul(f0, den[2 * r]), gl_mul(f1, den[2 * r + 1]));
            }
            free(den); free(pre);
        }
        /* Z[r] = sum_{s>=r} sum_m h_m[s]  (suffix sum, computed from the last row backwards) */
        uint64_t *Z = aux + (size_t)(nch * H + j) * n;
        uint64_t run = 0;
        for (size_t r = n; r-- > 0;) {
            uint64_t s = 0;
            for (uint32_t m = 0; m < H; m++) s = gl_add(s, aux[(size_t)(j * H + m) * n + r]);
            run = gl_add(run, s);
            Z[r] = run;
        }
    }
    return bad ? ZK_ERR_ZERO_INVERSE : ZK_OK;
}

/* ------------------------------------------------------------ constraint program interpreter
 * stands in for Stark::eval_packed_generic + ConstraintConsumer (A9): acc_j = acc_j*alpha_j + c*filter */
typedef struct {
    const uint64_t *local, *next;   /* rows of trace|aux values (base field) */
    const uint64_t *pub, *chal;
    uint64_t z_last, l_first, l_last;
} eval_ctx;

static inline uint64_t opnd_base(const zk_air *air, const eval_ctx *e, const uint64_t *regs, uint32_t o) {
    uint32_t k = o >> 24, i = o & 0xFFFFFF;
    switch (k) {
    case ZK_OPND_REG: return regs[i];
    case ZK_OPND_LOCAL: return e->local[i];
    case ZK_OPND_NEXT: return e->next[i];
    case ZK_OPND_CONST: return gl_canon(air->constants[i]);
    case ZK_OPND_PUBLIC: return gl_canon(e->pub[i]);
    default: return gl_canon(e->chal[i]);
    }
}
static void eval_base(const zk_air *air, const eval_ctx *e, uint32_t nch, const uint64_t *alphas, uint64_t *acc, uint64_t *regs) {
    for (uint32_t j = 0; j < nch; j++) acc[j] = 0;
    for (uint32_t p = 0; p < air->n_instrs; p++) {
        const zk_instr *in = &air->instrs[p];
        uint64_t a = opnd_base(air, e, regs, in->a);
        if (in->op <= ZK_OP_MUL) {
            uint64_t b = opnd_base(air, e, regs, in->b);
            regs[in->dst & 0xFFFFFF] = in->op == ZK_OP_ADD ? gl_add(a, b) : in->op == ZK_OP_SUB ? gl_sub(a, b) : gl_mul(a, b);
        } else {
            uint64_t f = in->op == ZK_OP_EMIT ? 1 : in->op == ZK_OP_EMIT_TRANSITION ? e->z_last : in->op == ZK_OP_EMIT_FIRST ? e->l_first : e->l_last;
            uint64_t c = in->op == ZK_OP_EMIT ? a : gl_mul(a, f);
            for (uint32_t j = 0; j < nch; j++) acc[j] = gl_add(gl_mul(acc[j], alphas[j]), c);
        }
    }
}

/* ------------------------------------------------------------ proof writer */
typedef struct { uint64_t *p; size_t cap, len; int overflow; } wr;
static void w1(wr *w, uint64_t v) { if (w->len < w->cap) w->p[w->len] = v; else w->overflow = 1; w->len++; }
static void wn(wr *w, const uint64_t *v, size_t n) { for (size_t i = 0; i < n; i++) w1(w, v[i]); }

/* FriReductionStrategy::ConstantArityBits(arity_bits, final_poly_bits) (A8).  Upstream computes in usize and asserts degree_bits >=
 * arity_bits: a (config, size) pair whose subtraction would underflow or whose assert would fail PANICS there; returns 0 for those. */
static int fri_arities_checked(const zk_stark_config *cfg, uint32_t degree_bits, uint32_t *out, uint32_t *n_out) {
    uint32_t d = degree_bits, k = 0;
    int ok = cfg->fri_arity_bits != 0;
    while (ok && d > cfg->fri_final_poly_bits) {
        if (d + cfg->rate_bits < cfg->fri_arity_bits) { ok = 0; break; }
        if (d + cfg->rate_bits - cfg->fri_arity_bits < cfg->cap_height) break;
        if (d < cfg->fri_arity_bits || k >= 64) { ok = 0; break; }
        out[k++] = cfg->fri_arity_bits; d -= cfg->fri_arity_bits;
    }
    *n_out = k;
    return ok;
}
static void fri_arities(const zk_stark_config *cfg, uint32_t degree_bits, uint32_t *out, uint32_t *n_out) { (void)fri_arities_checked(cfg, degree_bits, out, n_out); }

static void ext_fft_coset(gl2 *v, uint32_t lg, uint64_t shift) {
    /* coset_fft over the extension: the shift and the roots are base-field, so it acts component-wise */
    size_t n = (size_t)1 << lg;
    uint64_t *t = (uint64_t *)malloc(2 * n * sizeof(uint64_t));
    for (size_t i = 0; i < n; i++) { t[i] = v[i].a; t[n + i] = v[i].b; }
    orc_fft(t, lg, 2, 0, shift);
    for (size_t i = 0; i < n; i++) { v[i].a = t[i]; v[i].b = t[n + i]; }
    free(t);
}

typedef struct {
    const orc_table *in;
    uint32_t n_aux;
    orc_batch *trace_b, *aux_b, *quot_b;
    uint64_t *aux_vals;
} tbl_state;

/* plonky2::fri::structure::FriInstanceInfo restated: oracles + batches of (point, [(oracle, polynomial)]) */
typedef struct {
    uint32_t n_oracles; const orc_batch *const *oracles;
    uint32_t n_batches; const gl2 *points; const uint32_t *cnt; const zk_fri_poly *const *polys;
} fri_instance;

/* PolynomialBatch::prove_openings up to fri_proof: final_poly coefficients (length L, upper part zero) and its values on the LDE
 * coset (natural order) for a given alpha */
static void fri_final_poly(const zk_stark_config *cfg, const fri_instance *I, gl2 alpha, gl2 **coeffs_out, gl2 **values_out) {
    uint32_t log_n = I->oracles[0]->log_n, lgL = log_n + cfg->rate_bits;
    size_t n = (size_t)1 << log_n, L = (size_t)1 << lgL;
    uint32_t nb = I->n_batches;
    gl2 *final_poly = (gl2 *)calloc(L, sizeof(gl2));
    gl2 *comp = (gl2 *)malloc(n * sizeof(gl2));
    for (uint32_t b = 0; b < nb; b++) {
        /* list the polynomials of this batch */
        uint32_t cnt = I->cnt[b];
        const uint64_t **polys = (const uint64_t **)malloc((cnt ? cnt : 1) * sizeof(uint64_t *));
        for (uint32_t k = 0; k < cnt; k++) polys[k] = I->oracles[I->polys[b][k].oracle_index]->coeffs + (size_t)I->polys[b][k].polynomial_index * n;
        /* ReducingFactor::reduce_polys_base: sum_j alpha^j f_j (Horner from the last polynomial) */
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < n; i++) {
            gl2 acc = {0, 0};
            for (uint32_t j = cnt; j-- > 0;) { acc = gl2_mul(acc, alpha); acc.a = gl_add(acc.a, polys[j][i]); }
            comp[i] = acc;
        }
        free(polys);
        /* divide_by_linear(point): Horner scan from the top coefficient, drop the remainder, push a zero */
        gl2 z = I->points[b], acc = {0, 0};
        gl2 *q = (gl2 *)malloc(n * sizeof(gl2));
        for (size_t i = n; i-- > 0;) { acc = gl2_add(gl2_mul(acc, z), comp[i]); if (i > 0) q[i - 1] = acc; }
        q[n - 1].a = 0; q[n - 1].b = 0;
        /* alpha.shift_poly(final_poly); final_poly += quotient */
        gl2 sh = gl2_pow(alpha, cnt);
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < n; i++) final_poly[i] = gl2_add(b == 0 ? (gl2){0, 0} : gl2_mul(final_poly[i], sh), q[i]);
        free(q);
    }
    free(comp);
    /* lde(rate_bits).coset_fft(g) */
    gl2 *values = (gl2 *)malloc(L * sizeof(gl2));
    memcpy(values, final_poly, L * sizeof(gl2));
    ext_fft_coset(values, lgL, GL_GEN);
    *coeffs_out = final_poly; *values_out = values;
}

/* PolynomialBatch::prove_openings + fri_proof */
static int32_t fri_prove_instance(const zk_stark_config *cfg, const fri_instance *I, orc_challenger *ch, wr *w) {
    uint32_t log_n = I->oracles[0]->log_n, lgL = log_n + cfg->rate_bits;
    size_t L = (size_t)1 << lgL;
    uint64_t ab[2]; orc_challenger_get_n(ch, 2, ab);
    gl2 alpha = {ab[0], ab[1]};
    gl2 *final_poly, *values;
    fri_final_poly(cfg, I, alpha, &final_poly, &values);
    gl2 *coeffs = final_poly;           /* length L, upper part zero */

    /* fri_committed_trees */
    uint32_t ar[64], nl; fri_arities(cfg, log_n, ar, &nl);
    w1(w, nl);
    size_t cur = L; uint32_t cur_lg = lgL;
    uint64_t shift = GL_GEN;
    uint64_t **layer_leaves = (uint64_t **)calloc(nl ? nl : 1, sizeof(uint64_t *));
    orc_tree **layer_tree = (orc_tree **)calloc(nl ? nl : 1, sizeof(orc_tree *));
    for (uint32_t l = 0; l < nl; l++) {
        uint32_t arity = 1u << ar[l];
        /* reverse_index_bits_in_place(values); leaves = chunks of `arity` ext values, flattened */
        size_t n_leaves = cur / arity;
        uint64_t *lv = (uint64_t *)malloc(cur * 2 * sizeof(uint64_t));
        for (size_t i = 0; i < cur; i++) { gl2 v = values[bitrev32((uint32_t)i, cur_lg)]; lv[2 * i] = v.a; lv[2 * i + 1] = v.b; }
        layer_tree[l] = orc_tree_new(lv, n_leaves, 2 * arity, cfg->cap_height);
        const uint64_t *cap = orc_tree_cap(layer_tree[l]);
        wn(w, cap, (size_t)4 << cfg->cap_height);
        orc_challenger_observe(ch, cap, 4u << cfg->cap_height);
        layer_leaves[l] = lv;
        uint64_t bb[2]; orc_challenger_get_n(ch, 2, bb);
        gl2 beta = {bb[0], bb[1]};
        /* coeffs'[i] = reduce_with_powers(coeffs[arity*i .. arity*(i+1)], beta) */
        size_t nxt = cur / arity;
        for (size_t i = 0; i < nxt; i++) {
            gl2 acc = {0, 0};
            for (uint32_t t = arity; t-- > 0;) acc = gl2_add(gl2_mul(acc, beta), coeffs[(size_t)arity * i + t]);
            coeffs[i] = acc;
        }
        shift = gl_pow(shift, arity);
        cur = nxt; cur_lg -= ar[l];
        memcpy(values, coeffs, cur * sizeof(gl2));
        ext_fft_coset(values, cur_lg, shift);
    }
    /* coeffs.truncate(len >> rate_bits); observe_extension_elements(final coeffs) */
    size_t flen = cur >> cfg->rate_bits;
    w1(w, flen);
    for (size_t i = 0; i < flen; i++) { uint64_t e[2] = {coeffs[i].a, coeffs[i].b}; wn(w, e, 2); orc_challenger_observe(ch, e, 2); }

    /* fri_proof_of_work: smallest witness (upstream's rayon find_any is non-deterministic) */
    uint64_t st[12]; memcpy(st, ch->state, sizeof st);
    for (uint32_t i = 0; i < ch->n_in; i++) st[i] = ch->inbuf[i];
    uint32_t pos = ch->n_in;
    uint64_t witness = 0; int found = 0;
    uint32_t min_lz = cfg->proof_of_work_bits;
    for (uint64_t base = 0; !found && base < ((uint64_t)1 << 40); base += 1 << 16) {
        uint64_t best = ~0ULL;
#pragma omp parallel for schedule(static) reduction(min : best)
        for (uint64_t c = base; c < base + (1 << 16); c++) {
            uint64_t s2[12]; memcpy(s2, st, sizeof s2);
            s2[pos] = c;
            orc_poseidon(s2);
            uint64_t r = s2[7];
            uint32_t lz = r ? (uint32_t)__builtin_clzll(r) : 64;
            if (lz >= min_lz && c < best) best = c;
        }
        if (best != ~0ULL) { witness = best; found = 1; }
    }
    if (!found) return ZK_ERR_POW;
    orc_challenger_observe(ch, &witness, 1);
    uint64_t resp = orc_challenger_get(ch);
    if ((resp ? (uint32_t)__builtin_clzll(resp) : 64) < min_lz) return ZK_ERR_POW;
    w1(w, witness);

    /* fri_prover_query_rounds */
    uint32_t nq = cfg->num_query_rounds;
    uint64_t *qs = (uint64_t *)malloc(nq * sizeof(uint64_t));
    orc_challenger_get_n(ch, nq, qs);
    const orc_batch *const *oracles = I->oracles; uint32_t no = I->n_oracles;
    uint32_t npath = lgL - cfg->cap_height;
    uint64_t *sib = (uint64_t *)malloc((size_t)(npath + 1) * 4 * sizeof(uint64_t));
    for (uint32_t q = 0; q < nq; q++) {
        size_t x = (size_t)(qs[q] % L);
        for (uint32_t o = 0; o < no; o++) {
            wn(w, oracles[o]->leaves + x * oracles[o]->n_cols, oracles[o]->n_cols);
            orc_batch_prove(oracles[o], (uint32_t)x, sib);
            wn(w, sib, (size_t)npath * 4);
        }
        uint32_t lg = lgL;
        for (uint32_t l = 0; l < nl; l++) {
            uint32_t arity = 1u << ar[l];
            size_t leaf = x >> ar[l];
            wn(w, layer_leaves[l] + leaf * 2 * arity, 2 * arity);
            uint32_t depth = lg - ar[l];
            orc_tree_prove(layer_tree[l], (uint32_t)leaf, sib);
            wn(w, sib, (size_t)(depth - cfg->cap_height) * 4);
            x = leaf; lg = depth;
        }
    }
    free(sib); free(qs);
    for (uint32_t l = 0; l < nl; l++) { free(layer_leaves[l]); orc_tree_free(layer_tree[l]); }
    free(layer_leaves); free(layer_tree); free(values); free(final_poly);
    return ZK_OK;
}

/* the STARK instance of one table: batches (zeta: trace|aux|quot) (g*zeta: trace|aux) (1: ctl Zs) over oracles trace[, aux], quot */
static int32_t fri_prove(const zk_stark_config *cfg, tbl_state *T, orc_challenger *ch, gl2 zeta, wr *w) {
    uint32_t log_n = T->in->log_n, nch = cfg->num_challenges;
    uint32_t C = T->in->air->n_trace_cols, A = T->n_aux, Q = T->quot_b->n_cols;
    const orc_batch *oracles[3]; uint32_t no = 0, o_aux = 0, o_q;
    oracles[no++] = T->trace_b; if (A) { o_aux = no; oracles[no++] = T->aux_b; } o_q = no; oracles[no++] = T->quot_b;
    gl2 points[3] = {zeta, gl2_scale(zeta, gl_root(log_n)), {1, 0}};
    uint32_t cnt[3] = {C + A + Q, C + A, nch};
    zk_fri_poly *pl[3];
    for (uint32_t b = 0; b < 3; b++) pl[b] = (zk_fri_poly *)malloc((cnt[b] ? cnt[b] : 1) * sizeof(zk_fri_poly));
    for (uint32_t b = 0; b < 2; b++) {
        uint32_t k = 0;
        for (uint32_t c = 0; c < C; c++) pl[b][k++] = (zk_fri_poly){0, c};
        for (uint32_t c = 0; c < A; c++) pl[b][k++] = (zk_fri_poly){o_aux, c};
        if (b == 0) for (uint32_t c = 0; c < Q; c++) pl[b][k++] = (zk_fri_poly){o_q, c};
    }
    if (A) for (uint32_t j = 0; j < nch; j++) pl[2][j] = (zk_fri_poly){o_aux, A - nch + j};
    fri_instance I = {no, oracles, A ? 3u : 2u, points, cnt, (const zk_fri_poly *const *)pl};
    int32_t rc = fri_prove_instance(cfg, &I, ch, w);
    for (uint32_t b = 0; b < 3; b++) free(pl[b]);
    return rc;
}

/* starky::prover::compute_quotient_polys (A9/A7 step 4) -> 2*num_challenges coefficient polys of length n */
static uint64_t *quotient_polys(const zk_stark_config *cfg, tbl_state *T, const uint64_t *chal, const uint64_t *alphas) {
    const zk_air *air = T->in->air;
    uint32_t log_n = T->in->log_n, lgL = log_n + cfg->rate_bits, nch = cfg->num_challenges;
    size_t n = (size_t)1 << log_n, L = (size_t)1 << lgL;
    uint32_t C = air->n_trace_cols, A = T->n_aux;
    /* lagrange_first / lagrange_last = selector(n, idx).lde_onto_coset(rate_bits) */
    uint64_t *lag = (uint64_t *)calloc(2 * L, sizeof(uint64_t));
    {
        uint64_t *sel = (uint64_t *)calloc(2 * n, sizeof(uint64_t));
        sel[0] = 1; sel[n + n - 1] = 1;
        orc_fft(sel, log_n, 2, 1, 0);
        memcpy(lag, sel, n * sizeof(uint64_t)); memcpy(lag + L, sel + n, n * sizeof(uint64_t));
        orc_fft(lag, lgL, 2, 0, GL_GEN);
        free(sel);
    }
    /* ZeroPolyOnCoset: Z_H(x) = x^n - 1 takes 2^rate_bits values on the coset */
    uint32_t rate = 1u << cfg->rate_bits;
    uint64_t zh_inv[16];
    uint64_t gn = gl_pow(GL_GEN, n), wr_ = gl_root(cfg->rate_bits);
    for (uint32_t i = 0; i < rate; i++) zh_inv[i] = gl_inv(gl_sub(gl_mul(gn, gl_pow(wr_, i)), 1));
    uint64_t w_L = gl_root(lgL), last = gl_inv(gl_root(log_n));
    uint64_t *qv = (uint64_t *)malloc((size_t)nch * L * sizeof(uint64_t));
    uint32_t next_step = rate;
#pragma omp parallel
    {
        uint64_t *rowl = (uint64_t *)malloc((C + A + 1) * sizeof(uint64_t)), *rown = (uint64_t *)malloc((C + A + 1) * sizeof(uint64_t));
        uint64_t *regs = (uint64_t *)malloc((air->n_regs + 1) * sizeof(uint64_t));
        uint64_t acc[16];
#pragma omp for schedule(static)
        for (size_t i = 0; i < L; i++) {
            size_t rl = bitrev32((uint32_t)i, lgL), rn = bitrev32((uint32_t)((i + next_step) % L), lgL);
            memcpy(rowl, T->trace_b->leaves + rl * C, C * sizeof(uint64_t));
            memcpy(rown, T->trace_b->leaves + rn * C, C * sizeof(uint64_t));
            if (A) { memcpy(rowl + C, T->aux_b->leaves + rl * A, A * sizeof(uint64_t)); memcpy(rown + C, T->aux_b->leaves + rn * A, A * sizeof(uint64_t)); }
            uint64_t x = gl_mul(GL_GEN, gl_pow(w_L, i));
            eval_ctx e = {rowl, rown, T->in->public_inputs, chal, gl_sub(x, last), lag[i], lag[L + i]};
            eval_base(air, &e, nch, alphas, acc, regs);
            for (uint32_t j = 0; j < nch; j++) qv[(size_t)j * L + i] = gl_mul(acc[j], zh_inv[i % rate]);
        }
        free(rowl); free(rown); free(regs);
    }
    free(lag);
    /* coset_ifft(g) then split each into quotient_degree_factor chunks of n coefficients */
    orc_fft(qv, lgL, nch, 1, GL_GEN);
    return qv; /* layout: poly (j, chunk) at qv + (j*rate + chunk)*n  -- contiguous since L = rate*n */
}

int32_t orc_prove(const zk_stark_config *cfg, uint32_t nt, const orc_table *tables, uint64_t *out, size_t cap_words, size_t *len_words) {
    /* the same parameter window as the product (include/zkgl.h, zk_prove_tables), so that refusals can be compared too */
    if (cfg->num_challenges == 0 || cfg->num_challenges > 4 || cfg->rate_bits > 3 || cfg->cap_height > 8 || cfg->fri_arity_bits < 2 || cfg->fri_arity_bits > 6) return ZK_ERR_INVALID_ARG;
    for (uint32_t t = 0; t < nt; t++) {
        uint32_t ar_[64], nl_;
        if (tables[t].log_n > 26 || tables[t].log_n + cfg->rate_bits < cfg->cap_height || !fri_arities_checked(cfg, tables[t].log_n, ar_, &nl_)) return ZK_ERR_INVALID_ARG;
    }
    if ((cfg->rate_bits != 1)) return ZK_ERR_INVALID_ARG; /* quotient_degree_bits == rate_bits == 1 only (constraint degree 3) */
    wr w = {out, cap_words, 0, 0};
    uint32_t nch = cfg->num_challenges, capw = 4u << cfg->cap_height;
    tbl_state *T = (tbl_state *)calloc(nt, sizeof(tbl_state));
    orc_challenger ch; orc_challenger_init(&ch);
    int32_t rc = ZK_OK;
    w1(&w, 0x31465250474C4B5AULL); /* "ZKGLPRF1" */
    w1(&w, nt);
    /* commit every trace, observe every trace cap, then the public values (A7 prelude) */
    for (uint32_t t = 0; t < nt; t++) {
        T[t].in = &tables[t];
        T[t].n_aux = n_aux_cols(tables[t].air, nch);
        T[t].trace_b = orc_batch_from_values(tables[t].trace, tables[t].log_n, tables[t].air->n_trace_cols, cfg->rate_bits, cfg->cap_height);
        orc_challenger_observe(&ch, T[t].trace_b->cap, capw);
    }
    for (uint32_t t = 0; t < nt; t++) orc_challenger_observe(&ch, tables[t].public_inputs, tables[t].air->n_public_inputs);
    uint64_t chal[16];
    orc_challenger_get_n(&ch, 2 * nch, chal); /* for each set: beta then gamma */
    for (uint32_t t = 0; t < nt && rc == ZK_OK; t++) {
        const zk_air *air = tables[t].air;
        uint32_t log_n = tables[t].log_n, C = air->n_trace_cols, A = T[t].n_aux;
        size_t n = (size_t)1 << log_n;
        uint64_t st12[12]; orc_challenger_compact(&ch, st12); /* init_challenger_state */
        w1(&w, log_n); w1(&w, C); w1(&w, A);
        wn(&w, T[t].trace_b->cap, capw);
        if (A) {
            T[t].aux_vals = (uint64_t *)malloc((size_t)A * n * sizeof(uint64_t));
            rc = orc_aux_columns(air, nch, chal, tables[t].trace, log_n, T[t].aux_vals);
            if (rc) break;
            T[t].aux_b = orc_batch_from_values(T[t].aux_vals, log_n, A, cfg->rate_bits, cfg->cap_height);
            orc_challenger_observe(&ch, T[t].aux_b->cap, capw);
            wn(&w, T[t].aux_b->cap, capw);
        }
        uint64_t alphas[16]; orc_challenger_get_n(&ch, nch, alphas);
        uint64_t *qc = quotient_polys(cfg, &T[t], chal, alphas);
        uint32_t Q = nch << cfg->rate_bits;
        T[t].quot_b = orc_batch_from_coeffs(qc, log_n, Q, cfg->rate_bits, cfg->cap_height);
        free(qc);
        orc_challenger_observe(&ch, T[t].quot_b->cap, capw);
        wn(&w, T[t].quot_b->cap, capw);
        uint64_t zb[2]; orc_challenger_get_n(&ch, 2, zb);
        gl2 zeta = {zb[0], zb[1]};
        if (gl2_eq(gl2_pow(zeta, n), (gl2){1, 0})) { rc = ZK_ERR_ZETA_IN_SUBGROUP; break; }
        /* StarkOpeningSet::new */
        uint64_t g_n = gl_root(log_n);
        uint64_t zn[2] = {gl_mul(zeta.a, g_n), gl_mul(zeta.b, g_n)};
        uint64_t *o_local = (uint64_t *)malloc(2 * (size_t)(C + 1) * 8), *o_next = (uint64_t *)malloc(2 * (size_t)(C + 1) * 8);
        uint64_t *o_aux = (uint64_t *)malloc(2 * (size_t)(A + 1) * 8), *o_auxn = (uint64_t *)malloc(2 * (size_t)(A + 1) * 8);
        uint64_t *o_q = (uint64_t *)malloc(2 * (size_t)Q * 8);
        orc_batch_eval_at(T[t].trace_b, zb, o_local);
        orc_batch_eval_at(T[t].trace_b, zn, o_next);
        if (A) { orc_batch_eval_at(T[t].aux_b, zb, o_aux); orc_batch_eval_at(T[t].aux_b, zn, o_auxn); }
        orc_batch_eval_at(T[t].quot_b, zb, o_q);
        uint64_t zs_first[16];
        for (uint32_t j = 0; j < nch && A; j++) {      /* eval at F::ONE in the base field = sum of coefficients */
            const uint64_t *p = T[t].aux_b->coeffs + (size_t)(A - nch + j) * n;
            uint64_t s = 0; for (size_t i = 0; i < n; i++) s = gl_add(s, p[i]);
            zs_first[j] = s;
        }
        wn(&w, o_local, 2 * C); wn(&w, o_next, 2 * C);
        wn(&w, o_aux, 2 * A); wn(&w, o_auxn, 2 * A);
        if (A) wn(&w, zs_first, nch);
        wn(&w, o_q, 2 * Q);
        /* challenger.observe_openings(&openings.to_fri_openings()) */
        orc_challenger_observe(&ch, o_local, 2 * C); orc_challenger_observe(&ch, o_aux, 2 * A); orc_challenger_observe(&ch, o_q, 2 * Q);
        orc_challenger_observe(&ch, o_next, 2 * C); orc_challenger_observe(&ch, o_auxn, 2 * A);
        for (uint32_t j = 0; j < nch && A; j++) { uint64_t e[2] = {zs_first[j], 0}; orc_challenger_observe(&ch, e, 2); }
        free(o_local); free(o_next); free(o_aux); free(o_auxn); free(o_q);
        rc = fri_prove(cfg, &T[t], &ch, zeta, &w);
    }
    for (uint32_t t = 0; t < nt; t++) { orc_batch_free(T[t].trace_b); orc_batch_free(T[t].aux_b); orc_batch_free(T[t].quot_b); free(T[t].aux_vals); }
    free(T);
    if (len_words) *len_words = w.len;
    if (rc) return rc;
    return w.overflow ? ZK_ERR_BUFFER_TOO_SMALL : ZK_OK;
}

size_t orc_proof_size_words(const zk_stark_config *cfg, uint32_t nt, const orc_table *tables) {
    size_t tot = 2;
    for (uint32_t t = 0; t < nt; t++) {   /* no such tree / upstream panics in the reduction strategy: orc_prove refuses */
        uint32_t ar_[64], nl_;
        if (tables[t].log_n + cfg->rate_bits < cfg->cap_height || !fri_arities_checked(cfg, tables[t].log_n, ar_, &nl_)) return 0;
    }
    uint32_t nch = cfg->num_challenges, capw = 4u << cfg->cap_height;
    for (uint32_t t = 0; t < nt; t++) {
        uint32_t C = tables[t].air->n_trace_cols, A = n_aux_cols(tables[t].air, nch), Q = nch << cfg->rate_bits;
        uint32_t log_n = tables[t].log_n, lgL = log_n + cfg->rate_bits;
        uint32_t ar[64], nl; fri_arities(cfg, log_n, ar, &nl);
        tot += 3 + (size_t)capw * (A ? 3 : 2) + 4 * (size_t)C + 4 * (size_t)A + (A ? nch : 0) + 2 * (size_t)Q;
        tot += 1 + (size_t)nl * capw;
        uint32_t d = log_n; for (uint32_t l = 0; l < nl; l++) d -= ar[l];
        tot += 1 + 2 * ((size_t)1 << d) + 1;
        size_t per_q = (size_t)(C + A + Q) + (size_t)(A ? 3 : 2) * 4 * (lgL - cfg->cap_height);
        uint32_t lg = lgL;
        for (uint32_t l = 0; l < nl; l++) { lg -= ar[l]; per_q += (2u << ar[l]) + 4 * (size_t)(lg - cfg->cap_height); }
        tot += per_q * cfg->num_query_rounds;
    }
    return tot;
}

/* ------------------------------------------------------------ finer-grained entry points (parity tests of the C-ABI pieces) */
/* starky::prover::compute_quotient_polys: out = (num_challenges << rate_bits) x n coefficients, natural order */
int32_t orc_quotient(const zk_stark_config *cfg, const zk_air *air, uint32_t log_n, const orc_batch *trace_b, const orc_batch *aux_b,
                     const uint64_t *public_inputs, const uint64_t *challenges, const uint64_t *alphas, uint64_t *out) {
    orc_table in = {air, log_n, 0, NULL, public_inputs};
    tbl_state T; memset(&T, 0, sizeof T);
    T.in = &in; T.n_aux = aux_b ? aux_b->n_cols : 0; T.trace_b = (orc_batch *)trace_b; T.aux_b = (orc_batch *)aux_b;
    uint64_t *q = quotient_polys(cfg, &T, challenges, alphas);
    memcpy(out, q, ((size_t)cfg->num_challenges << (log_n + cfg->rate_bits)) * sizeof(uint64_t));
    free(q);
    return ZK_OK;
}
/* plonky2::fri::prover::fri_proof_of_work: smallest witness w with leading_zeros(permute(state | w at pos)[7]) >= min_zeros */
int32_t orc_pow_grind(const uint64_t state[12], uint32_t pos, uint32_t min_zeros, uint64_t *witness) {
    for (uint64_t c = 0; c < ((uint64_t)1 << 40); c++) {
        uint64_t s2[12]; memcpy(s2, state, sizeof s2);
        s2[pos] = c;
        orc_poseidon(s2);
        uint64_t r = s2[7];
        if ((r ? (uint32_t)__builtin_clzll(r) : 64) >= min_zeros) { *witness = c; return ZK_OK; }
    }
    return ZK_ERR_POW;
}
/* PolynomialBatch::prove_openings + fri_proof for one table: FRI proof words (n_layers ... queries) appended to out */
int32_t orc_fri_prove(const zk_stark_config *cfg, const orc_batch *trace_b, const orc_batch *aux_b, const orc_batch *quot_b,
                      orc_challenger *ch, const uint64_t zeta[2], uint64_t *out, size_t cap_words, size_t *len_words) {
    zk_air air; memset(&air, 0, sizeof air);
    air.n_trace_cols = trace_b->n_cols;
    orc_table in = {&air, trace_b->log_n, 0, NULL, NULL};
    tbl_state T; memset(&T, 0, sizeof T);
    T.in = &in; T.n_aux = aux_b ? aux_b->n_cols : 0;
    T.trace_b = (orc_batch *)trace_b; T.aux_b = (orc_batch *)aux_b; T.quot_b = (orc_batch *)quot_b;
    wr w = {out, cap_words, 0, 0};
    gl2 z = {gl_canon(zeta[0]), gl_canon(zeta[1])};
    int32_t rc = fri_prove(cfg, &T, ch, z, &w);
    if (len_words) *len_words = w.len;
    if (rc) return rc;
    return w.overflow ? ZK_ERR_BUFFER_TOO_SMALL : ZK_OK;
}
/* PolynomialBatch::prove_openings for any FriInstanceInfo (counterpart of zk_fri_prove_instance) */
/* the counterpart of zk_fri_combine: lde_final_values in leaf (bit-reversed) order, [re, im] interleaved, for a given alpha */
int32_t orc_fri_combine(const zk_stark_config *cfg, uint32_t n_oracles, const orc_batch *const *oracles, uint32_t n_batches,
                        const zk_fri_batch *batches, const uint64_t alpha[2], uint64_t *out) {
    if (!cfg || !oracles || !batches || !alpha || !out || n_oracles == 0 || n_batches == 0 || n_batches > 4) return ZK_ERR_INVALID_ARG;
    gl2 pts[4]; uint32_t cnt[4]; const zk_fri_poly *pl[4];
    for (uint32_t b = 0; b < n_batches; b++) { pts[b] = (gl2){gl_canon(batches[b].point[0]), gl_canon(batches[b].point[1])}; cnt[b] = batches[b].n_polys; pl[b] = batches[b].polys; }
    fri_instance I = {n_oracles, oracles, n_batches, pts, cnt, pl};
    gl2 *co, *va;
    fri_final_poly(cfg, &I, (gl2){gl_canon(alpha[0]), gl_canon(alpha[1])}, &co, &va);
    uint32_t lgL = oracles[0]->log_n + cfg->rate_bits;
    size_t L = (size_t)1 << lgL;
    for (size_t i = 0; i < L; i++) { gl2 v = va[bitrev32((uint32_t)i, lgL)]; out[2 * i] = v.a; out[2 * i + 1] = v.b; }
    free(co); free(va);
    return ZK_OK;
}

int32_t orc_fri_prove_instance(const zk_stark_config *cfg, uint32_t n_oracles, const orc_batch *const *oracles, uint32_t n_batches,
                               const zk_fri_batch *batches, orc_challenger *ch, uint64_t *out, size_t cap_words, size_t *len_words) {
    if (!n_oracles || n_oracles > 8 || !n_batches || n_batches > 4) return ZK_ERR_INVALID_ARG;
    gl2 points[4]; uint32_t cnt[4]; const zk_fri_poly *pl[4];
    for (uint32_t b = 0; b < n_batches; b++) {
        points[b] = (gl2){gl_canon(batches[b].point[0]), gl_canon(batches[b].point[1])};
        cnt[b] = batches[b].n_polys; pl[b] = batches[b].polys;
        for (uint32_t k = 0; k < cnt[b]; k++)
            if (pl[b][k].oracle_index >= n_oracles || pl[b][k].polynomial_index >= oracles[pl[b][k].oracle_index]->n_cols) return ZK_ERR_INVALID_ARG;
    }
    for (uint32_t o = 1; o < n_oracles; o++) if (oracles[o]->log_n != oracles[0]->log_n) return ZK_ERR_INVALID_ARG;
    { uint32_t ar_[64], nl_; if (oracles[0]->log_n + cfg->rate_bits < cfg->cap_height || !fri_arities_checked(cfg, oracles[0]->log_n, ar_, &nl_)) return ZK_ERR_INVALID_ARG; }
    fri_instance I = {n_oracles, oracles, n_batches, points, cnt, pl};
    wr w = {out, cap_words, 0, 0};
    int32_t rc = fri_prove_instance(cfg, &I, ch, &w);
    if (len_words) *len_words = w.len;
    if (rc) return rc;
    return w.overflow ? ZK_ERR_BUFFER_TOO_SMALL : ZK_OK;
}
