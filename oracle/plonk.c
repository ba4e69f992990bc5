/* ORACLE -- test infrastructure, not product code (see oracle.h for who may call this).
 * Plonk-shaped prover and verifier: plain-C restatement of plonky2 0.2.2 `plonk::prover::prove_with_partition_witness`
 * (wires commit, permutation partial products and Z, vanishing polynomial on the 8n coset, quotient chunks, OpeningSet,
 * prove_openings over the four oracles) and of `plonk::verifier::verify_with_challenges` + `fri::verifier::verify_fri_proof`
 * for ANY FriInstanceInfo.  What the reference runs after the STARK proofs inside
 *   /root/reference/ops/src/lib.rs:52,72,95   generate_txn_proof / generate_agg_proof / generate_block_proof
 * with CircuitConfig::standard_recursion_config().  Sources absent from /root/reference (Cargo.lock:3441-3443); spec = SURVEY.md
 * A9 (last paragraph) and 8f-2; the circuit itself is plain data (zk_plonk_circuit): PARITY UNPINNED, evidence = the verifier
 * below accepts what the prover emits and rejects single-word corruptions.
 */
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include "gl.h"
#include "oracle.h"

typedef struct { uint64_t *p; size_t cap, len; } pw;
static void p1(pw *w, uint64_t v) { if (w->len < w->cap) w->p[w->len] = v; w->len++; }
static void pn(pw *w, const uint64_t *v, size_t n) { for (size_t i = 0; i < n; i++) p1(w, v[i]); }

static uint32_t brev(uint32_t x, uint32_t bits) { uint32_t r = 0; for (uint32_t i = 0; i < bits; i++) { r = (r << 1) | (x & 1); x >>= 1; } return r; }

/* FriReductionStrategy::ConstantArityBits(arity_bits, final_poly_bits) (A8).  Upstream computes in usize and asserts degree_bits >=
 * arity_bits: a (config, size) pair whose subtraction would underflow or whose assert would fail PANICS there; returns 0 for those. */
static int fri_arities_p_checked(const zk_stark_config *cfg, uint32_t degree_bits, uint32_t *out, uint32_t *n_out) {
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
static void fri_arities_p(const zk_stark_config *cfg, uint32_t degree_bits, uint32_t *out, uint32_t *n_out) { (void)fri_arities_p_checked(cfg, degree_bits, out, n_out); }

/* gate program over the base field: LOCAL i < W is wire i, otherwise constant column i - W.  Emitted terms are appended to out. */
/* number of gate constraint terms: ZK_OP_EMIT writes term (number of EMITs before it), ZK_OP_EMIT_SLOT term `dst`; terms ADD UP
 * (upstream: evaluate_gate_constraints_base_batch sums the filtered constraints of all gates position by position) */
static uint32_t n_gate_terms(const zk_air *g) {
    uint32_t c = 0, hi = 0;
    for (uint32_t p = 0; p < g->n_instrs; p++) {
        if (g->instrs[p].op == ZK_OP_EMIT) c++;
        else if (g->instrs[p].op == ZK_OP_EMIT_SLOT && g->instrs[p].dst + 1 > hi) hi = g->instrs[p].dst + 1;
    }
    return c > hi ? c : hi;
}
static uint32_t gates_base(const zk_air *g, const uint64_t *wires, const uint64_t *consts, uint32_t W, const uint64_t *pih, uint64_t *regs, uint64_t *out) {
    uint32_t n_out = 0, NG = n_gate_terms(g);
    for (uint32_t j = 0; j < NG; j++) out[j] = 0;
    for (uint32_t p = 0; p < g->n_instrs; p++) {
        const zk_instr *in = &g->instrs[p];
        uint64_t v[2]; uint32_t ops[2] = {in->a, in->b};
        for (int s = 0; s < 2; s++) {
            uint32_t k = ops[s] >> 24, i = ops[s] & 0xFFFFFF;
            v[s] = k == ZK_OPND_REG ? regs[i] : k == ZK_OPND_LOCAL ? (i < W ? wires[i] : consts[i - W]) : k == ZK_OPND_CONST ? gl_canon(g->constants[i]) : k == ZK_OPND_PUBLIC ? pih[i] : 0;
            if (in->op > ZK_OP_MUL) break;
        }
        if (in->op <= ZK_OP_MUL) regs[in->dst & 0xFFFFFF] = in->op == ZK_OP_ADD ? gl_add(v[0], v[1]) : in->op == ZK_OP_SUB ? gl_sub(v[0], v[1]) : gl_mul(v[0], v[1]);
        else if (in->op == ZK_OP_EMIT) { out[n_out] = gl_add(out[n_out], v[0]); n_out++; }
        else out[in->dst] = gl_add(out[in->dst], v[0]);
    }
    return NG;
}
static uint32_t gates_ext(const zk_air *g, const gl2 *wires, const gl2 *consts, uint32_t W, const uint64_t *pih, gl2 *regs, gl2 *out) {
    uint32_t n_out = 0, NG = n_gate_terms(g);
    for (uint32_t j = 0; j < NG; j++) out[j] = gl2_from(0);
    for (uint32_t p = 0; p < g->n_instrs; p++) {
        const zk_instr *in = &g->instrs[p];
        gl2 v[2]; uint32_t ops[2] = {in->a, in->b};
        for (int s = 0; s < 2; s++) {
            uint32_t k = ops[s] >> 24, i = ops[s] & 0xFFFFFF;
            v[s] = k == ZK_OPND_REG ? regs[i] : k == ZK_OPND_LOCAL ? (i < W ? wires[i] : consts[i - W]) : k == ZK_OPND_CONST ? gl2_from(g->constants[i]) : k == ZK_OPND_PUBLIC ? gl2_from(pih[i]) : gl2_from(0);
            if (in->op > ZK_OP_MUL) break;
        }
        if (in->op <= ZK_OP_MUL) regs[in->dst & 0xFFFFFF] = in->op == ZK_OP_ADD ? gl2_add(v[0], v[1]) : in->op == ZK_OP_SUB ? gl2_sub(v[0], v[1]) : gl2_mul(v[0], v[1]);
        else if (in->op == ZK_OP_EMIT) { out[n_out] = gl2_add(out[n_out], v[0]); n_out++; }
        else out[in->dst] = gl2_add(out[in->dst], v[0]);
    }
    return NG;
}

static int circuit_ok(const zk_stark_config *cfg, const zk_plonk_circuit *c) {
    if (!cfg || !c || !c->gates || !c->constants || !c->sigmas) return 0;
    if (c->degree_bits < 2 || c->degree_bits > 24 || c->num_wires == 0 || c->num_routed_wires == 0 || c->num_routed_wires > c->num_wires) return 0;
    if (c->n_domain_separator > 4096 || (c->n_domain_separator && !c->domain_separator) || n_gate_terms(c->gates) > 4096) return 0;
    if (c->quotient_degree_factor != (1u << cfg->rate_bits) || c->num_routed_wires % c->quotient_degree_factor) return 0;
    if (cfg->num_challenges == 0 || cfg->num_challenges > 4) return 0;
    { uint32_t ar_[64], nl_; if (c->degree_bits + cfg->rate_bits < cfg->cap_height || !fri_arities_p_checked(cfg, c->degree_bits, ar_, &nl_)) return 0; }
    for (uint32_t p = 0; p < c->gates->n_instrs; p++) {
        const zk_instr *in = &c->gates->instrs[p];
        if (in->op > ZK_OP_EMIT && in->op != ZK_OP_EMIT_SLOT) return 0;
        if (in->op == ZK_OP_EMIT_SLOT && in->dst >= 4096) return 0;
        uint32_t ops[2] = {in->a, in->b};
        for (int s = 0; s < (in->op <= ZK_OP_MUL ? 2 : 1); s++) {
            uint32_t k = ops[s] >> 24, i = ops[s] & 0xFFFFFF;
            if (k == ZK_OPND_REG ? i >= c->gates->n_regs : k == ZK_OPND_LOCAL ? i >= c->num_wires + c->num_constants : k == ZK_OPND_CONST ? i >= c->gates->n_constants : k == ZK_OPND_PUBLIC ? i >= 4 : 1) return 0;
        }
        if (in->op <= ZK_OP_MUL && (in->dst >> 24 || (in->dst & 0xFFFFFF) >= c->gates->n_regs)) return 0;
    }
    return 1;
}

size_t orc_plonk_proof_size_words(const zk_stark_config *cfg, const zk_plonk_circuit *c) {
    if (!circuit_ok(cfg, c)) return 0;
    uint32_t nch = cfg->num_challenges, qdf = c->quotient_degree_factor, npp = c->num_routed_wires / qdf - 1, capw = 4u << cfg->cap_height;
    uint32_t NC = c->num_constants, R = c->num_routed_wires, W = c->num_wires, k = c->degree_bits, lgL = k + cfg->rate_bits;
    size_t open = 2 * (size_t)(NC + R + W + nch + nch + nch * npp + nch * qdf);
    uint32_t ar[64], nl; fri_arities_p(cfg, k, ar, &nl);
    uint32_t d = k; for (uint32_t l = 0; l < nl; l++) d -= ar[l];
    size_t per_q = (size_t)(NC + R) + W + (size_t)nch * (1 + npp) + (size_t)nch * qdf + (size_t)4 * 4 * (lgL - cfg->cap_height);
    uint32_t lg = lgL;
    for (uint32_t l = 0; l < nl; l++) { lg -= ar[l]; per_q += (2u << ar[l]) + 4 * (size_t)(lg - cfg->cap_height); }
    return 5 + 3 * (size_t)capw + open + 1 + (size_t)nl * capw + 1 + 2 * ((size_t)1 << d) + 1 + per_q * cfg->num_query_rounds;
}

/* prover data: constants | sigmas committed once per circuit (CircuitBuilder::build) */
struct orc_plonk_state { zk_plonk_circuit c; orc_batch *cs; uint64_t digest[4]; };

orc_plonk_state *orc_plonk_setup(const zk_stark_config *cfg, const zk_plonk_circuit *c) {
    if (!circuit_ok(cfg, c)) return NULL;
    size_t n = (size_t)1 << c->degree_bits;
    uint32_t NC = c->num_constants, R = c->num_routed_wires;
    uint64_t *cols = (uint64_t *)malloc((size_t)(NC + R) * n * sizeof(uint64_t));
    memcpy(cols, c->constants, (size_t)NC * n * sizeof(uint64_t));
    memcpy(cols + (size_t)NC * n, c->sigmas, (size_t)R * n * sizeof(uint64_t));
    orc_plonk_state *st = (orc_plonk_state *)calloc(1, sizeof *st);
    st->c = *c;
    st->cs = orc_batch_from_values(cols, c->degree_bits, NC + R, cfg->rate_bits, cfg->cap_height);
    free(cols);
    /* CircuitBuilder::build: circuit_digest = hash_no_pad(constants_sigmas_cap | hash_pad(domain_separator) | degree_bits), with
     * hash_pad(x) = hash_no_pad(x | 1 | 0.. | 1) padded to a multiple of the rate (8): the empty separator hashes [1,0,0,0,0,0,0,1] */
    uint32_t capw = 4u << cfg->cap_height;
    uint64_t *in = (uint64_t *)calloc(capw + 5, sizeof(uint64_t));
    memcpy(in, orc_batch_cap(st->cs), capw * sizeof(uint64_t));
    {
        uint32_t nd = c->n_domain_separator, np = nd + 1;
        while ((np + 1) % ORC_HASH_PAD_MULTIPLE) np++;
        np++;
        uint64_t *pad = (uint64_t *)calloc(np, sizeof(uint64_t));
        for (uint32_t i = 0; i < nd; i++) pad[i] = gl_canon(c->domain_separator[i]);
        pad[nd] = 1; pad[np - 1] = 1;
        orc_hash_no_pad(pad, np, in + capw);
        free(pad);
    }
    in[capw + 4] = c->degree_bits;
    orc_hash_no_pad(in, capw + 5, st->digest);
    free(in);
    return st;
}
void orc_plonk_state_free(orc_plonk_state *st) { if (st) { orc_batch_free(st->cs); free(st); } }
void orc_plonk_digest(const orc_plonk_state *st, uint64_t out[4]) { memcpy(out, st->digest, 4 * sizeof(uint64_t)); }

int32_t orc_prove_plonk(const zk_stark_config *cfg, orc_plonk_state *st, const uint64_t *wires, const uint64_t *pub, uint32_t n_pub,
                        uint64_t *out, size_t cap_words, size_t *len_words) {
    const zk_plonk_circuit *c = &st->c;
    const uint32_t nch = cfg->num_challenges, qdf = c->quotient_degree_factor, R = c->num_routed_wires, W = c->num_wires, NC = c->num_constants;
    const uint32_t k = c->degree_bits, lgL = k + cfg->rate_bits, nchunks = R / qdf, npp = nchunks - 1, capw = 4u << cfg->cap_height;
    const size_t n = (size_t)1 << k, L = (size_t)1 << lgL;
    size_t total = orc_plonk_proof_size_words(cfg, c);
    if (len_words) *len_words = total;
    if (cap_words < total) return ZK_ERR_BUFFER_TOO_SMALL;
    pw w = {out, cap_words, 0};
    p1(&w, 0x314B4C504C474B5AULL); p1(&w, k); p1(&w, W); p1(&w, R); p1(&w, NC);          /* "ZKGLPLK1" */
    int32_t rc = ZK_OK;
    /* ---- wires commitment; transcript */
    orc_batch *wb = orc_batch_from_values(wires, k, W, cfg->rate_bits, cfg->cap_height);
    pn(&w, orc_batch_cap(wb), capw);
    orc_challenger ch; orc_challenger_init(&ch);
    uint64_t pih[4]; orc_hash_no_pad(pub, n_pub, pih);
    orc_challenger_observe(&ch, st->digest, 4);
    orc_challenger_observe(&ch, pih, 4);
    orc_challenger_observe(&ch, orc_batch_cap(wb), capw);
    uint64_t betas[4], gammas[4];
    orc_challenger_get_n(&ch, nch, betas);
    orc_challenger_get_n(&ch, nch, gammas);
    /* ---- wires_permutation_partial_products_and_zs: per row the products of num/den over chunks of qdf wires, then the running
     * product Z(g x) = Z(x) * prod(chunks); polynomials [Z_0 .. Z_{nch-1}, partial products of challenge 0 .., of challenge 1 ..] */
    uint32_t NZ = nch * (1 + npp);
    uint64_t *zpp = (uint64_t *)malloc((size_t)NZ * n * sizeof(uint64_t));
    uint64_t w_n = gl_root(k);
    uint64_t *kis = (uint64_t *)malloc(R * sizeof(uint64_t));
    { uint64_t x = 1; for (uint32_t j = 0; j < R; j++) { kis[j] = x; x = gl_mul(x, GL_GEN); } }
    for (uint32_t ci = 0; ci < nch; ci++) {
        uint64_t z = 1, x = 1;
        for (size_t r = 0; r < n; r++) {
            uint64_t acc = z;
            zpp[(size_t)ci * n + r] = z;
            for (uint32_t cidx = 0; cidx < nchunks; cidx++) {
                uint64_t num = 1, den = 1;
                for (uint32_t t = 0; t < qdf; t++) {
                    uint32_t j = cidx * qdf + t;
                    uint64_t wv = gl_canon(wires[(size_t)j * n + r]);
                    num = gl_mul(num, gl_add(gl_add(wv, gl_mul(betas[ci], gl_mul(kis[j], x))), gammas[ci]));
                    uint64_t dd = gl_add(gl_add(wv, gl_mul(betas[ci], gl_canon(c->sigmas[(size_t)j * n + r]))), gammas[ci]);
                    if (dd == 0) rc = ZK_ERR_ZERO_INVERSE;
                    den = gl_mul(den, dd);
                }
                acc = gl_mul(acc, gl_mul(num, gl_inv(den)));
                if (cidx < npp) zpp[((size_t)nch + (size_t)ci * npp + cidx) * n + r] = acc;
            }
            z = acc;
            x = gl_mul(x, w_n);
        }
    }
    if (rc) { free(zpp); free(kis); orc_batch_free(wb); return rc; }
    orc_batch *zb = orc_batch_from_values(zpp, k, NZ, cfg->rate_bits, cfg->cap_height);
    free(zpp);
    pn(&w, orc_batch_cap(zb), capw);
    orc_challenger_observe(&ch, orc_batch_cap(zb), capw);
    uint64_t alphas[4]; orc_challenger_get_n(&ch, nch, alphas);
    /* ---- compute_quotient_polys: vanishing terms [L_0 (Z_i - 1)] [partial-product checks of every challenge] [gate constraints],
     * reduce_with_powers(terms, alpha_j) = sum_t term_t alpha_j^t, times Z_H(x)^-1, on the coset g*<w_L>, L = qdf * n */
    const uint32_t NG = n_gate_terms(c->gates), NT = nch + nch * nchunks + NG;
    uint64_t *qv = (uint64_t *)malloc((size_t)nch * L * sizeof(uint64_t));
    {
        uint64_t *l0 = (uint64_t *)calloc(L, sizeof(uint64_t));
        { uint64_t *sel = (uint64_t *)calloc(n, sizeof(uint64_t)); sel[0] = 1; orc_fft(sel, k, 1, 1, 0); memcpy(l0, sel, n * sizeof(uint64_t)); free(sel); orc_fft(l0, lgL, 1, 0, GL_GEN); }
        uint64_t zh_inv[16], gn = gl_pow(GL_GEN, n), wr_ = gl_root(cfg->rate_bits), w_L = gl_root(lgL);
        for (uint32_t i = 0; i < qdf; i++) zh_inv[i] = gl_inv(gl_sub(gl_mul(gn, gl_pow(wr_, i)), 1));
        const uint64_t *lw = orc_batch_leaves(wb), *lz = orc_batch_leaves(zb), *lc = orc_batch_leaves(st->cs);
#pragma omp parallel
        {
            uint64_t *terms = (uint64_t *)malloc((NT + 1) * sizeof(uint64_t)), *regs = (uint64_t *)malloc((c->gates->n_regs + 1) * sizeof(uint64_t));
#pragma omp for schedule(static)
            for (size_t i = 0; i < L; i++) {
                size_t rl = brev((uint32_t)i, lgL), rn = brev((uint32_t)((i + qdf) % L), lgL);
                const uint64_t *wr = lw + rl * W, *zr = lz + rl * NZ, *zn = lz + rn * NZ, *cr = lc + rl * (NC + R);
                uint64_t x = gl_mul(GL_GEN, gl_pow(w_L, i));
                uint32_t t = 0;
                for (uint32_t ci = 0; ci < nch; ci++) terms[t++] = gl_mul(l0[i], gl_sub(zr[ci], 1));
                for (uint32_t ci = 0; ci < nch; ci++)
                    for (uint32_t cidx = 0; cidx < nchunks; cidx++) {
                        uint64_t num = 1, den = 1;
                        for (uint32_t u = 0; u < qdf; u++) {
                            uint32_t j = cidx * qdf + u;
                            num = gl_mul(num, gl_add(gl_add(wr[j], gl_mul(betas[ci], gl_mul(kis[j], x))), gammas[ci]));
                            den = gl_mul(den, gl_add(gl_add(wr[j], gl_mul(betas[ci], cr[NC + j])), gammas[ci]));
                        }
                        uint64_t prev = cidx == 0 ? zr[ci] : zr[nch + ci * npp + cidx - 1];
                        uint64_t next = cidx == nchunks - 1 ? zn[ci] : zr[nch + ci * npp + cidx];
                        terms[t++] = gl_sub(gl_mul(prev, num), gl_mul(next, den));
                    }
                t += gates_base(c->gates, wr, cr, W, pih, regs, terms + t);
                for (uint32_t j = 0; j < nch; j++) {
                    uint64_t acc = 0;
                    for (uint32_t u = t; u-- > 0;) acc = gl_add(gl_mul(acc, alphas[j]), terms[u]);
                    qv[(size_t)j * L + i] = gl_mul(acc, zh_inv[i % qdf]);
                }
            }
            free(terms); free(regs);
        }
        free(l0);
    }
    orc_fft(qv, lgL, nch, 1, GL_GEN);            /* coset_ifft: each challenge's L coefficients = qdf chunks of n */
    /* quotient degree < qdf * n is what trim_to_len asserts upstream; the top coefficient chunk is simply committed like the others */
    orc_batch *qb = orc_batch_from_coeffs(qv, k, nch * qdf, cfg->rate_bits, cfg->cap_height);
    free(qv);
    pn(&w, orc_batch_cap(qb), capw);
    orc_challenger_observe(&ch, orc_batch_cap(qb), capw);
    uint64_t zt[2]; orc_challenger_get_n(&ch, 2, zt);
    gl2 zeta = {zt[0], zt[1]};
    gl2 gz = gl2_scale(zeta, w_n);
    if (gl2_eq(gl2_pow(zeta, n), gl2_from(1))) rc = ZK_ERR_ZETA_IN_SUBGROUP;
    /* ---- OpeningSet::new + observe_openings(to_fri_openings): zeta batch, then the zeta*g batch (Zs only) */
    size_t n_open0 = (size_t)(NC + R) + W + NZ + (size_t)nch * qdf;
    uint64_t *op = (uint64_t *)malloc(2 * (n_open0 + nch) * sizeof(uint64_t));
    {
        uint64_t *tmp = (uint64_t *)malloc(2 * (size_t)(NC + R + W + NZ + nch * qdf) * sizeof(uint64_t));
        size_t o = 0;
        orc_batch_eval_at(st->cs, (uint64_t *)&zeta, tmp); memcpy(op + o, tmp, 2 * (size_t)(NC + R) * sizeof(uint64_t)); o += 2 * (size_t)(NC + R);
        orc_batch_eval_at(wb, (uint64_t *)&zeta, tmp); memcpy(op + o, tmp, 2 * (size_t)W * sizeof(uint64_t)); o += 2 * (size_t)W;
        orc_batch_eval_at(zb, (uint64_t *)&zeta, tmp);
        memcpy(op + o, tmp, 2 * (size_t)nch * sizeof(uint64_t)); o += 2 * (size_t)nch;                     /* plonk_zs */
        uint64_t *tmpn = (uint64_t *)malloc(2 * (size_t)NZ * sizeof(uint64_t));
        orc_batch_eval_at(zb, (uint64_t *)&gz, tmpn);
        memcpy(op + o, tmpn, 2 * (size_t)nch * sizeof(uint64_t)); o += 2 * (size_t)nch;                    /* plonk_zs_next */
        memcpy(op + o, tmp + 2 * nch, 2 * (size_t)nch * npp * sizeof(uint64_t)); o += 2 * (size_t)nch * npp; /* partial_products */
        free(tmpn);
        orc_batch_eval_at(qb, (uint64_t *)&zeta, tmp); memcpy(op + o, tmp, 2 * (size_t)nch * qdf * sizeof(uint64_t)); o += 2 * (size_t)nch * qdf;
        free(tmp);
        pn(&w, op, o);
        /* observe: zeta batch = constants, sigmas, wires, zs, partial products, quotient; then zs_next */
        size_t oz = 2 * (size_t)(NC + R + W), on = oz + 2 * nch, opp = on + 2 * nch;
        orc_challenger_observe(&ch, op, (uint32_t)(oz + 2 * nch));
        orc_challenger_observe(&ch, op + opp, (uint32_t)(2 * (size_t)nch * npp + 2 * (size_t)nch * qdf));
        orc_challenger_observe(&ch, op + on, 2 * nch);
    }
    free(op);
    /* ---- prove_openings over [constants_sigmas, wires, zs_partial_products, quotient] */
    if (rc == ZK_OK) {
        const orc_batch *oracles[4] = {st->cs, wb, zb, qb};
        uint32_t cnt0 = (NC + R) + W + NZ + nch * qdf;
        zk_fri_poly *p0 = (zk_fri_poly *)malloc(cnt0 * sizeof(zk_fri_poly)), *p1_ = (zk_fri_poly *)malloc(nch * sizeof(zk_fri_poly));
        uint32_t q = 0;
        for (uint32_t i = 0; i < NC + R; i++) p0[q++] = (zk_fri_poly){0, i};
        for (uint32_t i = 0; i < W; i++) p0[q++] = (zk_fri_poly){1, i};
        for (uint32_t i = 0; i < NZ; i++) p0[q++] = (zk_fri_poly){2, i};
        for (uint32_t i = 0; i < nch * qdf; i++) p0[q++] = (zk_fri_poly){3, i};
        for (uint32_t i = 0; i < nch; i++) p1_[i] = (zk_fri_poly){2, i};
        zk_fri_batch bt[2] = {{{zeta.a, zeta.b}, cnt0, 0, p0}, {{gz.a, gz.b}, nch, 0, p1_}};
        size_t fl = 0;
        rc = orc_fri_prove_instance(cfg, 4, oracles, 2, bt, &ch, out + w.len, cap_words - w.len, &fl);
        w.len += fl;
        free(p0); free(p1_);
    }
    free(kis);
    orc_batch_free(wb); orc_batch_free(zb); orc_batch_free(qb);
    if (rc == ZK_OK && w.len != total) rc = ZK_ERR_INVALID_ARG;
    return rc;
}

/* ------------------------------------------------------------------------------------------------ verifier */
typedef struct { const uint64_t *p; size_t len, pos; int bad; } prd;
static uint64_t q1(prd *r) { if (r->pos < r->len) return r->p[r->pos++]; r->bad = 1; return 0; }
static const uint64_t *qn(prd *r, size_t n) { if (r->pos + n <= r->len) { const uint64_t *p = r->p + r->pos; r->pos += n; return p; } r->bad = 1; return NULL; }
static int canon_n(const uint64_t *p, size_t n) { for (size_t i = 0; i < n; i++) if (p[i] >= GL_P) return 0; return 1; }

/* fri::verifier::verify_fri_proof for any instance: oracles given by cap and width; batches by point, polynomial list and the
 * claimed openings (one ext value per listed polynomial).  Reads the FriProof words from r, continues the transcript in ch. */
static int32_t fri_verify_instance(const zk_stark_config *cfg, uint32_t log_n, uint32_t no, const uint64_t *const *caps, const uint32_t *ncols,
                                   uint32_t nb, const gl2 *points, const uint32_t *cnt, const zk_fri_poly *const *polys, const gl2 *const *opens,
                                   orc_challenger *ch, prd *r) {
    const uint32_t lgL = log_n + cfg->rate_bits, capw = 4u << cfg->cap_height;
    const size_t L = (size_t)1 << lgL;
    uint64_t ab[2]; orc_challenger_get_n(ch, 2, ab);
    gl2 alpha = {ab[0], ab[1]};
    uint32_t ar[64], nl; fri_arities_p(cfg, log_n, ar, &nl);
    if (q1(r) != nl) return 230;
    const uint64_t *layer_caps[64]; gl2 betas[64];
    for (uint32_t l = 0; l < nl; l++) {
        layer_caps[l] = qn(r, capw);
        if (!layer_caps[l] || !canon_n(layer_caps[l], capw)) return 231;
        orc_challenger_observe(ch, layer_caps[l], capw);
        uint64_t bb[2]; orc_challenger_get_n(ch, 2, bb);
        betas[l] = (gl2){bb[0], bb[1]};
    }
    uint32_t dfin = log_n; for (uint32_t l = 0; l < nl; l++) dfin -= ar[l];
    size_t flen = (size_t)q1(r);
    if (flen != ((size_t)1 << dfin)) return 232;
    const uint64_t *final_poly = qn(r, 2 * flen);
    if (!final_poly || !canon_n(final_poly, 2 * flen)) return 233;
    orc_challenger_observe(ch, final_poly, (uint32_t)(2 * flen));
    uint64_t witness = q1(r);
    if (witness >= GL_P) return 234;
    orc_challenger_observe(ch, &witness, 1);
    uint64_t resp = orc_challenger_get(ch);
    if ((resp ? (uint32_t)__builtin_clzll(resp) : 64) < cfg->proof_of_work_bits) return 235;
    /* PrecomputedReducedOpenings: reduce each batch's openings with alpha (Horner from the last) */
    gl2 red[8];
    for (uint32_t b = 0; b < nb; b++) { gl2 acc = {0, 0}; for (uint32_t k = cnt[b]; k-- > 0;) acc = gl2_add(gl2_mul(acc, alpha), opens[b][k]); red[b] = acc; }
    uint32_t nq = cfg->num_query_rounds;
    uint64_t *qs = (uint64_t *)malloc(nq * sizeof(uint64_t));
    orc_challenger_get_n(ch, nq, qs);
    const uint32_t npath = lgL - cfg->cap_height;
    const uint64_t w_L = gl_root(lgL);
    int32_t rc = 0;
    for (uint32_t q = 0; q < nq && !rc; q++) {
        size_t x = (size_t)(qs[q] % L);
        const uint64_t *leaf[8];
        for (uint32_t o = 0; o < no; o++) {
            leaf[o] = qn(r, ncols[o]);
            const uint64_t *path = qn(r, 4 * (size_t)npath);
            if (r->bad) { rc = 240; break; }
            if (!canon_n(leaf[o], ncols[o])) { rc = 241; break; }
            if (orc_merkle_verify(leaf[o], ncols[o], (uint32_t)x, path, npath, caps[o])) { rc = 242; break; }
        }
        if (rc) break;
        /* fri_combine_initial */
        uint64_t sx = gl_mul(GL_GEN, gl_pow(w_L, brev((uint32_t)x, lgL)));
        gl2 sum = {0, 0};
        for (uint32_t b = 0; b < nb; b++) {
            gl2 acc = {0, 0};
            for (uint32_t k = cnt[b]; k-- > 0;) { acc = gl2_mul(acc, alpha); acc.a = gl_add(acc.a, leaf[polys[b][k].oracle_index][polys[b][k].polynomial_index]); }
            gl2 num = gl2_sub(acc, red[b]), den = gl2_sub(gl2_from(sx), points[b]);
            sum = gl2_mul(sum, gl2_pow(alpha, cnt[b]));
            sum = gl2_add(sum, gl2_mul(num, gl2_inv(den)));
        }
        gl2 old_eval = sum;
        uint32_t lg = lgL;
        for (uint32_t l = 0; l < nl; l++) {
            uint32_t arity = 1u << ar[l];
            size_t coset = x >> ar[l], within = x & (arity - 1);
            const uint64_t *ev = qn(r, 2 * (size_t)arity);
            uint32_t depth = lg - ar[l];
            const uint64_t *pp = qn(r, 4 * (size_t)(depth - cfg->cap_height));
            if (r->bad) { rc = 250; break; }
            if (!canon_n(ev, 2 * (size_t)arity)) { rc = 251; break; }
            if (!(ev[2 * within] == old_eval.a && ev[2 * within + 1] == old_eval.b)) { rc = 252; break; }
            if (orc_merkle_verify(ev, 2 * arity, (uint32_t)coset, pp, depth - cfg->cap_height, layer_caps[l])) { rc = 253; break; }
            gl2 xs[64];
            uint64_t w_cur = gl_root(lg), shift_l = gl_pow(GL_GEN, (uint64_t)1 << (lgL - lg));
            for (uint32_t s = 0; s < arity; s++) xs[s] = gl2_from(gl_mul(shift_l, gl_pow(w_cur, brev((uint32_t)(coset * arity + s), lg))));
            gl2 res = {0, 0};
            for (uint32_t s = 0; s < arity; s++) {
                gl2 numr = {1, 0}, denr = {1, 0};
                for (uint32_t u = 0; u < arity; u++) if (u != s) { numr = gl2_mul(numr, gl2_sub(betas[l], xs[u])); denr = gl2_mul(denr, gl2_sub(xs[s], xs[u])); }
                res = gl2_add(res, gl2_mul((gl2){ev[2 * s], ev[2 * s + 1]}, gl2_mul(numr, gl2_inv(denr))));
            }
            old_eval = res;
            x = coset; lg = depth;
        }
        if (rc) break;
        uint64_t shift_f = gl_pow(GL_GEN, (uint64_t)1 << (lgL - lg));
        uint64_t xf = gl_mul(shift_f, gl_pow(gl_root(lg), brev((uint32_t)x, lg)));
        gl2 fe = {0, 0};
        for (size_t i = flen; i-- > 0;) fe = gl2_add(gl2_scale(fe, xf), (gl2){final_poly[2 * i], final_poly[2 * i + 1]});
        if (!gl2_eq(fe, old_eval)) { rc = 260; break; }
    }
    free(qs);
    return rc;
}

/* plonk::verifier: needs only the circuit's verifier data (constants_sigmas cap and digest from st, the gate program, k_is) */
int32_t orc_verify_plonk(const zk_stark_config *cfg, const orc_plonk_state *st, const uint64_t *pub, uint32_t n_pub, const uint64_t *proof, size_t len) {
    const zk_plonk_circuit *c = &st->c;
    const uint32_t nch = cfg->num_challenges, qdf = c->quotient_degree_factor, R = c->num_routed_wires, W = c->num_wires, NC = c->num_constants;
    const uint32_t k = c->degree_bits, nchunks = R / qdf, npp = nchunks - 1, capw = 4u << cfg->cap_height, NZ = nch * (1 + npp);
    const size_t n = (size_t)1 << k;
    prd r = {proof, len, 0, 0};
    if (q1(&r) != 0x314B4C504C474B5AULL) return 200;
    if (q1(&r) != k || q1(&r) != W || q1(&r) != R || q1(&r) != NC) return 201;
    const uint64_t *wcap = qn(&r, capw), *zcap = qn(&r, capw), *qcap = qn(&r, capw);
    if (r.bad || !canon_n(wcap, capw) || !canon_n(zcap, capw) || !canon_n(qcap, capw)) return 202;
    orc_challenger ch; orc_challenger_init(&ch);
    uint64_t pih[4]; orc_hash_no_pad(pub, n_pub, pih);
    orc_challenger_observe(&ch, st->digest, 4);
    orc_challenger_observe(&ch, pih, 4);
    orc_challenger_observe(&ch, wcap, capw);
    uint64_t betas[4], gammas[4], alphas[4];
    orc_challenger_get_n(&ch, nch, betas);
    orc_challenger_get_n(&ch, nch, gammas);
    orc_challenger_observe(&ch, zcap, capw);
    orc_challenger_get_n(&ch, nch, alphas);
    orc_challenger_observe(&ch, qcap, capw);
    uint64_t zt[2]; orc_challenger_get_n(&ch, 2, zt);
    gl2 zeta = {zt[0], zt[1]};
    const size_t n_cs = NC + R;
    const uint64_t *o_cs = qn(&r, 2 * n_cs), *o_w = qn(&r, 2 * (size_t)W), *o_z = qn(&r, 2 * (size_t)nch), *o_zn = qn(&r, 2 * (size_t)nch);
    const uint64_t *o_pp = qn(&r, 2 * (size_t)nch * npp), *o_q = qn(&r, 2 * (size_t)nch * qdf);
    if (r.bad) return 203;
    if (!canon_n(o_cs, 2 * n_cs) || !canon_n(o_w, 2 * (size_t)W) || !canon_n(o_z, 2 * nch) || !canon_n(o_zn, 2 * nch) || !canon_n(o_pp, 2 * (size_t)nch * npp) ||
        !canon_n(o_q, 2 * (size_t)nch * qdf)) return 204;
    orc_challenger_observe(&ch, o_cs, (uint32_t)(2 * n_cs)); orc_challenger_observe(&ch, o_w, 2 * W); orc_challenger_observe(&ch, o_z, 2 * nch);
    orc_challenger_observe(&ch, o_pp, 2 * nch * npp); orc_challenger_observe(&ch, o_q, 2 * nch * qdf);
    orc_challenger_observe(&ch, o_zn, 2 * nch);
    /* ---- vanishing(zeta) == Z_H(zeta) * sum_i zeta^(n i) quotient_chunk_i(zeta), per challenge */
    int32_t rc = 0;
    {
        gl2 one = {1, 0}, zn_ = gl2_pow(zeta, n), zh = gl2_sub(zn_, one);
        if (gl2_eq(zh, gl2_from(0))) return 205;
        gl2 l0 = gl2_mul(gl2_scale(zh, gl_inv(n % GL_P)), gl2_inv(gl2_sub(zeta, one)));
        const uint32_t NG = n_gate_terms(c->gates), NT = nch + nch * nchunks + NG;
        gl2 *terms = (gl2 *)malloc((NT + 1) * sizeof(gl2)), *regs = (gl2 *)malloc((c->gates->n_regs + 1) * sizeof(gl2));
        gl2 *wv = (gl2 *)malloc(W * sizeof(gl2)), *cv = (gl2 *)malloc((n_cs + 1) * sizeof(gl2));
        for (uint32_t i = 0; i < W; i++) wv[i] = (gl2){o_w[2 * i], o_w[2 * i + 1]};
        for (uint32_t i = 0; i < n_cs; i++) cv[i] = (gl2){o_cs[2 * i], o_cs[2 * i + 1]};
        uint32_t t = 0;
        for (uint32_t ci = 0; ci < nch; ci++) terms[t++] = gl2_mul(l0, gl2_sub((gl2){o_z[2 * ci], o_z[2 * ci + 1]}, one));
        uint64_t kj = 1;
        uint64_t *kis = (uint64_t *)malloc(R * sizeof(uint64_t));
        for (uint32_t j = 0; j < R; j++) { kis[j] = kj; kj = gl_mul(kj, GL_GEN); }
        for (uint32_t ci = 0; ci < nch; ci++)
            for (uint32_t cidx = 0; cidx < nchunks; cidx++) {
                gl2 num = one, den = one;
                for (uint32_t u = 0; u < qdf; u++) {
                    uint32_t j = cidx * qdf + u;
                    num = gl2_mul(num, gl2_add(gl2_add(wv[j], gl2_scale(zeta, gl_mul(betas[ci], kis[j]))), gl2_from(gammas[ci])));
                    den = gl2_mul(den, gl2_add(gl2_add(wv[j], gl2_scale(cv[NC + j], betas[ci])), gl2_from(gammas[ci])));
                }
                gl2 prev = cidx == 0 ? (gl2){o_z[2 * ci], o_z[2 * ci + 1]} : (gl2){o_pp[2 * (ci * npp + cidx - 1)], o_pp[2 * (ci * npp + cidx - 1) + 1]};
                gl2 next = cidx == nchunks - 1 ? (gl2){o_zn[2 * ci], o_zn[2 * ci + 1]} : (gl2){o_pp[2 * (ci * npp + cidx)], o_pp[2 * (ci * npp + cidx) + 1]};
                terms[t++] = gl2_sub(gl2_mul(prev, num), gl2_mul(next, den));
            }
        free(kis);
        t += gates_ext(c->gates, wv, cv, W, pih, regs, terms + t);
        for (uint32_t j = 0; j < nch; j++) {
            gl2 acc = {0, 0};
            for (uint32_t u = t; u-- > 0;) acc = gl2_add(gl2_scale(acc, alphas[j]), terms[u]);
            gl2 s = {0, 0};
            for (uint32_t q = qdf; q-- > 0;) s = gl2_add(gl2_mul(s, zn_), (gl2){o_q[2 * (j * qdf + q)], o_q[2 * (j * qdf + q) + 1]});
            if (!gl2_eq(acc, gl2_mul(zh, s))) rc = 220;
        }
        free(terms); free(regs); free(wv); free(cv);
        if (rc) return rc;
    }
    /* ---- FRI over the four oracles */
    {
        const uint64_t *caps[4] = {orc_batch_cap(st->cs), wcap, zcap, qcap};
        uint32_t ncols[4] = {(uint32_t)n_cs, W, NZ, nch * qdf};
        uint32_t cnt[2] = {(uint32_t)n_cs + W + NZ + nch * qdf, nch};
        zk_fri_poly *p0 = (zk_fri_poly *)malloc(cnt[0] * sizeof(zk_fri_poly)), *p1_ = (zk_fri_poly *)malloc(nch * sizeof(zk_fri_poly));
        gl2 *op0 = (gl2 *)malloc(cnt[0] * sizeof(gl2)), *op1 = (gl2 *)malloc(nch * sizeof(gl2));
        uint32_t q = 0;
        for (uint32_t i = 0; i < n_cs; i++) { p0[q] = (zk_fri_poly){0, i}; op0[q++] = (gl2){o_cs[2 * i], o_cs[2 * i + 1]}; }
        for (uint32_t i = 0; i < W; i++) { p0[q] = (zk_fri_poly){1, i}; op0[q++] = (gl2){o_w[2 * i], o_w[2 * i + 1]}; }
        for (uint32_t i = 0; i < nch; i++) { p0[q] = (zk_fri_poly){2, i}; op0[q++] = (gl2){o_z[2 * i], o_z[2 * i + 1]}; }
        for (uint32_t i = 0; i < nch * npp; i++) { p0[q] = (zk_fri_poly){2, nch + i}; op0[q++] = (gl2){o_pp[2 * i], o_pp[2 * i + 1]}; }
        for (uint32_t i = 0; i < nch * qdf; i++) { p0[q] = (zk_fri_poly){3, i}; op0[q++] = (gl2){o_q[2 * i], o_q[2 * i + 1]}; }
        for (uint32_t i = 0; i < nch; i++) { p1_[i] = (zk_fri_poly){2, i}; op1[i] = (gl2){o_zn[2 * i], o_zn[2 * i + 1]}; }
        gl2 points[2] = {zeta, gl2_scale(zeta, gl_root(k))};
        const zk_fri_poly *pl[2] = {p0, p1_};
        const gl2 *ops[2] = {op0, op1};
        rc = fri_verify_instance(cfg, k, 4, caps, ncols, 2, points, cnt, pl, ops, &ch, &r);
        free(p0); free(p1_); free(op0); free(op1);
    }
    if (!rc && r.pos != len) rc = 270;
    return rc;
}
