/* ORACLE -- test infrastructure, not product code (see oracle.h for who may call this).
 * STARK + FRI verifier: plain-C restatement of starky 0.4.0 `verifier::verify_stark_proof_with_challenges` and
 * plonky2 0.2.2 `fri::verifier::{verify_fri_proof, fri_combine_initial, fri_verifier_query_round}`,
 * `fri::validate_shape`, `hash::merkle_proofs::verify_merkle_proof_to_cap`.  Sources absent from /root/reference
 * (Cargo.lock:3441-3443, :4529-4531); spec = SURVEY.md A6-A9; entry from the reference: ops/src/lib.rs:52.
 * With no upstream oracle available, "the prover's output verifies" is the strongest end-to-end evidence
 * (hard parts); it is written independently of stark.c's prover (value-domain checks,
 * extension-field constraint evaluation, barycentric folding) so the two do not share mistakes.
 */
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include "gl.h"
#include "oracle.h"

typedef struct { const uint64_t *p; size_t len, pos; int bad; } rd;
static uint64_t r1(rd *r) { if (r->pos < r->len) return r->p[r->pos++]; r->bad = 1; return 0; }
static const uint64_t *rn(rd *r, size_t n) {
    if (r->pos + n <= r->len) { const uint64_t *q = r->p + r->pos; r->pos += n; return q; }
    r->bad = 1; return NULL;
}
static int canon_all(const uint64_t *p, size_t n) { for (size_t i = 0; i < n; i++) if (p[i] >= GL_P) return 0; return 1; }

/* constraint program over the extension field (Stark::eval_ext_circuit's role on the verifier side) */
typedef struct { const gl2 *local, *next; const uint64_t *pub, *chal; gl2 z_last, l_first, l_last; } xctx;
static gl2 opnd_ext(const zk_air *air, const xctx *e, const gl2 *regs, uint32_t o) {
    uint32_t k = o >> 24, i = o & 0xFFFFFF;
    switch (k) {
    case ZK_OPND_REG: return regs[i];
    case ZK_OPND_LOCAL: return e->local[i];
    case ZK_OPND_NEXT: return e->next[i];
    case ZK_OPND_CONST: return gl2_from(air->constants[i]);
    case ZK_OPND_PUBLIC: return gl2_from(e->pub[i]);
    default: return gl2_from(e->chal[i]);
    }
}
static void eval_ext(const zk_air *air, const xctx *e, uint32_t nch, const uint64_t *alphas, gl2 *acc) {
    gl2 *regs = (gl2 *)calloc(air->n_regs + 1, sizeof(gl2));
    for (uint32_t j = 0; j < nch; j++) acc[j] = gl2_from(0);
    for (uint32_t p = 0; p < air->n_instrs; p++) {
        const zk_instr *in = &air->instrs[p];
        gl2 a = opnd_ext(air, e, regs, in->a);
        if (in->op <= ZK_OP_MUL) {
            gl2 b = opnd_ext(air, e, regs, in->b);
            regs[in->dst & 0xFFFFFF] = in->op == ZK_OP_ADD ? gl2_add(a, b) : in->op == ZK_OP_SUB ? gl2_sub(a, b) : gl2_mul(a, b);
        } else {
            gl2 c = in->op == ZK_OP_EMIT ? a : gl2_mul(a, in->op == ZK_OP_EMIT_TRANSITION ? e->z_last : in->op == ZK_OP_EMIT_FIRST ? e->l_first : e->l_last);
            for (uint32_t j = 0; j < nch; j++) acc[j] = gl2_add(gl2_scale(acc[j], alphas[j]), c);
        }
    }
    free(regs);
}

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

#define FAIL(code) do { rc = (code); goto done; } while (0)

int32_t orc_verify(const zk_stark_config *cfg, uint32_t nt, const orc_table *tables, const uint64_t *proof, size_t len) {
    rd r = {proof, len, 0, 0};
    int32_t rc = 0;
    uint32_t nch = cfg->num_challenges, capw = 4u << cfg->cap_height;
    if (r1(&r) != 0x31465250474C4B5AULL) return 100;
    if (r1(&r) != nt) return 101;
    orc_challenger ch; orc_challenger_init(&ch);
    /* The trace caps of all tables are observed first, so locate them with a pre-pass over the proof. */
    size_t *tbl_pos = (size_t *)calloc(nt + 1, sizeof(size_t));
    {
        rd s = r;
        for (uint32_t t = 0; t < nt; t++) {
            tbl_pos[t] = s.pos;
            uint32_t log_n = (uint32_t)r1(&s), C = (uint32_t)r1(&s), A = (uint32_t)r1(&s);
            if (s.bad || log_n != tables[t].log_n || C != tables[t].air->n_trace_cols || A != zk_air_n_aux(tables[t].air, nch)) { free(tbl_pos); return 102; }
            const uint64_t *cap = rn(&s, capw);
            if (!cap || !canon_all(cap, capw)) { free(tbl_pos); return 103; }
            orc_challenger_observe(&ch, cap, capw);
            orc_table one = tables[t];
            size_t sz = orc_proof_size_words(cfg, 1, &one) - 2;   /* per-table payload */
            s.pos = tbl_pos[t] + sz;
        }
        if (s.pos != len) { free(tbl_pos); return 104; }
    }
    for (uint32_t t = 0; t < nt; t++) orc_challenger_observe(&ch, tables[t].public_inputs, tables[t].air->n_public_inputs);
    uint64_t chal[16]; orc_challenger_get_n(&ch, 2 * nch, chal);

    for (uint32_t t = 0; t < nt; t++) {
        const zk_air *air = tables[t].air;
        r.pos = tbl_pos[t];
        uint32_t log_n = (uint32_t)r1(&r), C = (uint32_t)r1(&r), A = (uint32_t)r1(&r), Q = nch << cfg->rate_bits;
        uint32_t lgL = log_n + cfg->rate_bits;
        size_t n = (size_t)1 << log_n, L = (size_t)1 << lgL;
        uint64_t st12[12]; orc_challenger_compact(&ch, st12);
        const uint64_t *trace_cap = rn(&r, capw), *aux_cap = NULL;
        if (A) { aux_cap = rn(&r, capw); if (!aux_cap || !canon_all(aux_cap, capw)) FAIL(110); orc_challenger_observe(&ch, aux_cap, capw); }
        uint64_t alphas[16]; orc_challenger_get_n(&ch, nch, alphas);
        const uint64_t *quot_cap = rn(&r, capw);
        if (!quot_cap || !canon_all(quot_cap, capw)) FAIL(111);
        orc_challenger_observe(&ch, quot_cap, capw);
        uint64_t zb[2]; orc_challenger_get_n(&ch, 2, zb);
        gl2 zeta = {zb[0], zb[1]};
        const uint64_t *o_local = rn(&r, 2 * C), *o_next = rn(&r, 2 * C), *o_aux = rn(&r, 2 * A), *o_auxn = rn(&r, 2 * A);
        const uint64_t *zs_first = A ? rn(&r, nch) : NULL;
        const uint64_t *o_q = rn(&r, 2 * Q);
        if (r.bad) FAIL(112);
        if (!canon_all(o_local, 2 * C) || !canon_all(o_next, 2 * C) || !canon_all(o_aux, 2 * A) || !canon_all(o_auxn, 2 * A) || !canon_all(o_q, 2 * Q) || (A && !canon_all(zs_first, nch))) FAIL(113);
        orc_challenger_observe(&ch, o_local, 2 * C); orc_challenger_observe(&ch, o_aux, 2 * A); orc_challenger_observe(&ch, o_q, 2 * Q);
        orc_challenger_observe(&ch, o_next, 2 * C); orc_challenger_observe(&ch, o_auxn, 2 * A);
        for (uint32_t j = 0; j < nch && A; j++) { uint64_t e[2] = {zs_first[j], 0}; orc_challenger_observe(&ch, e, 2); }

        /* ---- constraint check at zeta:  vanishing(zeta) == Z_H(zeta) * sum_i zeta^(n i) q_i(zeta) */
        {
            gl2 *loc = (gl2 *)malloc((C + A + 1) * sizeof(gl2)), *nxt = (gl2 *)malloc((C + A + 1) * sizeof(gl2));
            for (uint32_t c = 0; c < C; c++) { loc[c] = (gl2){o_local[2 * c], o_local[2 * c + 1]}; nxt[c] = (gl2){o_next[2 * c], o_next[2 * c + 1]}; }
            for (uint32_t c = 0; c < A; c++) { loc[C + c] = (gl2){o_aux[2 * c], o_aux[2 * c + 1]}; nxt[C + c] = (gl2){o_auxn[2 * c], o_auxn[2 * c + 1]}; }
            gl2 zn = gl2_pow(zeta, n), one = {1, 0};
            gl2 zh = gl2_sub(zn, one);
            if (gl2_eq(zh, gl2_from(0))) { free(loc); free(nxt); FAIL(114); }
            uint64_t g_n = gl_root(log_n), ninv = gl_inv(n % GL_P);
            /* eval_l_0_and_l_last: L_0(x) = (x^n-1)/(n(x-1)), L_last(x) = (x^n-1)/(n(g x - 1)) */
            gl2 l0 = gl2_mul(gl2_scale(zh, ninv), gl2_inv(gl2_sub(zeta, one)));
            gl2 ll = gl2_mul(gl2_scale(zh, ninv), gl2_inv(gl2_sub(gl2_scale(zeta, g_n), one)));
            xctx e = {loc, nxt, tables[t].public_inputs, chal, gl2_sub(zeta, gl2_from(gl_inv(g_n))), l0, ll};
            gl2 acc[16];
            eval_ext(air, &e, nch, alphas, acc);
            uint32_t qdf = 1u << cfg->rate_bits;
            int ok = 1;
            for (uint32_t j = 0; j < nch; j++) {
                gl2 s = {0, 0};
                for (uint32_t k = qdf; k-- > 0;) s = gl2_add(gl2_mul(s, zn), (gl2){o_q[2 * (j * qdf + k)], o_q[2 * (j * qdf + k) + 1]});
                if (!gl2_eq(acc[j], gl2_mul(zh, s))) ok = 0;
            }
            /* the Z columns must open at row 0 to the claimed ctl_zs_first; starky checks the cross-table product,
               here a single table has no partner so we only bind the value into the transcript (already observed) */
            free(loc); free(nxt);
            if (!ok) FAIL(120);
        }

        /* ---- FRI */
        uint64_t ab[2]; orc_challenger_get_n(&ch, 2, ab);
        gl2 alpha = {ab[0], ab[1]};
        uint32_t ar[64], nl;
        if (!fri_arities_checked(cfg, log_n, ar, &nl)) FAIL(129);      /* a (config, size) pair upstream panics on: no valid proof exists */
        if (r1(&r) != nl) FAIL(130);
        const uint64_t *layer_caps[64]; gl2 betas[64];
        for (uint32_t l = 0; l < nl; l++) {
            layer_caps[l] = rn(&r, capw);
            if (!layer_caps[l] || !canon_all(layer_caps[l], capw)) FAIL(131);
            orc_challenger_observe(&ch, layer_caps[l], capw);
            uint64_t bb[2]; orc_challenger_get_n(&ch, 2, bb);
            betas[l] = (gl2){bb[0], bb[1]};
        }
        uint32_t dfin = log_n; for (uint32_t l = 0; l < nl; l++) dfin -= ar[l];
        size_t flen = (size_t)r1(&r);
        if (flen != ((size_t)1 << dfin)) FAIL(132);
        const uint64_t *final_poly = rn(&r, 2 * flen);
        if (!final_poly || !canon_all(final_poly, 2 * flen)) FAIL(133);
        orc_challenger_observe(&ch, final_poly, (uint32_t)(2 * flen));
        uint64_t witness = r1(&r);
        if (witness >= GL_P) FAIL(134);
        orc_challenger_observe(&ch, &witness, 1);
        uint64_t resp = orc_challenger_get(&ch);
        if ((resp ? (uint32_t)__builtin_clzll(resp) : 64) < cfg->proof_of_work_bits) FAIL(135);

        /* PrecomputedReducedOpenings::from_os_and_alpha: reduce each batch's openings with alpha (Horner from the last) */
        uint32_t nb = A ? 3 : 2;
        gl2 red[3];
        for (uint32_t b = 0; b < nb; b++) {
            gl2 acc = {0, 0};
            if (b == 0) {
                for (uint32_t k = Q; k-- > 0;) acc = gl2_add(gl2_mul(acc, alpha), (gl2){o_q[2 * k], o_q[2 * k + 1]});
                for (uint32_t k = A; k-- > 0;) acc = gl2_add(gl2_mul(acc, alpha), (gl2){o_aux[2 * k], o_aux[2 * k + 1]});
                for (uint32_t k = C; k-- > 0;) acc = gl2_add(gl2_mul(acc, alpha), (gl2){o_local[2 * k], o_local[2 * k + 1]});
            } else if (b == 1) {
                for (uint32_t k = A; k-- > 0;) acc = gl2_add(gl2_mul(acc, alpha), (gl2){o_auxn[2 * k], o_auxn[2 * k + 1]});
                for (uint32_t k = C; k-- > 0;) acc = gl2_add(gl2_mul(acc, alpha), (gl2){o_next[2 * k], o_next[2 * k + 1]});
            } else {
                for (uint32_t k = nch; k-- > 0;) acc = gl2_add(gl2_mul(acc, alpha), gl2_from(zs_first[k]));
            }
            red[b] = acc;
        }
        uint64_t g_n = gl_root(log_n);
        gl2 points[3] = {zeta, gl2_scale(zeta, g_n), {1, 0}};
        uint32_t cnts[3] = {C + A + Q, C + A, nch};

        uint32_t nq = cfg->num_query_rounds;
        uint64_t *qs = (uint64_t *)malloc(nq * sizeof(uint64_t));
        orc_challenger_get_n(&ch, nq, qs);
        uint32_t npath = lgL - cfg->cap_height;
        uint64_t w_L = gl_root(lgL);
        for (uint32_t q = 0; q < nq && !rc; q++) {
            size_t x = (size_t)(qs[q] % L);
            const uint64_t *lt = rn(&r, C), *pt = rn(&r, 4 * (size_t)npath);
            const uint64_t *la = NULL, *pa = NULL;
            if (A) { la = rn(&r, A); pa = rn(&r, 4 * (size_t)npath); }
            const uint64_t *lq = rn(&r, Q), *pq = rn(&r, 4 * (size_t)npath);
            if (r.bad) { rc = 140; break; }
            if (!canon_all(lt, C) || (A && !canon_all(la, A)) || !canon_all(lq, Q)) { rc = 141; break; }
            if (orc_merkle_verify(lt, C, (uint32_t)x, pt, npath, trace_cap)) { rc = 142; break; }
            if (A && orc_merkle_verify(la, A, (uint32_t)x, pa, npath, aux_cap)) { rc = 143; break; }
            if (orc_merkle_verify(lq, Q, (uint32_t)x, pq, npath, quot_cap)) { rc = 144; break; }
            /* fri_combine_initial */
            uint64_t sx = gl_mul(GL_GEN, gl_pow(w_L, bitrev32((uint32_t)x, lgL)));
            gl2 sum = {0, 0};
            for (uint32_t b = 0; b < nb; b++) {
                gl2 acc = {0, 0};
                if (b == 0) {
                    for (uint32_t k = Q; k-- > 0;) { acc = gl2_mul(acc, alpha); acc.a = gl_add(acc.a, lq[k]); }
                    for (uint32_t k = A; k-- > 0;) { acc = gl2_mul(acc, alpha); acc.a = gl_add(acc.a, la[k]); }
                    for (uint32_t k = C; k-- > 0;) { acc = gl2_mul(acc, alpha); acc.a = gl_add(acc.a, lt[k]); }
                } else if (b == 1) {
                    for (uint32_t k = A; k-- > 0;) { acc = gl2_mul(acc, alpha); acc.a = gl_add(acc.a, la[k]); }
                    for (uint32_t k = C; k-- > 0;) { acc = gl2_mul(acc, alpha); acc.a = gl_add(acc.a, lt[k]); }
                } else {
                    for (uint32_t k = nch; k-- > 0;) { acc = gl2_mul(acc, alpha); acc.a = gl_add(acc.a, la[A - nch + k]); }
                }
                gl2 num = gl2_sub(acc, red[b]);
                gl2 den = gl2_sub(gl2_from(sx), points[b]);
                sum = gl2_mul(sum, gl2_pow(alpha, cnts[b]));
                sum = gl2_add(sum, gl2_mul(num, gl2_inv(den)));
            }
            gl2 old_eval = sum;
            uint32_t lg = lgL;
            for (uint32_t l = 0; l < nl; l++) {
                uint32_t arity = 1u << ar[l];
                size_t coset = x >> ar[l], within = x & (arity - 1);
                const uint64_t *ev = rn(&r, 2 * (size_t)arity);
                uint32_t depth = lg - ar[l];
                const uint64_t *pp = rn(&r, 4 * (size_t)(depth - cfg->cap_height));
                if (r.bad) { rc = 150; break; }
                if (!canon_all(ev, 2 * (size_t)arity)) { rc = 151; break; }
                if (!(ev[2 * within] == old_eval.a && ev[2 * within + 1] == old_eval.b)) { rc = 152; break; }
                if (orc_merkle_verify(ev, 2 * arity, (uint32_t)coset, pp, depth - cfg->cap_height, layer_caps[l])) { rc = 153; break; }
                /* compute_evaluation: interpolate {(x_s, P(x_s))} over the 16-point coset and evaluate at beta.
                   leaf slot s holds the value at domain index  bitrev_lg(coset*arity + s)  of this layer's domain. */
                gl2 xs[64];
                uint64_t w_cur = gl_root(lg), shift_l = gl_pow(GL_GEN, (uint64_t)1 << (lgL - lg));
                for (uint32_t s = 0; s < arity; s++) xs[s] = gl2_from(gl_mul(shift_l, gl_pow(w_cur, bitrev32((uint32_t)(coset * arity + s), lg))));
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
            /* final_poly.eval(subgroup_x) at the fully folded point */
            uint64_t shift_f = gl_pow(GL_GEN, (uint64_t)1 << (lgL - lg));
            uint64_t xf = gl_mul(shift_f, gl_pow(gl_root(lg), bitrev32((uint32_t)x, lg)));
            gl2 fe = {0, 0};
            for (size_t i = flen; i-- > 0;) fe = gl2_add(gl2_scale(fe, xf), (gl2){final_poly[2 * i], final_poly[2 * i + 1]});
            if (!gl2_eq(fe, old_eval)) { rc = 160; break; }
        }
        free(qs);
        if (rc) goto done;
        (void)st12;
    }
    if (r.pos != len) rc = 170;
done:
    free(tbl_pos);
    return rc;
}
