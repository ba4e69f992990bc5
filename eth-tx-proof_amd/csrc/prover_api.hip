// Finer-grained C-ABI entry points at the starky / plonky2 trait surface: the pieces zk_prove_tables is
// made of, callable on their own so that a Rust -sys crate can replace upstream items one at a time.
//   zk_aux_columns          starky::lookup::lookup_helper_columns + cross_table_lookup::partial_sums
//   zk_quotient             starky::prover::compute_quotient_polys
//   zk_pow_grind            plonky2::fri::prover::fri_proof_of_work (the search)
//   zk_fri_prove_openings   plonky2::fri::oracle::PolynomialBatch::prove_openings -> fri_proof
// (crates pinned at /root/reference/Cargo.lock:4529-4531 and :3441-3443; reached from /root/reference/ops/src/lib.rs:52)
#include "zk_internal.h"
#include "prover.h"

static u32 n_aux_of(const zk_air *air, u32 nch) { return zk_air_n_aux(air, nch); }

extern "C" {

int32_t zk_air_validate(const zk_air *air, uint32_t nch) {
    if (!air || nch == 0 || nch > 4) return ZK_ERR_INVALID_ARG;
    if (air->n_trace_cols == 0 || air->n_trace_cols > (1u << 20) || air->n_helpers > (1u << 16) || air->n_regs > (1u << 16) ||
        air->n_instrs > (1u << 24) || air->n_constants > (1u << 24) || air->n_public_inputs > (1u << 24))
        return ZK_ERR_INVALID_ARG;
    if ((air->n_instrs && !air->instrs) || (air->n_constants && !air->constants) || (air->n_helpers && !air->helpers)) return ZK_ERR_INVALID_ARG;
    if (air->lookup) {
        // a lookup argument rides in front of the CTL-shaped columns and needs them present (every zkEVM table with lookups has
        // CTLs too; the Z(1) opening batch and the proof layout key on the CTL columns)
        const zk_lookup *lk = air->lookup;
        if (lk->n_cols == 0 || lk->n_cols > (1u << 16) || !lk->cols || air->n_helpers == 0) return ZK_ERR_INVALID_ARG;
        if (lk->table_col >= air->n_trace_cols || lk->freq_col >= air->n_trace_cols) return ZK_ERR_INVALID_ARG;
        for (u32 i = 0; i < lk->n_cols; i++)
            if (lk->cols[i].col >= air->n_trace_cols || (lk->cols[i].filter != ZK_NO_FILTER && lk->cols[i].filter >= air->n_trace_cols)) return ZK_ERR_INVALID_ARG;
    }
    const u32 C = air->n_trace_cols, A = n_aux_of(air, nch);
    for (u32 m = 0; m < air->n_helpers; m++) {
        const zk_helper_desc &h = air->helpers[m];
        if (h.a0 >= C || h.b0 >= C || h.a1 >= C || h.b1 >= C) return ZK_ERR_INVALID_ARG;
        if (air->helpers_ext) {
            const zk_helper_ext &x = air->helpers_ext[m];
            for (int k = 0; k < 2; k++) {
                if (x.n_extra[k] > ZK_CTL_MAX_EXTRA || (x.filter[k] != ZK_NO_FILTER && x.filter[k] >= C)) return ZK_ERR_INVALID_ARG;
                for (u32 i = 0; i < x.n_extra[k]; i++) if (x.extra[k][i] >= C) return ZK_ERR_INVALID_ARG;
            }
        }
    }
    auto src_ok = [&](u32 o) {
        const u32 kind = o >> 24, idx = o & 0xFFFFFFu;
        switch (kind) {
        case ZK_OPND_REG: return idx < air->n_regs;
        case ZK_OPND_LOCAL: case ZK_OPND_NEXT: return idx < C + A;
        case ZK_OPND_CONST: return idx < air->n_constants;
        case ZK_OPND_PUBLIC: return idx < air->n_public_inputs;
        case ZK_OPND_CHALLENGE: return idx < 2 * nch;
        default: return false;
        }
    };
    for (u32 i = 0; i < air->n_instrs; i++) {
        const zk_instr &in = air->instrs[i];
        if (in.op > ZK_OP_EMIT_LAST) return ZK_ERR_INVALID_ARG;
        if (!src_ok(in.a)) return ZK_ERR_INVALID_ARG;
        if (in.op <= ZK_OP_MUL) {                                  // dst = a (op) b; dst is a plain register number
            if (!src_ok(in.b)) return ZK_ERR_INVALID_ARG;
            if ((in.dst >> 24) != ZK_OPND_REG || (in.dst & 0xFFFFFFu) >= air->n_regs) return ZK_ERR_INVALID_ARG;
        }
    }
    return ZK_OK;
}

int32_t zk_aux_columns(zk_ctx *ctx, const zk_air *air, uint32_t num_challenges, const uint64_t *challenges,
                       const uint64_t *trace, uint32_t log_n, uint64_t *aux_out) {
    if (!ctx || !air || !challenges || !trace || !aux_out || num_challenges == 0 || num_challenges > 4 || log_n > 26) return ZK_ERR_INVALID_ARG;
    if (zk_air_validate(air, num_challenges) != ZK_OK) return ctx->fail(ZK_ERR_INVALID_ARG, "aux_columns: malformed AIR");
    hipSetDevice(ctx->device);
    const u32 A = n_aux_of(air, num_challenges);
    if (!A) return ZK_OK;
    const size_t n = (size_t)1 << log_n;
    u64 *d_tr = (u64 *)ctx->alloc(n * air->n_trace_cols * sizeof(u64)), *d_aux = (u64 *)ctx->alloc(n * A * sizeof(u64));
    u64 *d_ch = (u64 *)ctx->alloc(2 * num_challenges * sizeof(u64));
    zk_helper_desc *d_h = (zk_helper_desc *)ctx->alloc(air->n_helpers * sizeof(zk_helper_desc));
    zk_helper_ext *d_x = (air->helpers_ext && air->n_helpers) ? (zk_helper_ext *)ctx->alloc(air->n_helpers * sizeof(zk_helper_ext)) : nullptr;
    int32_t rc = (d_tr && d_aux && d_ch && d_h && (d_x || !(air->helpers_ext && air->n_helpers))) ? ZK_OK : ZK_ERR_OOM;
    std::vector<u64> ch(challenges, challenges + 2 * num_challenges);
    for (auto &v : ch) v = gl_canon(v);
    if (rc == ZK_OK) rc = zk_dev_upload(ctx, d_tr, trace, n * air->n_trace_cols * sizeof(u64));
    if (rc == ZK_OK) rc = zk_dev_upload(ctx, d_ch, ch.data(), ch.size() * sizeof(u64));
    if (rc == ZK_OK) rc = zk_dev_upload(ctx, d_h, air->helpers, air->n_helpers * sizeof(zk_helper_desc));
    if (rc == ZK_OK && d_x) rc = zk_dev_upload(ctx, d_x, air->helpers_ext, air->n_helpers * sizeof(zk_helper_ext));
    // the trace is used as given by the helper kernel: canonicalise it through a commit-free pass
    if (rc == ZK_OK) rc = zk_aux_columns_dev(ctx, air, d_h, d_x, num_challenges, d_ch, d_tr, log_n, d_aux);
    if (rc == ZK_OK) rc = zk_dev_download(ctx, aux_out, d_aux, n * A * sizeof(u64));
    u64 err = 0;
    if (rc == ZK_OK) rc = zk_dev_download(ctx, &err, ctx->d_err, sizeof(u64));
    if (rc == ZK_OK && err) { rc = (int32_t)(uint32_t)err; hipMemsetAsync(ctx->d_err, 0, sizeof(u64), ctx->stream); }
    ctx->release(d_tr); ctx->release(d_aux); ctx->release(d_ch); ctx->release(d_h); ctx->release(d_x);
    return rc;
}

int32_t zk_quotient(zk_ctx *ctx, const zk_stark_config *cfg, const zk_air *air, zk_batch *trace, zk_batch *aux,
                    const uint64_t *public_inputs, const uint64_t *challenges, const uint64_t *alphas, uint64_t *quotient_coeffs_out) {
    if (!ctx || !cfg || !air || !trace || !alphas || !quotient_coeffs_out) return ZK_ERR_INVALID_ARG;
    if (trace->n_cols != air->n_trace_cols || (aux && aux->log_n != trace->log_n)) return ZK_ERR_INVALID_ARG;
    if (zk_air_validate(air, cfg->num_challenges) != ZK_OK) return ctx->fail(ZK_ERR_INVALID_ARG, "quotient: malformed AIR");
    {   // the auxiliary batch must hold exactly the columns the program may address
        const u32 A = n_aux_of(air, cfg->num_challenges);
        if ((A != 0) != (aux != nullptr) || (aux && aux->n_cols != A)) return ctx->fail(ZK_ERR_INVALID_ARG, "quotient: auxiliary batch does not match the AIR");
        if (A && !challenges) return ctx->fail(ZK_ERR_INVALID_ARG, "quotient: challenges missing");
        if (air->n_public_inputs && !public_inputs) return ctx->fail(ZK_ERR_INVALID_ARG, "quotient: public inputs missing");
    }
    hipSetDevice(ctx->device);
    const u32 nch = cfg->num_challenges;
    const size_t L = trace->L();
    DevAir da; da.host = air;
    da.d_instrs = (zk_instr *)ctx->alloc((size_t)(air->n_instrs ? air->n_instrs : 1) * sizeof(zk_instr));
    da.d_constants = (u64 *)ctx->alloc((size_t)(air->n_constants ? air->n_constants : 1) * sizeof(u64));
    da.d_helpers = nullptr; da.d_helpers_ext = nullptr;
    u64 *d_pub = (u64 *)ctx->alloc((size_t)(air->n_public_inputs ? air->n_public_inputs : 1) * sizeof(u64));
    u64 *d_ch = (u64 *)ctx->alloc(16 * sizeof(u64)), *d_al = (u64 *)ctx->alloc(8 * sizeof(u64));
    u64 *d_q = (u64 *)ctx->alloc((size_t)nch * L * sizeof(u64));
    int32_t rc = (da.d_instrs && da.d_constants && d_pub && d_ch && d_al && d_q) ? ZK_OK : ZK_ERR_OOM;
    auto up = [&](void *d, const uint64_t *src, size_t cnt) {
        if (rc != ZK_OK || !cnt) return;
        std::vector<u64> c(src, src + cnt);
        for (auto &v : c) v = gl_canon(v);
        rc = zk_dev_upload(ctx, d, c.data(), cnt * sizeof(u64));
    };
    if (rc == ZK_OK && air->n_instrs) rc = zk_dev_upload(ctx, da.d_instrs, air->instrs, air->n_instrs * sizeof(zk_instr));
    up(da.d_constants, air->constants, air->n_constants);
    up(d_pub, public_inputs, public_inputs ? air->n_public_inputs : 0);
    up(d_ch, challenges, challenges ? 2 * nch : 0);
    up(d_al, alphas, nch);
    if (rc == ZK_OK) rc = zk_quotient_dev(ctx, da, *cfg, trace, aux, d_pub, d_ch, d_al, d_q);
    if (rc == ZK_OK) rc = zk_dev_download(ctx, quotient_coeffs_out, d_q, (size_t)nch * L * sizeof(u64));
    ctx->release(da.d_instrs); ctx->release(da.d_constants); ctx->release(d_pub); ctx->release(d_ch); ctx->release(d_al); ctx->release(d_q);
    return rc;
}

int32_t zk_pow_grind(zk_ctx *ctx, const uint64_t state[12], uint32_t pos, uint32_t min_zeros, uint64_t *witness_out) {
    if (!ctx || !state || !witness_out || pos >= 8 || min_zeros > 34) return ZK_ERR_INVALID_ARG;
    hipSetDevice(ctx->device);
    u64 h[14];
    for (int i = 0; i < 12; i++) h[i] = gl_canon(state[i]);
    h[12] = pos; h[13] = ~0ULL;
    u64 *d = (u64 *)ctx->alloc(16 * sizeof(u64));
    if (!d) return ZK_ERR_OOM;
    int32_t rc = zk_dev_upload(ctx, d, h, sizeof h);
    if (rc == ZK_OK) rc = zk_pow_search_dev(ctx, min_zeros, d);
    if (rc == ZK_OK) rc = zk_dev_download(ctx, h, d, sizeof h);
    ctx->release(d);
    if (rc) return rc;
    if (h[13] == ~0ULL) return ZK_ERR_POW;
    *witness_out = h[13];
    return ZK_OK;
}

size_t zk_fri_proof_size_words(const zk_stark_config *cfg, uint32_t log_n, uint32_t n_trace_cols, uint32_t n_aux_cols) {
    if (!cfg || log_n + cfg->rate_bits < cfg->cap_height || !zk_fri_reduction_ok(*cfg, log_n)) return 0;
    const u32 capw = 4u << cfg->cap_height, Q = cfg->num_challenges << cfg->rate_bits, lgL = log_n + cfg->rate_bits;
    u32 ar[64], nl; zk_fri_arities(*cfg, log_n, ar, &nl);
    u32 d = log_n; for (u32 l = 0; l < nl; l++) d -= ar[l];
    size_t per_q = (size_t)(n_trace_cols + n_aux_cols + Q) + (size_t)(n_aux_cols ? 3 : 2) * 4 * (lgL - cfg->cap_height);
    u32 lg = lgL;
    for (u32 l = 0; l < nl; l++) { lg -= ar[l]; per_q += (2u << ar[l]) + 4 * (size_t)(lg - cfg->cap_height); }
    return 1 + (size_t)nl * capw + 1 + 2 * ((size_t)1 << d) + 1 + per_q * cfg->num_query_rounds;
}

int32_t zk_fri_prove_openings(zk_ctx *ctx, const zk_stark_config *cfg, zk_batch *trace, zk_batch *aux, zk_batch *quot,
                              zk_challenger *ch, const uint64_t zeta[2], uint64_t *fri_proof_out, size_t cap_words, size_t *len_words) {
    if (!ctx || !cfg || !trace || !quot || !ch || !zeta || !fri_proof_out) return ZK_ERR_INVALID_ARG;
    if (quot->log_n != trace->log_n || (aux && aux->log_n != trace->log_n) || ch->ctx != ctx) return ZK_ERR_INVALID_ARG;
    hipSetDevice(ctx->device);
    const u32 log_n = trace->log_n, nch = cfg->num_challenges, C = trace->n_cols, A = aux ? aux->n_cols : 0, Q = quot->n_cols;
    const size_t n = trace->n();
    const size_t fri_words = zk_fri_proof_size_words(cfg, log_n, C, A);
    if (len_words) *len_words = fri_words;
    if (cap_words < fri_words) return ZK_ERR_BUFFER_TOO_SMALL;
    // private proof-shaped buffer: openings first (prove_openings re-derives P(z) from them), then the FRI section
    FriLayout lay = {};
    size_t o = 0;
    lay.o_local = o; o += 2 * (size_t)C; lay.o_next = o; o += 2 * (size_t)C;
    lay.o_aux = o; o += 2 * (size_t)A; lay.o_auxn = o; o += 2 * (size_t)A;
    lay.o_zs = o; if (A) o += nch;
    lay.o_q = o; o += 2 * (size_t)Q;
    const size_t fri_base = o;
    u32 ar[64], nl; zk_fri_arities(*cfg, log_n, ar, &nl);
    const u32 capw = 4u << cfg->cap_height;
    lay.o_nl = o; o += 1; lay.o_layer_caps = o; o += (size_t)nl * capw;
    u32 d = log_n; for (u32 l = 0; l < nl; l++) d -= ar[l];
    lay.o_flen = o; o += 1; lay.o_final = o; o += 2 * ((size_t)1 << d); lay.o_witness = o; o += 1;
    lay.o_queries = o;
    lay.per_query = (fri_words - (o - fri_base)) / cfg->num_query_rounds;
    const size_t total = fri_base + fri_words;
    u64 *d_buf = (u64 *)ctx->alloc(total * sizeof(u64)), *d_zeta = (u64 *)ctx->alloc(8 * sizeof(u64)), *zs = (u64 *)ctx->alloc(2 * 8 * sizeof(u64));
    if (!d_buf || !d_zeta || !zs) return ZK_ERR_OOM;
    u64 g_n = gl_root(log_n);
    u64 hz[6] = {gl_canon(zeta[0]), gl_canon(zeta[1]), gl_mul(gl_canon(zeta[0]), g_n), gl_mul(gl_canon(zeta[1]), g_n), 1, 0};
    int32_t rc = zk_dev_upload(ctx, d_zeta, hz, sizeof hz);
    if (rc == ZK_OK) rc = zk_eval_cols_at(ctx, trace->coeffs, n, log_n, C, d_zeta, d_buf + lay.o_local);
    if (rc == ZK_OK) rc = zk_eval_cols_at(ctx, trace->coeffs, n, log_n, C, d_zeta + 2, d_buf + lay.o_next);
    if (rc == ZK_OK && A) rc = zk_eval_cols_at(ctx, aux->coeffs, n, log_n, A, d_zeta, d_buf + lay.o_aux);
    if (rc == ZK_OK && A) rc = zk_eval_cols_at(ctx, aux->coeffs, n, log_n, A, d_zeta + 2, d_buf + lay.o_auxn);
    if (rc == ZK_OK) rc = zk_eval_cols_at(ctx, quot->coeffs, n, log_n, Q, d_zeta, d_buf + lay.o_q);
    if (rc == ZK_OK && A) {
        rc = zk_eval_cols_at(ctx, aux->coeffs + (size_t)(A - nch) * n, n, log_n, nch, d_zeta + 4, zs);
        std::vector<u64> tmp(2 * nch), re(nch);
        if (rc == ZK_OK) rc = zk_dev_download(ctx, tmp.data(), zs, 2 * nch * sizeof(u64));
        for (u32 j = 0; j < nch; j++) re[j] = tmp[2 * j];
        if (rc == ZK_OK) rc = zk_dev_upload(ctx, d_buf + lay.o_zs, re.data(), nch * sizeof(u64));
    }
    if (rc == ZK_OK) rc = zk_fri_prove_dev(ctx, *cfg, ch->d, trace, aux, quot, d_zeta, d_buf, lay);
    if (rc == ZK_OK) rc = zk_dev_download(ctx, fri_proof_out, d_buf + fri_base, fri_words * sizeof(u64));
    u64 err = 0;
    if (rc == ZK_OK) rc = zk_dev_download(ctx, &err, ctx->d_err, sizeof(u64));
    if (rc == ZK_OK && err) { rc = (int32_t)(uint32_t)err; hipMemsetAsync(ctx->d_err, 0, sizeof(u64), ctx->stream); }
    ctx->release(d_buf); ctx->release(d_zeta); ctx->release(zs);
    return rc;
}

size_t zk_fri_instance_proof_size_words(const zk_stark_config *cfg, uint32_t log_n, uint32_t n_oracles, const uint32_t *oracle_n_cols) {
    if (!cfg || !oracle_n_cols || cfg->num_query_rounds == 0 || log_n + cfg->rate_bits < cfg->cap_height || !zk_fri_reduction_ok(*cfg, log_n)) return 0;
    const u32 capw = 4u << cfg->cap_height, lgL = log_n + cfg->rate_bits;
    u32 ar[64], nl; zk_fri_arities(*cfg, log_n, ar, &nl);
    u32 d = log_n; for (u32 l = 0; l < nl; l++) d -= ar[l];
    size_t per_q = 0;
    for (u32 o = 0; o < n_oracles; o++) per_q += oracle_n_cols[o] + (size_t)4 * (lgL - cfg->cap_height);
    u32 lg = lgL;
    for (u32 l = 0; l < nl; l++) { lg -= ar[l]; per_q += (2u << ar[l]) + 4 * (size_t)(lg - cfg->cap_height); }
    return 1 + (size_t)nl * capw + 1 + 2 * ((size_t)1 << d) + 1 + per_q * cfg->num_query_rounds;
}

int32_t zk_fri_prove_instance(zk_ctx *ctx, const zk_stark_config *cfg, uint32_t n_oracles, zk_batch *const *oracles, uint32_t n_batches,
                              const zk_fri_batch *batches, zk_challenger *ch, uint64_t *fri_proof_out, size_t cap_words, size_t *len_words) {
    if (!ctx || !cfg || !oracles || !batches || !ch || !fri_proof_out || ch->ctx != ctx) return ZK_ERR_INVALID_ARG;
    if (n_oracles == 0 || n_oracles > 8 || n_batches == 0 || n_batches > 4) return ctx->fail(ZK_ERR_INVALID_ARG, "fri instance: 1..8 oracles, 1..4 batches");
    if (cfg->rate_bits > 4 || cfg->cap_height > 8 || cfg->fri_arity_bits < 2 || cfg->fri_arity_bits > 6 || cfg->num_query_rounds == 0 ||
        cfg->proof_of_work_bits > 34)
        return ctx->fail(ZK_ERR_INVALID_ARG, "fri instance: unsupported FRI parameters");
    u32 ncols[8];
    for (u32 o = 0; o < n_oracles; o++) {
        if (!oracles[o] || oracles[o]->ctx != ctx) return ctx->fail(ZK_ERR_INVALID_ARG, "fri instance: oracle of another context");
        ncols[o] = oracles[o]->n_cols;
    }
    hipSetDevice(ctx->device);
    const u32 log_n = oracles[0]->log_n;
    if (log_n + cfg->rate_bits < cfg->cap_height) return ctx->fail(ZK_ERR_INVALID_ARG, "fri instance: cap_height above the tree");
    if (!zk_fri_reduction_ok(*cfg, log_n)) return ctx->fail(ZK_ERR_INVALID_ARG, "fri instance: the reduction strategy does not fit this size (upstream panics)");
    const size_t fri_words = zk_fri_instance_proof_size_words(cfg, log_n, n_oracles, ncols);
    if (len_words) *len_words = fri_words;
    if (cap_words < fri_words) return ZK_ERR_BUFFER_TOO_SMALL;
    FriLayout lay = {};
    u32 ar[64], nl; zk_fri_arities(*cfg, log_n, ar, &nl);
    const u32 capw = 4u << cfg->cap_height;
    size_t o = 0;
    lay.o_nl = o; o += 1; lay.o_layer_caps = o; o += (size_t)nl * capw;
    u32 d = log_n; for (u32 l = 0; l < nl; l++) d -= ar[l];
    lay.o_flen = o; o += 1; lay.o_final = o; o += 2 * ((size_t)1 << d); lay.o_witness = o; o += 1;
    lay.o_queries = o;
    lay.per_query = (fri_words - o) / cfg->num_query_rounds;
    FriBatchDesc desc[4];
    for (u32 b = 0; b < n_batches; b++) {
        if (!batches[b].polys) return ctx->fail(ZK_ERR_INVALID_ARG, "fri instance: batch without polynomials");
        desc[b].point[0] = gl_canon(batches[b].point[0]); desc[b].point[1] = gl_canon(batches[b].point[1]);
        desc[b].n_polys = batches[b].n_polys; desc[b].polys = batches[b].polys;
    }
    u64 *d_buf = (u64 *)ctx->alloc(fri_words * sizeof(u64));
    if (!d_buf) return ZK_ERR_OOM;
    int32_t rc = zk_fri_prove_instance_dev(ctx, *cfg, ch->d, oracles, n_oracles, desc, n_batches, d_buf, lay);
    if (rc == ZK_OK) rc = zk_dev_download(ctx, fri_proof_out, d_buf, fri_words * sizeof(u64));
    u64 err = 0;
    if (rc == ZK_OK) rc = zk_dev_download(ctx, &err, ctx->d_err, sizeof(u64));
    if (rc == ZK_OK && err) { rc = (int32_t)(uint32_t)err; hipMemsetAsync(ctx->d_err, 0, sizeof(u64), ctx->stream); }
    ctx->release(d_buf);
    return rc;
}

}  // extern "C"
