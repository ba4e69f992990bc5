// Batched constraint-quotient evaluation over the LDE coset.
// Mirrors starky 0.4.0 prover::compute_quotient_polys + constraint_consumer::ConstraintConsumer +
// vanishing_poly::eval_vanishing_poly (crate pinned at /root/reference/Cargo.lock:4529-4531).
// The table's `Stark::eval_packed_generic` body is supplied as data (zk_air: a straight-line program), because the
// seven zkEVM tables' bodies live in evm_arithmetization 0.1.3 whose source is absent (Cargo.lock:1675-1677).
//
// One thread = one point of the coset g*<w_L> in leaf order (position s <-> natural index i = bitrev(s)), so every
// column read is a coalesced wave access; the "next row" is position bitrev(i + 2^rate_bits), which differs from s
// only in high bits for all but a vanishing fraction of waves.  Temporaries live in LDS ([reg][thread]); the two
// alpha-Horner accumulators stay in VGPRs.  Output: quotient values in leaf order, then one inverse DIT NTT gives
// natural-order coefficients whose n-sized chunks are the quotient polynomials.
#include "zk_internal.h"
#include "prover.h"

struct QParams {
    const zk_instr *instrs; u32 n_instrs, n_regs;
    const u64 *constants, *pub, *chal, *alphas;
    const u64 *trace_lde, *aux_lde;
    u32 C, A, lgL, nch, rate_bits;
    size_t L;
    const u64 *lag;            // L_first | L_last on the coset, leaf order
    const u64 *xA, *xB;        // x_i = g * w_L^i (two-level, natural index)
    u64 last_inv;              // w_n^-1 : z_last(x) = x - w_n^-1
    u64 zh_inv[8];             // 1/Z_H on the coset: index i mod 2^rate_bits
    u64 *out;                  // nch x L, leaf order
};

__device__ __forceinline__ u64 q_fetch(const QParams &p, u32 o, const u64 *sh, u32 t, size_t s, size_t sn) {
    u32 kind = o >> 24, idx = o & 0xFFFFFF;
    switch (kind) {
    case ZK_OPND_REG: return sh[idx * 256 + t];
    case ZK_OPND_LOCAL: return idx < p.C ? p.trace_lde[(size_t)idx * p.L + s] : p.aux_lde[(size_t)(idx - p.C) * p.L + s];
    case ZK_OPND_NEXT: return idx < p.C ? p.trace_lde[(size_t)idx * p.L + sn] : p.aux_lde[(size_t)(idx - p.C) * p.L + sn];
    case ZK_OPND_CONST: return p.constants[idx];
    case ZK_OPND_PUBLIC: return p.pub[idx];
    default: return p.chal[idx];
    }
}

// one workgroup-sized slice of the coset (256*NP points from `block`*256*NP) through the bytecode interpreter.  NP = 2: every thread
// carries two points (s and s + 256), so instruction fetch, decode and the operand-kind dispatch -- two thirds of the interpreter's
// instruction stream -- are paid once for two evaluations; both points' accesses stay coalesced.
// EXACT = false: products through glx<false> (gl64.cuh: the two 1-in-2^32 repairs skipped behind a wave-uniform sticky word); the
// caller re-runs the slice with EXACT = true when the word comes back non-zero, so the stored values are the same bit for bit.
// (`sh`, the temporaries, is typed as LDS: through a generic pointer hipcc reads and writes it with flat instructions)
typedef __attribute__((address_space(3))) u64 *q_lds_ptr;
template <int NP, bool EXACT> __device__ __forceinline__ u64 quotient_block(const QParams &p, u32 block, q_lds_ptr sh) {
    glx<EXACT> g; g.fl = 0;
    const u32 n_instrs = (u32)__builtin_amdgcn_readfirstlane((int)p.n_instrs), nch = (u32)__builtin_amdgcn_readfirstlane((int)p.nch);   // wave-uniform, and known to be
    const u32 t = threadIdx.x;
    size_t s[NP], sn[NP]; u32 i[NP]; u64 z_last[NP], l_first[NP], l_last[NP];
    u64 acc[NP][4], al[4];
    bool live[NP];
#pragma unroll
    for (int k = 0; k < NP; k++) {
        s[k] = (size_t)block * (256 * NP) + 256 * k + t;
        live[k] = s[k] < p.L;
        const size_t sc = live[k] ? s[k] : 0;                    // dead points (ragged tail) evaluate point 0 and store nothing
        i[k] = bitrev_u32((u32)sc, p.lgL);
        sn[k] = bitrev_u32((i[k] + (1u << p.rate_bits)) & (u32)(p.L - 1), p.lgL);
        s[k] = sc;
        const u64 x = g.mul(p.xA[i[k] & 2047], p.xB[i[k] >> 11]);
        z_last[k] = gl_sub(x, p.last_inv);
        l_first[k] = p.lag[sc]; l_last[k] = p.lag[p.L + sc];
#pragma unroll
        for (int j = 0; j < 4; j++) acc[k][j] = 0;
    }
#pragma unroll
    for (int j = 0; j < 4; j++) al[j] = j < (int)nch ? p.alphas[j] : 0;
    for (u32 pc = 0; pc < n_instrs; pc++) {
        // the instruction word is the same for every lane; saying so keeps the whole loop body in wave-uniform control flow,
        // which the scalar sticky word of glx<false> needs
        zk_instr in = p.instrs[pc];
        in.op = (uint32_t)__builtin_amdgcn_readfirstlane((int)in.op); in.dst = (uint32_t)__builtin_amdgcn_readfirstlane((int)in.dst);
        in.a = (uint32_t)__builtin_amdgcn_readfirstlane((int)in.a); in.b = (uint32_t)__builtin_amdgcn_readfirstlane((int)in.b);
        u64 a[NP];
        {
            const u32 kind = in.a >> 24, idx = in.a & 0xFFFFFF;
            switch (kind) {
            case ZK_OPND_REG:
#pragma unroll
                for (int k = 0; k < NP; k++) a[k] = sh[(idx * NP + k) * 256 + t];
                break;
            case ZK_OPND_LOCAL: {
                const u64 *col = idx < p.C ? p.trace_lde + (size_t)idx * p.L : p.aux_lde + (size_t)(idx - p.C) * p.L;
#pragma unroll
                for (int k = 0; k < NP; k++) a[k] = col[s[k]];
                break; }
            case ZK_OPND_NEXT: {
                const u64 *col = idx < p.C ? p.trace_lde + (size_t)idx * p.L : p.aux_lde + (size_t)(idx - p.C) * p.L;
#pragma unroll
                for (int k = 0; k < NP; k++) a[k] = col[sn[k]];
                break; }
            default: {
                const u64 v = kind == ZK_OPND_CONST ? p.constants[idx] : kind == ZK_OPND_PUBLIC ? p.pub[idx] : p.chal[idx];
#pragma unroll
                for (int k = 0; k < NP; k++) a[k] = v;
                break; }
            }
        }
        if (in.op <= ZK_OP_MUL) {
            u64 b[NP];
            const u32 kind = in.b >> 24, idx = in.b & 0xFFFFFF;
            switch (kind) {
            case ZK_OPND_REG:
#pragma unroll
                for (int k = 0; k < NP; k++) b[k] = sh[(idx * NP + k) * 256 + t];
                break;
            case ZK_OPND_LOCAL: {
                const u64 *col = idx < p.C ? p.trace_lde + (size_t)idx * p.L : p.aux_lde + (size_t)(idx - p.C) * p.L;
#pragma unroll
                for (int k = 0; k < NP; k++) b[k] = col[s[k]];
                break; }
            case ZK_OPND_NEXT: {
                const u64 *col = idx < p.C ? p.trace_lde + (size_t)idx * p.L : p.aux_lde + (size_t)(idx - p.C) * p.L;
#pragma unroll
                for (int k = 0; k < NP; k++) b[k] = col[sn[k]];
                break; }
            default: {
                const u64 v = kind == ZK_OPND_CONST ? p.constants[idx] : kind == ZK_OPND_PUBLIC ? p.pub[idx] : p.chal[idx];
#pragma unroll
                for (int k = 0; k < NP; k++) b[k] = v;
                break; }
            }
            const u32 dst = in.dst & 0xFFFFFF;
            if (in.op == ZK_OP_ADD) {
#pragma unroll
                for (int k = 0; k < NP; k++) sh[(dst * NP + k) * 256 + t] = gl_add(a[k], b[k]);
            } else if (in.op == ZK_OP_SUB) {
#pragma unroll
                for (int k = 0; k < NP; k++) sh[(dst * NP + k) * 256 + t] = gl_sub(a[k], b[k]);
            } else {
#pragma unroll
                for (int k = 0; k < NP; k++) sh[(dst * NP + k) * 256 + t] = g.mul(a[k], b[k]);
            }
        } else {
#pragma unroll
            for (int k = 0; k < NP; k++) {
                // the filter is a select, the product unconditional (a plain EMIT multiplies by 1): no branch around a product
                const u64 m = in.op == ZK_OP_EMIT_TRANSITION ? z_last[k] : in.op == ZK_OP_EMIT_FIRST ? l_first[k] : in.op == ZK_OP_EMIT_LAST ? l_last[k] : 1;
                const u64 c = g.mul(a[k], m);
#pragma unroll
                for (int j = 0; j < 4; j++) if (j < (int)nch) acc[k][j] = g.mul_add(acc[k][j], al[j], c);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < NP; k++) {
        // (a select over constant indices: a run-time index into the by-value parameter struct would pin the WHOLE struct in scratch,
        // and every pointer in it would then be dereferenced with flat instead of global loads)
        const u32 zsel = i[k] & ((1u << p.rate_bits) - 1);
        u64 zi = p.zh_inv[0];
#pragma unroll
        for (u32 j = 1; j < 8; j++) zi = zsel == j ? p.zh_inv[j] : zi;
#pragma unroll
        for (int j = 0; j < 4; j++) if (j < (int)nch) {
            const u64 q = g.mul(acc[k][j], zi);                      // products stay outside lane-dependent control flow
            if (live[k]) p.out[(size_t)j * p.L + s[k]] = q;
        }
    }
    return g.fl;
}
__global__ __launch_bounds__(256) void quotient_kernel(QParams p, u64 force_redo) {
    extern __shared__ u64 sh[];
    // temporaries are [reg][point][thread]: a wave only ever touches its own lanes' slots, so a flagged wave can redo its part alone
    if (quotient_block<2, false>(p, blockIdx.x, (q_lds_ptr)sh) | force_redo) quotient_block<2, true>(p, blockIdx.x, (q_lds_ptr)sh);
}
__global__ void canon_copy_kernel(const u64 *in, u64 *out, u32 n) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = gl_canon(in[i]);
}
__global__ void set_indicator_kernel(u64 *ind, size_t n) { if (threadIdx.x == 0 && blockIdx.x == 0) { ind[0] = 1; ind[n + n - 1] = 1; } }

// lagrange_first / lagrange_last = PolynomialValues::selector(n, idx).lde_onto_coset(rate_bits), leaf order; cached per size
int32_t zk_lagrange_table(zk_ctx *ctx, u32 log_n, u32 rate_bits, u64 **out) {
    auto key = std::make_pair((int)log_n, (int)rate_bits);
    auto it = ctx->lagrange_tab.find(key);
    if (it != ctx->lagrange_tab.end()) { *out = it->second; return ZK_OK; }
    size_t n = (size_t)1 << log_n, L = n << rate_bits;
    u64 *ind = (u64 *)ctx->alloc(2 * n * sizeof(u64)), *co = (u64 *)ctx->alloc(2 * n * sizeof(u64)), *tmp = (u64 *)ctx->alloc(2 * n * sizeof(u64));
    u64 *lde = (u64 *)ctx->alloc(2 * L * sizeof(u64));
    if (!ind || !co || !tmp || !lde) return ZK_ERR_OOM;
    hipMemsetAsync(ind, 0, 2 * n * sizeof(u64), ctx->stream);
    ZK_LAUNCH(ctx, "set_indicator", set_indicator_kernel, dim3(1), dim3(1), 0, ind, n);
    int32_t rc = zk_ntt_dif(ctx, ind, n, co, n, log_n, 2, 1, nullptr, gl_inv((u64)n % GL_P), 0);
    if (rc == ZK_OK) rc = zk_ntt_lde(ctx, co, n, lde, L, log_n, 2, rate_bits, tmp, n);
    ctx->release(ind); ctx->release(co); ctx->release(tmp);
    if (rc != ZK_OK) { ctx->release(lde); return rc; }
    ctx->lagrange_tab[key] = lde;
    *out = lde;
    return ZK_OK;
}

int32_t zk_quotient_dev(zk_ctx *ctx, const DevAir &air, const zk_stark_config &cfg, const zk_batch *trace, const zk_batch *aux,
                        const u64 *d_public, const u64 *d_chal, const u64 *d_alphas, u64 *d_qcoeffs) {
    const u32 log_n = trace->log_n, rate_bits = cfg.rate_bits, lgL = log_n + rate_bits, nch = cfg.num_challenges;
    const size_t n = (size_t)1 << log_n, L = n << rate_bits;
    if (nch > 4 || rate_bits > 3) return ctx->fail(ZK_ERR_INVALID_ARG, "quotient: num_challenges <= 4, rate_bits <= 3");
    if (air.host->constraint_degree != 3 || rate_bits != 1)   // quotient_degree_bits == rate_bits == 1
        return ctx->fail(ZK_ERR_INVALID_ARG, "quotient: only constraint degree 3 with rate_bits 1 is supported");
    if (air.host->n_regs * 512 * sizeof(u64) > 64 * 1024) return ctx->fail(ZK_ERR_INVALID_ARG, "quotient: too many registers");
    QParams p = {};
    p.instrs = air.d_instrs; p.n_instrs = air.host->n_instrs; p.n_regs = air.host->n_regs;
    p.constants = air.d_constants; p.pub = d_public; p.chal = d_chal; p.alphas = d_alphas;
    p.trace_lde = trace->lde; p.aux_lde = aux ? aux->lde : nullptr;
    p.C = trace->n_cols; p.A = aux ? aux->n_cols : 0; p.lgL = lgL; p.nch = nch; p.rate_bits = rate_bits; p.L = L;
    u64 *lag = nullptr;
    int32_t rc = zk_lagrange_table(ctx, log_n, rate_bits, &lag);
    if (rc) return rc;
    p.lag = lag;
    TwoLevel xs = zk_scale_table(ctx, gl_root(lgL), GL_GEN, lgL);
    p.xA = xs.A; p.xB = xs.B;
    p.last_inv = gl_inv(gl_root(log_n));
    u64 gn = gl_pow(GL_GEN, n), wr = gl_root(rate_bits);
    for (u32 i = 0; i < (1u << rate_bits); i++) p.zh_inv[i] = gl_inv(gl_sub(gl_mul(gn, gl_pow(wr, i)), 1));
    p.out = d_qcoeffs;
    const size_t shmem = (size_t)(air.host->n_regs ? air.host->n_regs : 1) * 256 * sizeof(u64);
    bool done = false;
    if (zk_quotient_jit_enabled(ctx, L)) {
        // compiled kernel for this program, if there is one (built with the library, or by hiprtc when that is switched on); a wave
        // that skipped a rare repair redoes its points exactly inside the same kernel
        QJitParams jp = {};
        jp.trace_lde = p.trace_lde; jp.aux_lde = p.aux_lde; jp.pub = p.pub; jp.chal = p.chal; jp.alphas = p.alphas; jp.lag = p.lag;
        jp.xA = p.xA; jp.xB = p.xB; jp.out = p.out; jp.L = L; jp.last_inv = p.last_inv; jp.lgL = lgL;
        jp.state = nullptr; jp.redo_cap = 0; jp.force = ctx->force_redo ? 1u : 0u;
        // blocks walk the cycle of `next` rows, one arc per XCD (zk_qblock, gl64.cuh); ZK_QUOTIENT_REMAP=0: plain order (A/B switch)
        static const int remap = getenv("ZK_QUOTIENT_REMAP") ? atoi(getenv("ZK_QUOTIENT_REMAP")) : 1;
        jp.pad = (remap && lgL >= 8 + 3 + rate_bits && (L & 2047) == 0) ? 1u : 0u;
        for (int i = 0; i < 8; i++) jp.zh_inv[i] = p.zh_inv[i];
        rc = zk_quotient_jit_launch(ctx, air.host, p.C, p.A, nch, rate_bits, jp);
        if (rc == ZK_OK) done = true;
        else if (rc != ZK_JIT_UNAVAILABLE) return rc;
    }
    if (!done) ZK_LAUNCH(ctx, "quotient_eval", quotient_kernel, dim3((unsigned)((L + 511) / 512)), dim3(256), 2 * shmem, p, ctx->force_redo);
    // coset_ifft(g): leaf-order values -> natural-order coefficients, scaled by L^-1 g^-j.  In place.
    TwoLevel post = zk_scale_table(ctx, gl_inv(GL_GEN), gl_inv((u64)L % GL_P), lgL);
    return zk_ntt_dit(ctx, d_qcoeffs, L, d_qcoeffs, L, lgL, nch, 1, nullptr, &post, 0, nullptr, 0);
}
