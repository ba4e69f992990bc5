// FRI opening proof on the device: batch combination, commit phase (fold + Merkle per layer), proof of work, queries.
// Mirrors plonky2 0.2.2 fri::oracle::PolynomialBatch::prove_openings and fri::prover::{fri_proof, fri_committed_trees,
// fri_proof_of_work, fri_prover_query_rounds} (crate pinned at /root/reference/Cargo.lock:3441-3443; SURVEY.md A8,
// 8a row a13).  Reached from the reference at /root/reference/ops/src/lib.rs:52.
//
// Same field values as upstream, different route (all exact arithmetic, so the bits agree):
//   * sum_j alpha^j f_j is formed on the bit-reversed coefficient arrays (one coalesced pass over every column);
//   * (P - P(z))/(X - z) is taken pointwise on the LDE coset -- P(z) is the alpha-reduction of the openings -- instead
//     of a serial synthetic division; the result IS layer 0 of the commit phase, already in leaf order;
//   * folding is done on natural-order coefficients (contiguous groups of 16), each next layer is one DIF NTT.
// Extension vectors are stored as two base columns (re | im) so the NTT kernels treat them as a 2-column batch.
#include <algorithm>
#include "zk_internal.h"
#include "prover.h"
#include "poseidon.cuh"

__device__ inline gl2 gl2_pow_u32(gl2 b, u32 e) {
    gl2 r = gl2_make(1, 0);
    while (e) { if (e & 1) r = gl2_mul(r, b); b = gl2_mul(b, b); e >>= 1; }
    return r;
}
__device__ __forceinline__ u64 gl_inv_fermat(u64 a) {
    u64 x2 = gl_mul(gl_sqr(a), a);
    u64 x4 = gl_mul(gl_sqr(gl_sqr(x2)), x2);
    u64 x8 = x4; for (int i = 0; i < 4; i++) x8 = gl_sqr(x8); x8 = gl_mul(x8, x4);
    u64 x16 = x8; for (int i = 0; i < 8; i++) x16 = gl_sqr(x16); x16 = gl_mul(x16, x8);
    u64 x24 = x16; for (int i = 0; i < 8; i++) x24 = gl_sqr(x24); x24 = gl_mul(x24, x8);
    u64 x28 = x24; for (int i = 0; i < 4; i++) x28 = gl_sqr(x28); x28 = gl_mul(x28, x4);
    u64 x30 = gl_mul(gl_sqr(gl_sqr(x28)), x2);
    u64 x31 = gl_mul(gl_sqr(x30), a);
    u64 x32 = gl_mul(gl_sqr(x31), a);
    u64 r = gl_sqr(x31);
    for (int i = 0; i < 32; i++) r = gl_sqr(r);
    return gl_mul(r, x32);
}

// alpha^j for j < cnt, plus the two ReducingFactor shifts alpha^(C+A), alpha^nch at the end
__global__ void alpha_pows_kernel(const u64 *alpha, u32 cnt, u32 sh1, u32 sh2, u64 *out) {
    u32 j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= cnt + 2) return;
    gl2 a = gl2_make(alpha[0], alpha[1]);
    u32 e = j < cnt ? j : (j == cnt ? sh1 : sh2);
    gl2 r = gl2_pow_u32(a, e);
    out[2 * j] = r.a; out[2 * j + 1] = r.b;
}

struct CombineParams {
    const u64 *seg[3]; u32 cnt[3];     // trace, aux, quotient coefficient matrices (column stride n)
    u32 nch;                           // the last nch aux columns are the Z columns (batch at the point 1)
    size_t n;
    const u64 *apow;                   // alpha^j (ext)
    u64 *P;                            // 6 columns of n: P0.re P0.im P1.re P1.im P2.re P2.im
};
// ReducingFactor::reduce_polys_base for all three batches in one pass over the coefficient columns
__global__ __launch_bounds__(256) void fri_combine_kernel(CombineParams p) {
    size_t pos = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (pos >= p.n) return;
    u64 ra = 0, rb = 0;
    u32 j = 0;
    u64 p1a = 0, p1b = 0;
    for (int sgi = 0; sgi < 3; sgi++) {
        if (sgi == 2) { p1a = ra; p1b = rb; }          // batch g*zeta = trace | aux prefix
        const u64 *base = p.seg[sgi] + pos;
        for (u32 c = 0; c < p.cnt[sgi]; c++, j++) {
            u64 v = base[(size_t)c * p.n];
            ra = gl_add(ra, gl_mul(v, p.apow[2 * j]));
            rb = gl_add(rb, gl_mul(v, p.apow[2 * j + 1]));
        }
    }
    p.P[pos] = ra; p.P[p.n + pos] = rb;
    p.P[2 * p.n + pos] = p1a; p.P[3 * p.n + pos] = p1b;
    if (p.nch && p.cnt[1]) {
        u64 za = 0, zb = 0;
        const u64 *zc = p.seg[1] + (size_t)(p.cnt[1] - p.nch) * p.n + pos;
        for (u32 k = 0; k < p.nch; k++) {
            u64 v = zc[(size_t)k * p.n];
            za = gl_add(za, gl_mul(v, p.apow[2 * k])); zb = gl_add(zb, gl_mul(v, p.apow[2 * k + 1]));
        }
        p.P[4 * p.n + pos] = za; p.P[5 * p.n + pos] = zb;
    }
}

// The same for small tables (n <= 2^16: fewer waves than SIMDs, and a thread walking thousands of columns is a long serial chain):
// 8 threads share a position, each takes an eighth of the columns, partial sums meet in LDS.  Field addition is exact, so the order
// of the summation does not show in the result.
__global__ __launch_bounds__(256) void fri_combine_split_kernel(CombineParams p) {
    __shared__ u64 part[4][8][32];
    const u32 t = threadIdx.x, lp = t & 31, slice = t >> 5;
    const size_t pos = (size_t)blockIdx.x * 32 + lp;
    const bool live = pos < p.n;
    const u32 total = p.cnt[0] + p.cnt[1] + p.cnt[2], per = (total + 7) / 8, lo = slice * per, hi = lo + per < total ? lo + per : total;
    u64 ra = 0, rb = 0, p1a = 0, p1b = 0;
    u32 j0 = 0;
    for (int sgi = 0; sgi < 3; sgi++) {
        const u32 cnt = p.cnt[sgi];
        const u32 c0 = lo > j0 ? lo - j0 : 0, c1 = hi > j0 ? (hi - j0 < cnt ? hi - j0 : cnt) : 0;
        if (live && c0 < c1) {
            const u64 *base = p.seg[sgi] + pos;
            u64 sa = 0, sb = 0;
            for (u32 c = c0; c < c1; c++) {
                const u64 v = base[(size_t)c * p.n];
                sa = gl_add(sa, gl_mul(v, p.apow[2 * (j0 + c)]));
                sb = gl_add(sb, gl_mul(v, p.apow[2 * (j0 + c) + 1]));
            }
            ra = gl_add(ra, sa); rb = gl_add(rb, sb);
            if (sgi < 2) { p1a = gl_add(p1a, sa); p1b = gl_add(p1b, sb); }         // batch g*zeta = trace | aux prefix
        }
        j0 += cnt;
    }
    part[0][slice][lp] = ra; part[1][slice][lp] = rb; part[2][slice][lp] = p1a; part[3][slice][lp] = p1b;
    __syncthreads();
    if (slice == 0 && live) {
        u64 acc[4];
#pragma unroll
        for (int k = 0; k < 4; k++) { acc[k] = part[k][0][lp]; for (int q = 1; q < 8; q++) acc[k] = gl_add(acc[k], part[k][q][lp]); }
        p.P[pos] = acc[0]; p.P[p.n + pos] = acc[1]; p.P[2 * p.n + pos] = acc[2]; p.P[3 * p.n + pos] = acc[3];
        if (p.nch && p.cnt[1]) {
            u64 za = 0, zb = 0;
            const u64 *zc = p.seg[1] + (size_t)(p.cnt[1] - p.nch) * p.n + pos;
            for (u32 k = 0; k < p.nch; k++) {
                u64 v = zc[(size_t)k * p.n];
                za = gl_add(za, gl_mul(v, p.apow[2 * k])); zb = gl_add(zb, gl_mul(v, p.apow[2 * k + 1]));
            }
            p.P[4 * p.n + pos] = za; p.P[5 * p.n + pos] = zb;
        }
    }
}

// PrecomputedReducedOpenings: y_b = sum_j alpha^j opening_j per batch, read from the proof buffer
__global__ void reduce_openings_kernel(const u64 *proof, FriLayout lay, u32 C, u32 A, u32 Q, u32 nch, const u64 *apow, u64 *y) {
    __shared__ u64 red[2 * 256];
    u32 b = blockIdx.x, t = threadIdx.x;
    gl2 acc = gl2_make(0, 0);
    u32 cnt = b == 0 ? C + A + Q : b == 1 ? C + A : nch;
    for (u32 j = t; j < cnt; j += 256) {
        gl2 v;
        if (b == 0) {
            const u64 *src = j < C ? proof + lay.o_local + 2 * j : j < C + A ? proof + lay.o_aux + 2 * (j - C) : proof + lay.o_q + 2 * (j - C - A);
            v = gl2_make(src[0], src[1]);
        } else if (b == 1) {
            const u64 *src = j < C ? proof + lay.o_next + 2 * j : proof + lay.o_auxn + 2 * (j - C);
            v = gl2_make(src[0], src[1]);
        } else v = gl2_make(proof[lay.o_zs + j], 0);
        acc = gl2_add(acc, gl2_mul(v, gl2_make(apow[2 * j], apow[2 * j + 1])));
    }
    red[2 * t] = acc.a; red[2 * t + 1] = acc.b; __syncthreads();
    for (u32 s = 128; s > 0; s >>= 1) {
        if (t < s) { red[2 * t] = gl_add(red[2 * t], red[2 * (t + s)]); red[2 * t + 1] = gl_add(red[2 * t + 1], red[2 * (t + s) + 1]); }
        __syncthreads();
    }
    if (t == 0) { y[2 * b] = red[0]; y[2 * b + 1] = red[1]; }
}

struct QuotParams {
    const u64 *Pval;          // 2*nb columns of L (leaf order)
    size_t L; u32 lgL, nb;
    const u64 *xA, *xB;       // x_i = g w_L^i
    const u64 *pts;           // zeta (2), g_n*zeta (2)
    const u64 *y;             // reduced openings (2 per batch)
    const u64 *shifts;        // alpha^(C+A) (2), alpha^nch (2)
    u64 *F;                   // F.re | F.im, leaf order
};
// final values  F(x) = ((q0 * a^cnt1 + q1) * a^cnt2 + q2)(x),  q_b(x) = (P_b(x) - y_b) / (x - z_b)
__global__ __launch_bounds__(256) void fri_quotient_kernel(QuotParams p) {
    size_t s = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= p.L) return;
    u32 i = bitrev_u32((u32)s, p.lgL);
    u64 x = gl_mul(p.xA[i & 2047], p.xB[i >> 11]);
    gl2 den[3], num[3];
    u64 nrm[3], pre[3];
    u64 acc = 1;
#pragma unroll
    for (int b = 0; b < 3; b++) {
        if (b < (int)p.nb) {
            gl2 z = b == 0 ? gl2_make(p.pts[0], p.pts[1]) : b == 1 ? gl2_make(p.pts[2], p.pts[3]) : gl2_make(1, 0);
            den[b] = gl2_make(gl_sub(x, z.a), gl_neg(z.b));
            u64 bb = gl_mul(den[b].b, den[b].b);
            u64 w7 = gl_add(gl_add(gl_dbl(gl_dbl(bb)), gl_dbl(bb)), bb);
            nrm[b] = gl_sub(gl_mul(den[b].a, den[b].a), w7);            // norm = a^2 - 7 b^2
            num[b] = gl2_make(gl_sub(p.Pval[(size_t)(2 * b) * p.L + s], p.y[2 * b]), gl_sub(p.Pval[(size_t)(2 * b + 1) * p.L + s], p.y[2 * b + 1]));
        } else { nrm[b] = 1; den[b] = gl2_make(1, 0); num[b] = gl2_make(0, 0); }
        pre[b] = acc; acc = gl_mul(acc, nrm[b]);
    }
    u64 inv = gl_inv_fermat(acc);
    gl2 F = gl2_make(0, 0);
    gl2 q[3];
#pragma unroll
    for (int b = 2; b >= 0; b--) {
        u64 ninv = gl_mul(inv, pre[b]);
        inv = gl_mul(inv, nrm[b]);
        gl2 dinv = gl2_make(gl_mul(den[b].a, ninv), gl_mul(gl_neg(den[b].b), ninv));   // conj / norm
        q[b] = gl2_mul(num[b], dinv);
    }
    F = q[0];
    F = gl2_add(gl2_mul(F, gl2_make(p.shifts[0], p.shifts[1])), q[1]);
    if (p.nb == 3) F = gl2_add(gl2_mul(F, gl2_make(p.shifts[2], p.shifts[3])), q[2]);
    p.F[s] = F.a; p.F[p.L + s] = F.b;
}

// leaves of a commit-phase layer: `arity` consecutive extension values, flattened [a0,b0,a1,b1,...]
template <bool EXACT> __device__ __forceinline__ u64 fri_leaf_sponge(const u64 *__restrict__ r, const u64 *__restrict__ m, u32 arity, u64 s[12]) {
    psd::core<EXACT> pc; pc.fl = 0;
#pragma unroll
    for (int k = 0; k < 12; k++) s[k] = 0;
    for (u32 c = 0; c < arity; c += 4) {
#pragma unroll
        for (int k = 0; k < 4; k++) { s[2 * k] = r[c + k]; s[2 * k + 1] = m[c + k]; }
        pc.run(s);
    }
    return pc.fl;
}
__global__ __launch_bounds__(256) void fri_leaf_hash_kernel(const u64 *__restrict__ re, const u64 *__restrict__ im, size_t n_leaves, u32 arity,
                                                            u64 *__restrict__ out, u64 force_redo) {
    size_t l = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (l >= n_leaves) return;
    u64 s[12];
    const u64 *r = re + l * arity, *m = im + l * arity;
    if (fri_leaf_sponge<false>(r, m, arity, s) | force_redo) fri_leaf_sponge<true>(r, m, arity, s);   // see poseidon.cuh: rare-repair redo
#pragma unroll
    for (int k = 0; k < 4; k++) out[4 * l + k] = s[k];
}

// coeffs'[i] = reduce_with_powers(coeffs[arity*i .. arity*(i+1)], beta)   (natural order, re|im columns)
__global__ __launch_bounds__(256) void fri_fold_kernel(const u64 *__restrict__ re, const u64 *__restrict__ im, size_t n_out, u32 arity,
                                                       const u64 *beta, u64 *__restrict__ ore, u64 *__restrict__ oim) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_out) return;
    gl2 b = gl2_make(beta[0], beta[1]);
    gl2 acc = gl2_make(0, 0);
    for (u32 t = arity; t-- > 0;) acc = gl2_add(gl2_mul(acc, b), gl2_make(re[i * arity + t], im[i * arity + t]));
    ore[i] = acc.a; oim[i] = acc.b;
}

__global__ void interleave_kernel(const u64 *re, const u64 *im, u32 n, u64 *out) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { out[2 * i] = re[i]; out[2 * i + 1] = im[i]; }
}
__global__ void copy_words_kernel(const u64 *src, u64 *dst, u32 n) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

struct QueryParams {
    u32 n_oracles, n_layers, cap_height, lgL;
    const u64 *lde[8]; const u64 *levels[8]; u32 n_cols[8];
    const u64 *lre[8], *lim[8], *llevels[8]; u32 arity_bits[8];
    const u64 *idx;
    u64 *out; size_t per_query;
};
// fri_prover_query_rounds: one workgroup per query round gathers leaves and Merkle paths into the proof buffer
__global__ __launch_bounds__(256) void fri_query_kernel(QueryParams p) {
    const u32 q = blockIdx.x, t = threadIdx.x;
    u64 *dst = p.out + (size_t)q * p.per_query;
    size_t x = (size_t)p.idx[q];
    const size_t L = (size_t)1 << p.lgL;
    const u32 npath = p.lgL - p.cap_height;
    for (u32 o = 0; o < p.n_oracles; o++) {
        for (u32 c = t; c < p.n_cols[o]; c += 256) dst[c] = p.lde[o][(size_t)c * L + x];
        dst += p.n_cols[o];
        for (u32 w = t; w < npath * 4; w += 256) {
            u32 k = w >> 2;
            size_t off = 0;
            for (u32 j = 0; j < k; j++) off += L >> j;
            dst[w] = p.levels[o][4 * (off + ((x >> k) ^ 1)) + (w & 3)];
        }
        dst += (size_t)npath * 4;
    }
    u32 lg = p.lgL;
    for (u32 l = 0; l < p.n_layers; l++) {
        u32 ab = p.arity_bits[l], arity = 1u << ab;
        size_t leaf = x >> ab;
        for (u32 w = t; w < 2 * arity; w += 256) dst[w] = (w & 1) ? p.lim[l][leaf * arity + (w >> 1)] : p.lre[l][leaf * arity + (w >> 1)];
        dst += 2 * arity;
        u32 depth = lg - ab, ns = depth - p.cap_height;
        size_t nl = (size_t)1 << depth;
        for (u32 w = t; w < ns * 4; w += 256) {
            u32 k = w >> 2;
            size_t off = 0;
            for (u32 j = 0; j < k; j++) off += nl >> j;
            dst[w] = p.llevels[l][4 * (off + ((leaf >> k) ^ 1)) + (w & 3)];
        }
        dst += (size_t)ns * 4;
        x = leaf; lg = depth;
    }
}

__global__ void write_word_kernel(u64 *dst, u64 v) { if (threadIdx.x == 0 && blockIdx.x == 0) *dst = v; }

static inline dim3 g1(size_t n) { return dim3((unsigned)((n + 255) / 256)); }

int32_t zk_fri_prove_dev(zk_ctx *ctx, const zk_stark_config &cfg, DevChallenger *ch, const zk_batch *trace, const zk_batch *aux,
                         const zk_batch *quot, const u64 *d_zeta, u64 *d_proof, const FriLayout &lay) {
    const u32 log_n = trace->log_n, rate_bits = cfg.rate_bits, lgL = log_n + rate_bits, nch = cfg.num_challenges;
    const size_t n = (size_t)1 << log_n, L = n << rate_bits;
    const u32 C = trace->n_cols, A = aux ? aux->n_cols : 0, Q = quot->n_cols;
    const u32 nb = A ? 3 : 2, capw = 4u << cfg.cap_height;
    u32 ar[64], nl; zk_fri_arities(cfg, log_n, ar, &nl);
    if (nl > 8) return ctx->fail(ZK_ERR_INVALID_ARG, "fri: more than 8 reduction layers");
    int32_t rc = ZK_OK;

    // ---- alpha, alpha powers, reduced openings
    ZkOwned own(ctx);                       // every exit path hands the scratch blocks back
    u64 *d_alpha = (u64 *)own.get(2 * sizeof(u64));
    u32 cnt = C + A + Q;
    u64 *d_apow = (u64 *)own.get((size_t)(cnt + 2) * 2 * sizeof(u64));
    u64 *d_y = (u64 *)own.get(6 * sizeof(u64));
    u64 *P = (u64 *)own.get(6 * n * sizeof(u64)), *Pval = (u64 *)own.get(6 * L * sizeof(u64)), *tmp = (u64 *)own.get(6 * n * sizeof(u64));
    u64 *F = (u64 *)own.get(2 * L * sizeof(u64));
    if (!d_alpha || !d_apow || !d_y || !P || !Pval || !tmp || !F) return ctx->fail(ZK_ERR_OOM, "fri: out of device memory");
    { TOpList a = {}; a.n = 1; a.op[0].kind = TOP_GET; a.op[0].dst = d_alpha; a.op[0].n = 2; zk_transcript_run(ctx, ch, a); }
    ZK_LAUNCH(ctx, "fri_alpha_pows", alpha_pows_kernel, g1(cnt + 2), dim3(256), 0, (const u64 *)d_alpha, cnt, C + A, nch, d_apow);
    ZK_LAUNCH(ctx, "fri_reduce_openings", reduce_openings_kernel, dim3(nb), dim3(256), 0, (const u64 *)d_proof, lay, C, A, Q, nch, (const u64 *)d_apow, d_y);

    // ---- sum_j alpha^j f_j on the coefficient arrays, then onto the LDE coset
    CombineParams cp = {};
    cp.seg[0] = trace->coeffs; cp.cnt[0] = C; cp.seg[1] = aux ? aux->coeffs : nullptr; cp.cnt[1] = A; cp.seg[2] = quot->coeffs; cp.cnt[2] = Q;
    cp.nch = nch; cp.n = n; cp.apow = d_apow; cp.P = P;
    if (n <= ((size_t)1 << 16)) ZK_LAUNCH(ctx, "fri_combine", fri_combine_split_kernel, dim3((unsigned)((n + 31) / 32)), dim3(256), 0, cp);
    else ZK_LAUNCH(ctx, "fri_combine", fri_combine_kernel, g1(n), dim3(256), 0, cp);
    if (rc == ZK_OK) rc = zk_ntt_lde(ctx, P, n, Pval, L, log_n, 2 * nb, rate_bits, tmp, n);
    if (rc) return rc;
    QuotParams qp = {};
    qp.Pval = Pval; qp.L = L; qp.lgL = lgL; qp.nb = nb;
    TwoLevel xs = zk_scale_table(ctx, gl_root(lgL), GL_GEN, lgL);
    qp.xA = xs.A; qp.xB = xs.B; qp.pts = d_zeta; qp.y = d_y; qp.shifts = d_apow + 2 * cnt; qp.F = F;
    ZK_LAUNCH(ctx, "fri_quotient", fri_quotient_kernel, g1(L), dim3(256), 0, qp);
    own.drop(d_alpha); own.drop(d_apow); own.drop(d_y); own.drop(P); own.drop(Pval); own.drop(tmp);   // reusable by the commit phase

    const zk_batch *oracles[3]; u32 no = 0;
    oracles[no++] = trace; if (A) oracles[no++] = aux; oracles[no++] = quot;
    return zk_fri_proof_dev(ctx, cfg, ch, log_n, F, oracles, no, d_proof, lay);
}

// plonky2::fri::prover::fri_proof on the final values F (re | im columns of L, leaf order): commit phase, PoW, query rounds
int32_t zk_fri_proof_dev(zk_ctx *ctx, const zk_stark_config &cfg, DevChallenger *ch, u32 log_n, u64 *F, const zk_batch *const *oracles,
                         u32 n_oracles, u64 *d_proof, const FriLayout &lay) {
    const u32 rate_bits = cfg.rate_bits, lgL = log_n + rate_bits;
    const size_t L = (size_t)1 << lgL;
    const u32 capw = 4u << cfg.cap_height;
    u32 ar[64], nl; zk_fri_arities(cfg, log_n, ar, &nl);
    if (nl > 8 || n_oracles > 8) return ctx->fail(ZK_ERR_INVALID_ARG, "fri: more than 8 reduction layers or oracles");
    int32_t rc = ZK_OK;
    ZkOwned own(ctx);
    u64 *Fco = (u64 *)own.get(2 * L * sizeof(u64));
    if (!Fco) return ctx->fail(ZK_ERR_OOM, "fri: out of device memory");
    // ---- lde_final_poly coefficients (natural order): coset_ifft(g) of the final values
    {
        TwoLevel post = zk_scale_table(ctx, gl_inv(GL_GEN), gl_inv((u64)L % GL_P), lgL);
        rc = zk_ntt_dit(ctx, F, L, Fco, L, lgL, 2, 1, nullptr, &post, 0, nullptr, 0);
        if (rc) return rc;
    }

    // ---- fri_committed_trees
    ZK_LAUNCH(ctx, "write_word", write_word_kernel, dim3(1), dim3(1), 0, d_proof + lay.o_nl, (u64)nl);
    const u64 *vre[9], *vim[9]; u64 *lvl[8];
    vre[0] = F; vim[0] = F + L;
    u64 *cre = Fco, *cim = Fco + L;          // current coefficients (natural order)
    size_t cur = L; u32 cur_lg = lgL;
    u64 shift = GL_GEN;
    u64 *d_beta = (u64 *)own.get(2 * 8 * sizeof(u64));
    if (!d_beta) return ctx->fail(ZK_ERR_OOM, "fri: out of device memory");
    for (u32 l = 0; l < nl; l++) {
        u32 arity = 1u << ar[l];
        size_t n_leaves = cur >> ar[l];
        u32 depth = cur_lg - ar[l];
        size_t n_dig = zk_level_offset(n_leaves, depth - cfg.cap_height + 1);
        lvl[l] = (u64 *)own.get(n_dig * 4 * sizeof(u64));
        if (!lvl[l]) return ctx->fail(ZK_ERR_OOM, "fri: out of device memory");
        if (!zk_fri_leaf_hash_mfma(ctx, vre[l], vim[l], n_leaves, arity, lvl[l]))
            ZK_LAUNCH(ctx, "fri_leaf_hash", fri_leaf_hash_kernel, g1(n_leaves), dim3(256), 0, vre[l], vim[l], n_leaves, arity, lvl[l], ctx->force_redo);
        zk_merkle_levels(ctx, lvl[l], n_leaves, cfg.cap_height);
        const u64 *cap = lvl[l] + 4 * zk_level_offset(n_leaves, depth - cfg.cap_height);
        u64 *pcap = d_proof + lay.o_layer_caps + (size_t)l * capw;
        ZK_LAUNCH(ctx, "copy_words", copy_words_kernel, g1(capw), dim3(256), 0, cap, pcap, capw);
        { TOpList a = {}; a.n = 2; a.op[0].kind = TOP_OBSERVE; a.op[0].src = pcap; a.op[0].n = capw;
          a.op[1].kind = TOP_GET; a.op[1].dst = d_beta + 2 * l; a.op[1].n = 2; zk_transcript_run(ctx, ch, a); }
        size_t nxt = cur >> ar[l];
        u64 *nco = (u64 *)own.get(2 * nxt * sizeof(u64)), *nva = (u64 *)own.get(2 * nxt * sizeof(u64));
        if (!nco || !nva) return ctx->fail(ZK_ERR_OOM, "fri: out of device memory");
        ZK_LAUNCH(ctx, "fri_fold", fri_fold_kernel, g1(nxt), dim3(256), 0, (const u64 *)cre, (const u64 *)cim, nxt, arity, (const u64 *)(d_beta + 2 * l), nco, nco + nxt);
        cre = nco; cim = nco + nxt;
        shift = gl_pow(shift, arity);
        cur = nxt; cur_lg = depth;
        TwoLevel pre = zk_scale_table(ctx, shift, 1, cur_lg);
        rc = zk_ntt_dif(ctx, nco, cur, nva, cur, cur_lg, 2, 0, &pre, 0, 0);
        if (rc) return rc;
        vre[l + 1] = nva; vim[l + 1] = nva + cur;
    }
    // ---- final polynomial: coeffs.truncate(len >> rate_bits); observe
    u32 flen = (u32)(cur >> rate_bits);
    ZK_LAUNCH(ctx, "write_word", write_word_kernel, dim3(1), dim3(1), 0, d_proof + lay.o_flen, (u64)flen);
    ZK_LAUNCH(ctx, "interleave", interleave_kernel, g1(flen), dim3(256), 0, (const u64 *)cre, (const u64 *)cim, flen, d_proof + lay.o_final);
    { TOpList a = {}; a.n = 1; a.op[0].kind = TOP_OBSERVE; a.op[0].src = d_proof + lay.o_final; a.op[0].n = 2 * flen; zk_transcript_run(ctx, ch, a); }

    // ---- proof of work, query indices
    u64 *d_pow = (u64 *)own.get(16 * sizeof(u64)), *d_idx = (u64 *)own.get((size_t)cfg.num_query_rounds * sizeof(u64));
    if (!d_pow || !d_idx) return ctx->fail(ZK_ERR_OOM, "fri: out of device memory");
    zk_pow_grind_dev(ctx, ch, cfg.proof_of_work_bits, d_pow, d_proof + lay.o_witness);
    { TOpList a = {}; a.n = 1; a.op[0].kind = TOP_GET_INDICES; a.op[0].dst = d_idx; a.op[0].n = cfg.num_query_rounds; a.op[0].aux = lgL; zk_transcript_run(ctx, ch, a); }

    QueryParams q = {};
    q.n_layers = nl; q.cap_height = cfg.cap_height; q.lgL = lgL;
    for (u32 o = 0; o < n_oracles; o++) { q.lde[o] = oracles[o]->lde; q.levels[o] = oracles[o]->levels; q.n_cols[o] = oracles[o]->n_cols; }
    q.n_oracles = n_oracles;
    for (u32 l = 0; l < nl; l++) { q.lre[l] = vre[l]; q.lim[l] = vim[l]; q.llevels[l] = lvl[l]; q.arity_bits[l] = ar[l]; }
    q.idx = d_idx; q.out = d_proof + lay.o_queries; q.per_query = lay.per_query;
    ZK_LAUNCH(ctx, "fri_query_gather", fri_query_kernel, dim3(cfg.num_query_rounds), dim3(256), 0, q);

    return ZK_OK;
}

// ------------------------------------------------------------------ any FriInstanceInfo (zk_fri_prove_instance)
#define FRI_MAX_BATCHES 4
struct GenPows { u32 cnt; u32 sh[FRI_MAX_BATCHES]; };
// alpha^j for j < cnt, then alpha^sh[b] (the ReducingFactor shift applied before batch b is added)
__global__ void alpha_pows_gen_kernel(const u64 *alpha, GenPows g, u64 *out) {
    u32 j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= g.cnt + FRI_MAX_BATCHES) return;
    gl2 a = gl2_make(alpha[0], alpha[1]);
    gl2 r = gl2_pow_u32(a, j < g.cnt ? j : g.sh[j - g.cnt]);
    out[2 * j] = r.a; out[2 * j + 1] = r.b;
}
struct CombineGenParams {
    const u64 *const *cols;            // device array of column pointers, batches back to back
    u32 start[FRI_MAX_BATCHES], cnt[FRI_MAX_BATCHES];
    size_t n;
    const u64 *apow;
    u64 *P;                            // per batch: re column, im column
};
// ReducingFactor::reduce_polys_base, one grid row per batch; coefficient order is whatever the oracles store (bit-reversed)
__global__ __launch_bounds__(256) void fri_combine_gen_kernel(CombineGenParams p) {
    size_t pos = (size_t)blockIdx.x * 256 + threadIdx.x;
    const u32 b = blockIdx.y;
    if (pos >= p.n) return;
    const u64 *const *cols = p.cols + p.start[b];
    u64 ra = 0, rb = 0;
    const u32 cnt = p.cnt[b];
    u32 j = 0;
    for (; j + 4 <= cnt; j += 4) {
        u64 v0 = cols[j][pos], v1 = cols[j + 1][pos], v2 = cols[j + 2][pos], v3 = cols[j + 3][pos];
        ra = gl_add(ra, gl_mul(v0, p.apow[2 * j])); rb = gl_add(rb, gl_mul(v0, p.apow[2 * j + 1]));
        ra = gl_add(ra, gl_mul(v1, p.apow[2 * j + 2])); rb = gl_add(rb, gl_mul(v1, p.apow[2 * j + 3]));
        ra = gl_add(ra, gl_mul(v2, p.apow[2 * j + 4])); rb = gl_add(rb, gl_mul(v2, p.apow[2 * j + 5]));
        ra = gl_add(ra, gl_mul(v3, p.apow[2 * j + 6])); rb = gl_add(rb, gl_mul(v3, p.apow[2 * j + 7]));
    }
    for (; j < cnt; j++) {
        u64 v = cols[j][pos];
        ra = gl_add(ra, gl_mul(v, p.apow[2 * j])); rb = gl_add(rb, gl_mul(v, p.apow[2 * j + 1]));
    }
    p.P[(size_t)(2 * b) * p.n + pos] = ra; p.P[(size_t)(2 * b + 1) * p.n + pos] = rb;
}
// the same for small polynomials: 8 threads share a position, each sums an eighth of the batch's columns, the partial sums meet in LDS
// (2^12 positions are 16 workgroups of the kernel above, every thread walking ~250 dependent column loads)
__global__ __launch_bounds__(256) void fri_combine_gen_split_kernel(CombineGenParams p) {
    __shared__ u64 part[2][8][32];
    const u32 t = threadIdx.x, lp = t & 31, slice = t >> 5, b = blockIdx.y;
    const size_t pos = (size_t)blockIdx.x * 32 + lp;
    const bool live = pos < p.n;
    const u64 *const *cols = p.cols + p.start[b];
    const u32 cnt = p.cnt[b], per = (cnt + 7) / 8, lo = slice * per, hi = lo + per < cnt ? lo + per : cnt;
    u64 ra = 0, rb = 0;
    if (live) {
        for (u32 j = lo; j < hi; j++) {
            const u64 v = cols[j][pos];
            ra = gl_add(ra, gl_mul(v, p.apow[2 * j])); rb = gl_add(rb, gl_mul(v, p.apow[2 * j + 1]));
        }
    }
    part[0][slice][lp] = ra; part[1][slice][lp] = rb;
    __syncthreads();
    if (slice < 2 && live) {
        u64 acc = part[slice][0][lp];
        for (int q = 1; q < 8; q++) acc = gl_add(acc, part[slice][q][lp]);
        p.P[(size_t)(2 * b + slice) * p.n + pos] = acc;
    }
}
// P_b(z_b) from the evaluations of its re and im columns: y = ev_re + X * ev_im, X^2 = 7
__global__ void fri_join_evals_kernel(const u64 *ev, u32 nb, u64 *y) {
    u32 b = threadIdx.x;
    if (b >= nb) return;
    const u64 *e = ev + 4 * b;
    u64 b7 = e[3];
    u64 w = gl_add(gl_add(gl_dbl(gl_dbl(b7)), gl_dbl(b7)), b7);
    y[2 * b] = gl_add(e[0], w); y[2 * b + 1] = gl_add(e[1], e[2]);
}
__global__ void fri_copy_point_kernel(const u64 *src, u64 *dst) { if (threadIdx.x < 2) dst[threadIdx.x] = src[threadIdx.x]; }
struct QuotGenParams {
    const u64 *Pval; size_t L; u32 lgL, nb; u64 *err;
    const u64 *xA, *xB;
    const u64 *pts;           // z_b (2 words each)
    const u64 *y;             // P_b(z_b)
    const u64 *shifts;        // alpha^cnt_b (2 words each)
    u64 *F;
};
// final values F(x) = (..(q_0 a^cnt_1 + q_1) a^cnt_2 + ..)(x),  q_b(x) = (P_b(x) - P_b(z_b)) / (x - z_b); one inversion per point
__global__ __launch_bounds__(256) void fri_quotient_gen_kernel(QuotGenParams p) {
    size_t s = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= p.L) return;
    u32 i = bitrev_u32((u32)s, p.lgL);
    u64 x = gl_mul(p.xA[i & 2047], p.xB[i >> 11]);
    gl2 den[FRI_MAX_BATCHES], num[FRI_MAX_BATCHES], q[FRI_MAX_BATCHES];
    u64 nrm[FRI_MAX_BATCHES], pre[FRI_MAX_BATCHES];
    u64 acc = 1;
#pragma unroll
    for (int b = 0; b < FRI_MAX_BATCHES; b++) {
        if (b < (int)p.nb) {
            den[b] = gl2_make(gl_sub(x, p.pts[2 * b]), gl_neg(p.pts[2 * b + 1]));
            u64 bb = gl_mul(den[b].b, den[b].b);
            u64 w7 = gl_add(gl_add(gl_dbl(gl_dbl(bb)), gl_dbl(bb)), bb);
            nrm[b] = gl_sub(gl_mul(den[b].a, den[b].a), w7);
            num[b] = gl2_make(gl_sub(p.Pval[(size_t)(2 * b) * p.L + s], p.y[2 * b]), gl_sub(p.Pval[(size_t)(2 * b + 1) * p.L + s], p.y[2 * b + 1]));
        } else { nrm[b] = 1; den[b] = gl2_make(1, 0); num[b] = gl2_make(0, 0); }
        pre[b] = acc; acc = gl_mul(acc, nrm[b]);
    }
    if (acc == 0) atomicCAS((unsigned long long *)p.err, 0ULL, (unsigned long long)(uint32_t)ZK_ERR_ZETA_IN_SUBGROUP);   // x == z_b: a point on the LDE coset
    u64 inv = gl_inv_fermat(acc);
#pragma unroll
    for (int b = FRI_MAX_BATCHES - 1; b >= 0; b--) {
        u64 ninv = gl_mul(inv, pre[b]);
        inv = gl_mul(inv, nrm[b]);
        gl2 dinv = gl2_make(gl_mul(den[b].a, ninv), gl_mul(gl_neg(den[b].b), ninv));
        q[b] = gl2_mul(num[b], dinv);
    }
    gl2 F = q[0];
#pragma unroll
    for (int b = 1; b < FRI_MAX_BATCHES; b++)
        if (b < (int)p.nb) F = gl2_add(gl2_mul(F, gl2_make(p.shifts[2 * b], p.shifts[2 * b + 1])), q[b]);
    p.F[s] = F.a; p.F[p.L + s] = F.b;
}

// PolynomialBatch::prove_openings up to (not including) fri_proof: the final values F (re | im columns of L, leaf order) of
//   sum_b alpha^(..) (P_b - P_b(z_b)) / (X - z_b),  P_b = sum_j alpha^j f_j over the batch's polynomials; alpha at d_alpha (device)
int32_t zk_fri_final_values_dev(zk_ctx *ctx, const zk_stark_config &cfg, const u64 *d_alpha, const zk_batch *const *oracles, u32 n_oracles,
                                const FriBatchDesc *batches, u32 nb, u64 *F) {
    const u32 log_n = oracles[0]->log_n, rate_bits = cfg.rate_bits, lgL = log_n + rate_bits;
    const size_t n = (size_t)1 << log_n, L = n << rate_bits;
    if (nb == 0 || nb > FRI_MAX_BATCHES || n_oracles == 0 || n_oracles > 8) return ctx->fail(ZK_ERR_INVALID_ARG, "fri instance: 1..4 batches, 1..8 oracles");
    std::vector<const u64 *> cols;
    CombineGenParams cp = {};
    GenPows gp = {};
    u64 hpts[2 * FRI_MAX_BATCHES] = {};
    for (u32 b = 0; b < nb; b++) {
        if (batches[b].n_polys == 0) return ctx->fail(ZK_ERR_INVALID_ARG, "fri instance: empty batch");
        cp.start[b] = (u32)cols.size(); cp.cnt[b] = batches[b].n_polys;
        for (u32 k = 0; k < batches[b].n_polys; k++) {
            const zk_fri_poly &fp = batches[b].polys[k];
            if (fp.oracle_index >= n_oracles || fp.polynomial_index >= oracles[fp.oracle_index]->n_cols)
                return ctx->fail(ZK_ERR_INVALID_ARG, "fri instance: polynomial reference out of range");
            cols.push_back(oracles[fp.oracle_index]->coeffs + (size_t)fp.polynomial_index * n);
        }
        gp.cnt = std::max(gp.cnt, batches[b].n_polys);
        gp.sh[b] = batches[b].n_polys;
        if (batches[b].d_point) continue;        // device-resident point: copied below, checked by the quotient kernel (zero denominator)
        hpts[2 * b] = batches[b].point[0]; hpts[2 * b + 1] = batches[b].point[1];
        // the pointwise quotient needs x != z on the LDE coset g*H_L: only a base-field z with (z/g)^L == 1 can collide
        if (hpts[2 * b + 1] == 0 && gl_pow(gl_mul(hpts[2 * b], gl_inv(GL_GEN)), (u64)L) == 1)
            return ctx->fail(ZK_ERR_ZETA_IN_SUBGROUP, "fri instance: opening point lies on the LDE coset");
    }
    for (u32 o = 0; o < n_oracles; o++)
        if (oracles[o]->log_n != log_n || oracles[o]->rate_bits != rate_bits || oracles[o]->cap_height != cfg.cap_height)
            return ctx->fail(ZK_ERR_INVALID_ARG, "fri instance: oracles must share log_n and the config's rate_bits / cap_height");
    int32_t rc = ZK_OK;
    ZkOwned own(ctx);
    u64 *d_apow = (u64 *)own.get((size_t)(gp.cnt + FRI_MAX_BATCHES) * 2 * sizeof(u64));
    u64 *d_pts = (u64 *)own.get(2 * FRI_MAX_BATCHES * sizeof(u64)), *d_ev = (u64 *)own.get(4 * FRI_MAX_BATCHES * sizeof(u64));
    u64 *d_y = (u64 *)own.get(2 * FRI_MAX_BATCHES * sizeof(u64));
    const u64 **d_cols = (const u64 **)own.get(cols.size() * sizeof(u64 *));
    u64 *P = (u64 *)own.get(2 * nb * n * sizeof(u64)), *Pval = (u64 *)own.get(2 * nb * L * sizeof(u64)), *tmp = (u64 *)own.get(2 * nb * n * sizeof(u64));
    if (!d_apow || !d_pts || !d_ev || !d_y || !d_cols || !P || !Pval || !tmp) return ctx->fail(ZK_ERR_OOM, "fri instance: out of device memory");
    rc = ctx->upload_small(d_pts, hpts, sizeof hpts);
    if (rc == ZK_OK) rc = ctx->upload_small(d_cols, cols.data(), cols.size() * sizeof(u64 *));
    if (rc) return rc;
    for (u32 b = 0; b < nb; b++)
        if (batches[b].d_point) ZK_LAUNCH(ctx, "fri_copy_point", fri_copy_point_kernel, dim3(1), dim3(64), 0, batches[b].d_point, d_pts + 2 * b);
    ZK_LAUNCH(ctx, "fri_alpha_pows", alpha_pows_gen_kernel, g1(gp.cnt + FRI_MAX_BATCHES), dim3(256), 0, d_alpha, gp, d_apow);
    cp.cols = d_cols; cp.n = n; cp.apow = d_apow; cp.P = P;
    if (n <= ((size_t)1 << 16)) ZK_LAUNCH(ctx, "fri_combine", fri_combine_gen_split_kernel, dim3((unsigned)((n + 31) / 32), nb), dim3(256), 0, cp);
    else ZK_LAUNCH(ctx, "fri_combine", fri_combine_gen_kernel, dim3((unsigned)((n + 255) / 256), nb), dim3(256), 0, cp);
    // P_b(z_b): what divide_by_linear drops as the remainder
    {
        zk_eval_job jobs[FRI_MAX_BATCHES];
        for (u32 b = 0; b < nb; b++) jobs[b] = {P + (size_t)2 * b * n, n, 2u, d_pts + 2 * b, d_ev + 4 * b};
        rc = zk_eval_cols_multi(ctx, (int)log_n, nb, jobs);
    }
    if (rc) return rc;
    ZK_LAUNCH(ctx, "fri_join_evals", fri_join_evals_kernel, dim3(1), dim3(64), 0, (const u64 *)d_ev, nb, d_y);
    if (rc == ZK_OK) rc = zk_ntt_lde(ctx, P, n, Pval, L, log_n, 2 * nb, rate_bits, tmp, n);
    if (rc) return rc;
    QuotGenParams qp = {};
    qp.Pval = Pval; qp.L = L; qp.lgL = lgL; qp.nb = nb; qp.err = ctx->d_err;
    TwoLevel xs = zk_scale_table(ctx, gl_root(lgL), GL_GEN, lgL);
    qp.xA = xs.A; qp.xB = xs.B; qp.pts = d_pts; qp.y = d_y; qp.shifts = d_apow + 2 * gp.cnt; qp.F = F;
    ZK_LAUNCH(ctx, "fri_quotient", fri_quotient_gen_kernel, g1(L), dim3(256), 0, qp);
    return ZK_OK;
}

int32_t zk_fri_prove_instance_dev(zk_ctx *ctx, const zk_stark_config &cfg, DevChallenger *ch, const zk_batch *const *oracles, u32 n_oracles,
                                  const FriBatchDesc *batches, u32 nb, u64 *d_proof, const FriLayout &lay) {
    if (n_oracles == 0 || n_oracles > 8) return ctx->fail(ZK_ERR_INVALID_ARG, "fri instance: 1..8 oracles");
    const u32 log_n = oracles[0]->log_n;
    const size_t L = (size_t)1 << (log_n + cfg.rate_bits);
    ZkOwned own(ctx);
    u64 *d_alpha = (u64 *)own.get(2 * sizeof(u64)), *F = (u64 *)own.get(2 * L * sizeof(u64));
    if (!d_alpha || !F) return ctx->fail(ZK_ERR_OOM, "fri instance: out of device memory");
    { TOpList a = {}; a.n = 1; a.op[0].kind = TOP_GET; a.op[0].dst = d_alpha; a.op[0].n = 2; zk_transcript_run(ctx, ch, a); }
    int32_t rc = zk_fri_final_values_dev(ctx, cfg, d_alpha, oracles, n_oracles, batches, nb, F);
    if (rc) return rc;
    return zk_fri_proof_dev(ctx, cfg, ch, log_n, F, oracles, n_oracles, d_proof, lay);
}

// ------------------------------------------------------------------ finer-grained FRI entry points
// The same steps as zk_fri_prove_instance_dev, one upstream item per call, synchronous, for callers that replace plonky2's FRI
// prover piece by piece: zk_fri_combine (the batch combination of PolynomialBatch::prove_openings), zk_fri_commit_begin /
// zk_fri_fold_commit_layer / zk_fri_commit_finish (fri_committed_trees, one reduction layer per call), zk_fri_query
// (fri_prover_query_round).  The proof of work between finish and the queries is zk_pow_grind.
__global__ void deinterleave_kernel(const u64 *in, size_t n, u64 *re, u64 *im) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { re[i] = gl_canon(in[2 * i]); im[i] = gl_canon(in[2 * i + 1]); }
}
__global__ void interleave_big_kernel(const u64 *re, const u64 *im, size_t n, u64 *out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { out[2 * i] = re[i]; out[2 * i + 1] = im[i]; }
}

struct zk_fri_commit {
    zk_ctx *ctx; zk_stark_config cfg;
    u32 log_n, lgL, ar[64], nl, done; bool finished;
    u64 *F, *Fco;                       // layer-0 values (re | im, leaf order), coefficient buffer
    u64 *cre, *cim; size_t cur; u32 cur_lg; u64 shift;
    const u64 *vre[9], *vim[9]; u64 *lvl[8];
    std::vector<void *> owned;
};

extern "C" {

int32_t zk_fri_combine(zk_ctx *ctx, const zk_stark_config *cfg, uint32_t n_oracles, zk_batch *const *oracles, uint32_t n_batches,
                       const zk_fri_batch *batches, const uint64_t alpha[2], uint64_t *final_values_out) {
    if (!ctx || !cfg || !oracles || !batches || !alpha || !final_values_out || n_oracles == 0 || n_oracles > 8 || n_batches == 0 || n_batches > FRI_MAX_BATCHES)
        return ZK_ERR_INVALID_ARG;
    if (cfg->rate_bits > 4 || cfg->cap_height > 8) return ctx->fail(ZK_ERR_INVALID_ARG, "fri combine: unsupported parameters");
    for (u32 o = 0; o < n_oracles; o++) if (!oracles[o] || oracles[o]->ctx != ctx) return ctx->fail(ZK_ERR_INVALID_ARG, "fri combine: oracle of another context");
    hipSetDevice(ctx->device);
    FriBatchDesc desc[FRI_MAX_BATCHES];
    for (u32 b = 0; b < n_batches; b++) {
        if (!batches[b].polys) return ctx->fail(ZK_ERR_INVALID_ARG, "fri combine: batch without polynomials");
        desc[b].point[0] = gl_canon(batches[b].point[0]); desc[b].point[1] = gl_canon(batches[b].point[1]);
        desc[b].n_polys = batches[b].n_polys; desc[b].polys = batches[b].polys;
    }
    const size_t L = (size_t)1 << (oracles[0]->log_n + cfg->rate_bits);
    ZkOwned own(ctx);
    u64 *d_alpha = (u64 *)own.get(2 * sizeof(u64)), *F = (u64 *)own.get(2 * L * sizeof(u64)), *out = (u64 *)own.get(2 * L * sizeof(u64));
    if (!d_alpha || !F || !out) return ctx->fail(ZK_ERR_OOM, "fri combine: out of device memory");
    u64 ha[2] = {gl_canon(alpha[0]), gl_canon(alpha[1])};
    int32_t rc = zk_dev_upload(ctx, d_alpha, ha, sizeof ha);
    if (rc == ZK_OK) rc = zk_fri_final_values_dev(ctx, *cfg, d_alpha, oracles, n_oracles, desc, n_batches, F);
    if (rc == ZK_OK) {
        ZK_LAUNCH(ctx, "fri_interleave", interleave_big_kernel, g1(L), dim3(256), 0, (const u64 *)F, (const u64 *)(F + L), L, out);
        rc = zk_dev_download(ctx, final_values_out, out, 2 * L * sizeof(u64));
    }
    return rc;
}

int32_t zk_fri_commit_begin(zk_ctx *ctx, const zk_stark_config *cfg, uint32_t log_n, const uint64_t *final_values, zk_fri_commit **out) {
    if (!ctx || !cfg || !final_values || !out) return ZK_ERR_INVALID_ARG;
    *out = nullptr;
    if (cfg->rate_bits > 4 || cfg->cap_height > 8 || cfg->fri_arity_bits < 2 || cfg->fri_arity_bits > 6 || log_n > 28 || log_n + cfg->rate_bits < cfg->cap_height)
        return ctx->fail(ZK_ERR_INVALID_ARG, "fri commit: unsupported FRI parameters");
    if (!zk_fri_reduction_ok(*cfg, log_n)) return ctx->fail(ZK_ERR_INVALID_ARG, "fri commit: the reduction strategy does not fit this size (upstream panics)");
    hipSetDevice(ctx->device);
    zk_fri_commit *s = new zk_fri_commit();
    s->ctx = ctx; s->cfg = *cfg; s->log_n = log_n; s->lgL = log_n + cfg->rate_bits; s->done = 0; s->finished = false;
    zk_fri_arities(*cfg, log_n, s->ar, &s->nl);
    const size_t L = (size_t)1 << s->lgL;
    if (s->nl > 8) { delete s; return ctx->fail(ZK_ERR_INVALID_ARG, "fri commit: more than 8 reduction layers"); }
    s->F = (u64 *)ctx->alloc(2 * L * sizeof(u64)); s->Fco = (u64 *)ctx->alloc(2 * L * sizeof(u64));
    u64 *tmp = (u64 *)ctx->alloc(2 * L * sizeof(u64));
    if (s->F) s->owned.push_back(s->F);
    if (s->Fco) s->owned.push_back(s->Fco);
    int32_t rc = (s->F && s->Fco && tmp) ? zk_dev_upload(ctx, tmp, final_values, 2 * L * sizeof(u64)) : ZK_ERR_OOM;
    if (rc == ZK_OK) {
        ZK_LAUNCH(ctx, "fri_deinterleave", deinterleave_kernel, g1(L), dim3(256), 0, (const u64 *)tmp, L, s->F, s->F + L);
        // coefficients of the final polynomial on the LDE domain: coset_ifft(g) of the values (leaf order -> natural coefficients)
        TwoLevel post = zk_scale_table(ctx, gl_inv(GL_GEN), gl_inv((u64)L % GL_P), s->lgL);
        rc = zk_ntt_dit(ctx, s->F, L, s->Fco, L, s->lgL, 2, 1, nullptr, &post, 0, nullptr, 0);
    }
    ctx->release(tmp);
    if (rc != ZK_OK) { for (void *q : s->owned) ctx->release(q); delete s; return rc; }
    s->vre[0] = s->F; s->vim[0] = s->F + L; s->cre = s->Fco; s->cim = s->Fco + L; s->cur = L; s->cur_lg = s->lgL; s->shift = GL_GEN;
    *out = s;
    return ZK_OK;
}

int32_t zk_fri_commit_layers(const zk_fri_commit *s, uint32_t *n_layers, uint32_t *layers_done) {
    if (!s) return ZK_ERR_INVALID_ARG;
    if (n_layers) *n_layers = s->nl;
    if (layers_done) *layers_done = s->done;
    return ZK_OK;
}

int32_t zk_fri_fold_commit_layer(zk_fri_commit *s, zk_challenger *ch, uint64_t *cap_out, uint64_t beta_out[2]) {
    if (!s || !ch || !cap_out || !beta_out || ch->ctx != s->ctx) return ZK_ERR_INVALID_ARG;
    zk_ctx *ctx = s->ctx;
    if (s->done >= s->nl || s->finished) return ctx->fail(ZK_ERR_INVALID_ARG, "fri commit: every reduction layer has been committed");
    hipSetDevice(ctx->device);
    const zk_stark_config &cfg = s->cfg;
    const u32 l = s->done, arity = 1u << s->ar[l], capw = 4u << cfg.cap_height;
    const size_t n_leaves = s->cur >> s->ar[l];
    const u32 depth = s->cur_lg - s->ar[l];
    const size_t n_dig = zk_level_offset(n_leaves, depth - cfg.cap_height + 1);
    s->lvl[l] = (u64 *)ctx->alloc(n_dig * 4 * sizeof(u64));
    u64 *d_beta = (u64 *)ctx->alloc(2 * sizeof(u64));
    const size_t nxt = s->cur >> s->ar[l];
    u64 *nco = (u64 *)ctx->alloc(2 * nxt * sizeof(u64)), *nva = (u64 *)ctx->alloc(2 * nxt * sizeof(u64));
    for (void *q : {(void *)s->lvl[l], (void *)nco, (void *)nva}) if (q) s->owned.push_back(q);
    if (!s->lvl[l] || !d_beta || !nco || !nva) { ctx->release(d_beta); return ctx->fail(ZK_ERR_OOM, "fri commit: out of device memory"); }
    if (!zk_fri_leaf_hash_mfma(ctx, s->vre[l], s->vim[l], n_leaves, arity, s->lvl[l]))
        ZK_LAUNCH(ctx, "fri_leaf_hash", fri_leaf_hash_kernel, g1(n_leaves), dim3(256), 0, s->vre[l], s->vim[l], n_leaves, arity, s->lvl[l], ctx->force_redo);
    zk_merkle_levels(ctx, s->lvl[l], n_leaves, cfg.cap_height);
    const u64 *cap = s->lvl[l] + 4 * zk_level_offset(n_leaves, depth - cfg.cap_height);
    { TOpList a = {}; a.n = 2; a.op[0].kind = TOP_OBSERVE; a.op[0].src = cap; a.op[0].n = capw;
      a.op[1].kind = TOP_GET; a.op[1].dst = d_beta; a.op[1].n = 2; zk_transcript_run(ctx, ch->d, a); }
    ZK_LAUNCH(ctx, "fri_fold", fri_fold_kernel, g1(nxt), dim3(256), 0, (const u64 *)s->cre, (const u64 *)s->cim, nxt, arity, (const u64 *)d_beta, nco, nco + nxt);
    s->cre = nco; s->cim = nco + nxt;
    s->shift = gl_pow(s->shift, arity);
    s->cur = nxt; s->cur_lg = depth;
    TwoLevel pre = zk_scale_table(ctx, s->shift, 1, s->cur_lg);
    int32_t rc = zk_ntt_dif(ctx, nco, s->cur, nva, s->cur, s->cur_lg, 2, 0, &pre, 0, 0);
    s->vre[l + 1] = nva; s->vim[l + 1] = nva + s->cur;
    if (rc == ZK_OK) rc = zk_dev_download(ctx, cap_out, cap, capw * sizeof(u64));
    if (rc == ZK_OK) rc = zk_dev_download(ctx, beta_out, d_beta, 2 * sizeof(u64));
    ctx->release(d_beta);
    if (rc == ZK_OK) s->done++;
    return rc;
}

int32_t zk_fri_commit_finish(zk_fri_commit *s, zk_challenger *ch, uint64_t *final_poly_out, uint32_t *len_ext) {
    if (!s || !ch || !final_poly_out || ch->ctx != s->ctx) return ZK_ERR_INVALID_ARG;
    zk_ctx *ctx = s->ctx;
    if (s->done != s->nl || s->finished) return ctx->fail(ZK_ERR_INVALID_ARG, "fri commit: layers outstanding, or already finished");
    hipSetDevice(ctx->device);
    const u32 flen = (u32)(s->cur >> s->cfg.rate_bits);          // coeffs.truncate(len >> rate_bits)
    u64 *d = (u64 *)ctx->alloc(2 * (size_t)flen * sizeof(u64));
    if (!d) return ctx->fail(ZK_ERR_OOM, "fri commit: out of device memory");
    ZK_LAUNCH(ctx, "interleave", interleave_kernel, g1(flen), dim3(256), 0, (const u64 *)s->cre, (const u64 *)s->cim, flen, d);
    { TOpList a = {}; a.n = 1; a.op[0].kind = TOP_OBSERVE; a.op[0].src = d; a.op[0].n = 2 * flen; zk_transcript_run(ctx, ch->d, a); }
    int32_t rc = zk_dev_download(ctx, final_poly_out, d, 2 * (size_t)flen * sizeof(u64));
    ctx->release(d);
    if (rc == ZK_OK) { s->finished = true; if (len_ext) *len_ext = flen; }
    return rc;
}

size_t zk_fri_query_size_words(const zk_fri_commit *s, uint32_t n_oracles, zk_batch *const *oracles) {
    if (!s || !oracles || n_oracles == 0 || n_oracles > 8) return 0;
    size_t per = 0;
    for (u32 o = 0; o < n_oracles; o++) { if (!oracles[o]) return 0; per += oracles[o]->n_cols + (size_t)4 * (s->lgL - s->cfg.cap_height); }
    u32 lg = s->lgL;
    for (u32 l = 0; l < s->nl; l++) { lg -= s->ar[l]; per += (2u << s->ar[l]) + 4 * (size_t)(lg - s->cfg.cap_height); }
    return per;
}

int32_t zk_fri_query(zk_fri_commit *s, uint32_t n_oracles, zk_batch *const *oracles, uint32_t x_index, uint64_t *out, size_t cap_words, size_t *len_words) {
    if (!s || !oracles || !out || n_oracles == 0 || n_oracles > 8) return ZK_ERR_INVALID_ARG;
    zk_ctx *ctx = s->ctx;
    if (s->done != s->nl) return ctx->fail(ZK_ERR_INVALID_ARG, "fri query: commit phase not complete");
    const size_t L = (size_t)1 << s->lgL;
    if (x_index >= L) return ctx->fail(ZK_ERR_INVALID_ARG, "fri query: index outside the LDE domain");
    for (u32 o = 0; o < n_oracles; o++)
        if (!oracles[o] || oracles[o]->ctx != ctx || oracles[o]->log_n != s->log_n || oracles[o]->rate_bits != s->cfg.rate_bits || oracles[o]->cap_height != s->cfg.cap_height)
            return ctx->fail(ZK_ERR_INVALID_ARG, "fri query: oracle does not match the commit phase");
    const size_t per = zk_fri_query_size_words(s, n_oracles, oracles);
    if (len_words) *len_words = per;
    if (cap_words < per) return ZK_ERR_BUFFER_TOO_SMALL;
    hipSetDevice(ctx->device);
    ZkOwned own(ctx);
    u64 *d_idx = (u64 *)own.get(sizeof(u64)), *d_out = (u64 *)own.get(per * sizeof(u64));
    if (!d_idx || !d_out) return ctx->fail(ZK_ERR_OOM, "fri query: out of device memory");
    u64 hx = x_index;
    int32_t rc = zk_dev_upload(ctx, d_idx, &hx, sizeof hx);
    if (rc) return rc;
    QueryParams q = {};
    q.n_layers = s->nl; q.cap_height = s->cfg.cap_height; q.lgL = s->lgL; q.n_oracles = n_oracles;
    for (u32 o = 0; o < n_oracles; o++) { q.lde[o] = oracles[o]->lde; q.levels[o] = oracles[o]->levels; q.n_cols[o] = oracles[o]->n_cols; }
    for (u32 l = 0; l < s->nl; l++) { q.lre[l] = s->vre[l]; q.lim[l] = s->vim[l]; q.llevels[l] = s->lvl[l]; q.arity_bits[l] = s->ar[l]; }
    q.idx = d_idx; q.out = d_out; q.per_query = per;
    ZK_LAUNCH(ctx, "fri_query_gather", fri_query_kernel, dim3(1), dim3(256), 0, q);
    return zk_dev_download(ctx, out, d_out, per * sizeof(u64));
}

int32_t zk_fri_commit_free(zk_fri_commit *s) {
    if (!s) return ZK_ERR_INVALID_ARG;
    hipStreamSynchronize(s->ctx->stream);
    for (void *q : s->owned) s->ctx->release(q);
    delete s;
    return ZK_OK;
}

}  // extern "C"
