// zk_prove_plonk: a Plonk-shaped proof on the device.
// Mirrors plonky2 0.2.2 plonk::prover::prove_with_partition_witness -- wires commitment, permutation partial products and Z
// (plonk::prover::wires_permutation_partial_products_and_zs, plonk::plonk_common::{quotient_chunk_products, partial_products_and_z_gx}),
// vanishing polynomial on the 8n coset (plonk::vanishing_poly::eval_vanishing_poly_base_batch, check_partial_products,
// reduce_with_powers), quotient chunks, OpeningSet, PolynomialBatch::prove_openings over the four oracles -- the work the reference
// does behind /root/reference/ops/src/lib.rs:52 (after the STARK proofs), :72 (AggProof::combine) and :95 (BlockProof::execute)
// with CircuitConfig::standard_recursion_config() (crate pinned at /root/reference/Cargo.lock:3441-3443).  The circuit is plain data
// (include/zkgl.h: zk_plonk_circuit); everything stays on the device: one host synchronisation per proof (the final download).
#include <string.h>
#include "zk_internal.h"
#include "prover.h"
#include "poseidon_wave.cuh"
#include "poseidon.cuh"

#define PLK_MAX_CHUNKS 16

// PoseidonHash::hash_no_pad of a short vector (public inputs, digest inputs): always the sponge, never hash_or_noop's copy branch.
// One wave, the wave-wide permutation (poseidon_wave.cuh): word l of the state in lane l -- 7.5 us per permutation where one thread
// running the scalar permutation took 45 (it pays the same per-instruction cost as a full wave and has four times the instructions).
__global__ __launch_bounds__(64) void plonk_hash_no_pad_kernel(const u64 *in, u32 n, u64 *out) {
    __shared__ u64 wtab[psdw::TAB_WORDS];
    psdw::fill<64>(wtab);
    __syncthreads();
    const int lane = threadIdx.x;
    u64 s = 0;
    for (u32 c = 0; c < n; c += 8) {
        if (lane < 8 && c + (u32)lane < n) s = gl_canon(in[c + lane]);      // overwrite mode: the rate words present in this chunk
        s = psdw::permute(s, lane, wtab, 0);
    }
    if (lane < 4) out[lane] = s;
}

struct zk_plonk_state {
    zk_ctx *ctx;
    zk_plonk_circuit c;                       // constants / sigmas / gates pointers are NOT kept (host arrays of the caller)
    zk_stark_config cfg;
    zk_batch *cs;                             // constants | sigmas
    zk_instr *d_instrs; u64 *d_gconst; u64 *d_kis; u64 *d_digest;
    u32 n_instrs, n_regs, n_gconst, n_gate_terms;
    u64 digest[4];
    std::vector<const void *> gates_fns; int gates_fn_is_module = 0; u32 gates_n_state = 0;     // the gate program compiled to gfx950 code (zk_plonk_gates_kernel), or none
};

// number of gate constraint terms (zkgl.h): ZK_OP_EMIT writes term (number of EMITs before it), ZK_OP_EMIT_SLOT term `dst`
static u32 plonk_gate_terms(const zk_air *g) {
    u32 c = 0, hi = 0;
    for (u32 p = 0; p < g->n_instrs; p++) {
        if (g->instrs[p].op == ZK_OP_EMIT) c++;
        else if (g->instrs[p].op == ZK_OP_EMIT_SLOT && g->instrs[p].dst + 1 > hi) hi = g->instrs[p].dst + 1;
    }
    return c > hi ? c : hi;
}
static int32_t plonk_circuit_check(const zk_stark_config *cfg, const zk_plonk_circuit *c) {
    if (!cfg || !c || !c->gates || !c->constants || !c->sigmas) return ZK_ERR_INVALID_ARG;
    if (c->degree_bits < 2 || c->degree_bits > 24 || c->num_wires == 0 || c->num_wires > 4096 || c->num_routed_wires == 0 || c->num_routed_wires > c->num_wires ||
        c->num_constants == 0 || c->num_constants > 4096)
        return ZK_ERR_INVALID_ARG;
    if (cfg->rate_bits == 0 || cfg->rate_bits > 3 || c->quotient_degree_factor != (1u << cfg->rate_bits) || c->num_routed_wires % c->quotient_degree_factor)
        return ZK_ERR_INVALID_ARG;
    if (c->n_domain_separator > 4096 || (c->n_domain_separator && !c->domain_separator)) return ZK_ERR_INVALID_ARG;
    if (cfg->num_challenges == 0 || cfg->num_challenges > 4 || cfg->cap_height > 8 || cfg->fri_arity_bits < 2 || cfg->fri_arity_bits > 6 || cfg->num_query_rounds == 0 ||
        cfg->proof_of_work_bits > 34 || c->degree_bits + cfg->rate_bits < cfg->cap_height || !zk_fri_reduction_ok(*cfg, c->degree_bits))
        return ZK_ERR_INVALID_ARG;
    const zk_air *g = c->gates;
    if ((g->n_instrs && !g->instrs) || (g->n_constants && !g->constants) || g->n_regs == 0 || g->n_regs > 32 || g->n_instrs > (1u << 20)) return ZK_ERR_INVALID_ARG;
    for (u32 p = 0; p < g->n_instrs; p++) {
        const zk_instr &in = g->instrs[p];
        if (in.op > ZK_OP_EMIT && in.op != ZK_OP_EMIT_SLOT) return ZK_ERR_INVALID_ARG;
        if (in.op == ZK_OP_EMIT_SLOT && in.dst >= 4096) return ZK_ERR_INVALID_ARG;
        const u32 ops[2] = {in.a, in.b};
        for (int s = 0; s < (in.op <= ZK_OP_MUL ? 2 : 1); s++) {
            const u32 k = ops[s] >> 24, i = ops[s] & 0xFFFFFFu;
            const bool ok = k == ZK_OPND_REG ? i < g->n_regs : k == ZK_OPND_LOCAL ? i < c->num_wires + c->num_constants : k == ZK_OPND_CONST ? i < g->n_constants : k == ZK_OPND_PUBLIC ? i < 4 : false;
            if (!ok) return ZK_ERR_INVALID_ARG;
        }
        if (in.op <= ZK_OP_MUL && ((in.dst >> 24) || (in.dst & 0xFFFFFFu) >= g->n_regs)) return ZK_ERR_INVALID_ARG;
    }
    if (plonk_gate_terms(g) > 4096) return ZK_ERR_INVALID_ARG;
    return ZK_OK;
}

// ---- layout of the proof words
struct PlonkLayout { size_t o_wcap, o_zcap, o_qcap, o_cs, o_w, o_z, o_zn, o_pp, o_q, o_fri, total; FriLayout fri; };
static PlonkLayout plonk_layout(const zk_stark_config &cfg, const zk_plonk_circuit &c) {
    PlonkLayout y = {};
    const u32 nch = cfg.num_challenges, qdf = c.quotient_degree_factor, npp = c.num_routed_wires / qdf - 1, capw = 4u << cfg.cap_height;
    const u32 NC = c.num_constants, R = c.num_routed_wires, W = c.num_wires, k = c.degree_bits, lgL = k + cfg.rate_bits;
    size_t o = 5;
    y.o_wcap = o; o += capw; y.o_zcap = o; o += capw; y.o_qcap = o; o += capw;
    y.o_cs = o; o += 2 * (size_t)(NC + R); y.o_w = o; o += 2 * (size_t)W; y.o_z = o; o += 2 * (size_t)nch; y.o_zn = o; o += 2 * (size_t)nch;
    y.o_pp = o; o += 2 * (size_t)nch * npp; y.o_q = o; o += 2 * (size_t)nch * qdf;
    y.o_fri = o;
    u32 ar[64], nl; zk_fri_arities(cfg, k, ar, &nl);
    u32 d = k; for (u32 l = 0; l < nl; l++) d -= ar[l];
    size_t f = 0;
    y.fri.o_nl = f; f += 1; y.fri.o_layer_caps = f; f += (size_t)nl * capw; y.fri.o_flen = f; f += 1; y.fri.o_final = f; f += 2 * ((size_t)1 << d);
    y.fri.o_witness = f; f += 1; y.fri.o_queries = f;
    size_t per_q = (size_t)(NC + R) + W + (size_t)nch * (1 + npp) + (size_t)nch * qdf + (size_t)4 * 4 * (lgL - cfg.cap_height);
    u32 lg = lgL;
    for (u32 l = 0; l < nl; l++) { lg -= ar[l]; per_q += (2u << ar[l]) + 4 * (size_t)(lg - cfg.cap_height); }
    y.fri.per_query = per_q;
    f += per_q * cfg.num_query_rounds;
    y.total = o + f;
    return y;
}

extern "C" size_t zk_plonk_proof_size_words(const zk_stark_config *cfg, const zk_plonk_circuit *c) {
    if (plonk_circuit_check(cfg, c) != ZK_OK) return 0;
    return plonk_layout(*cfg, *c).total;
}

// ---- kernels
__device__ __forceinline__ u64 plk_inv(u64 a) {          // a^(p-2), square-and-multiply over the fixed exponent
    u64 r = 1, b = a;
    u64 e = GL_P - 2;
    for (int i = 0; i < 64; i++) { if (e & 1) r = gl_mul(r, b); b = gl_mul(b, b); e >>= 1; }
    return r;
}
struct PermParams {
    const u64 *wires; const u64 *sigmas; size_t n;      // values on H, column-major (sigmas: the last R columns of the constants|sigmas input)
    const u64 *kis; const u64 *bg;                       // k_j; betas[nch] | gammas[nch]
    const u64 *xA, *xB;                                  // w_n^r two-level table
    u32 R, qdf, nchunks, nch;
    u64 *q;                                              // [challenge][chunk][row]: prod over chunks <= c of (prod num / prod den)
    u64 *err;
};
// quotient_chunk_products + the within-row running product (partial_products_and_z_gx without the Z(x) factor): one thread = one row
__global__ __launch_bounds__(256) void plonk_perm_rows_kernel(PermParams p) {
    const size_t r = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= p.n) return;
    const u64 x = gl_mul(p.xA[r & 2047], p.xB[r >> 11]);
    for (u32 ci = 0; ci < p.nch; ci++) {
        const u64 beta = p.bg[ci], gamma = p.bg[p.nch + ci];
        u64 N[PLK_MAX_CHUNKS], D[PLK_MAX_CHUNKS];
#pragma unroll
        for (int c = 0; c < PLK_MAX_CHUNKS; c++) { N[c] = 1; D[c] = 1; }
#pragma unroll
        for (int c = 0; c < PLK_MAX_CHUNKS; c++) {
            if ((u32)c < p.nchunks) {
                u64 num = 1, den = 1;
                for (u32 t = 0; t < p.qdf; t++) {
                    const u32 j = (u32)c * p.qdf + t;
                    const u64 w = p.wires[(size_t)j * p.n + r];
                    num = gl_mul(num, gl_add(gl_add(w, gl_mul(beta, gl_mul(p.kis[j], x))), gamma));
                    den = gl_mul(den, gl_add(gl_add(w, gl_mul(beta, p.sigmas[(size_t)j * p.n + r])), gamma));
                }
                N[c] = num; D[c] = den;
            }
        }
        // prefix products of the numerators; 1 / (prefix product of the denominators) from ONE inversion of the full product
        u64 dall = 1;
#pragma unroll
        for (int c = 0; c < PLK_MAX_CHUNKS; c++) dall = gl_mul(dall, D[c]);
        if (dall == 0) { atomicCAS((unsigned long long *)p.err, 0ULL, (unsigned long long)(uint32_t)ZK_ERR_ZERO_INVERSE); }
        u64 dinv = plk_inv(dall);                        // = 1 / (D_0 .. D_{last})
        u64 prefN[PLK_MAX_CHUNKS];
        u64 a = 1;
#pragma unroll
        for (int c = 0; c < PLK_MAX_CHUNKS; c++) { a = gl_mul(a, N[c]); prefN[c] = a; }
        // walking down: dinv_c = 1 / (D_0 .. D_c) = dinv_{c+1} * D_{c+1}
#pragma unroll
        for (int c = PLK_MAX_CHUNKS - 1; c >= 0; c--) {
            if ((u32)c < p.nchunks) p.q[((size_t)ci * p.nchunks + c) * p.n + r] = gl_mul(prefN[c], dinv);
            dinv = gl_mul(dinv, D[c]);
        }
    }
}
// The same with the chunks of a row spread over the workgroup: 32 rows x 8 slots, slot s takes the (challenge, chunk) pairs s, s + 8, ...
// (a 2^12-row circuit is 16 workgroups of the row-per-thread kernel, each thread walking 2 x 80 dependent loads: 146 us of pure
// latency; here a thread walks at most 3 chunks of 8 wires and one wave per workgroup does the 64 inversions: same values, ~25 us).
// LDS: numerator and denominator products [pair][row].  nch * nchunks <= 64.
__global__ __launch_bounds__(256) void plonk_perm_rows_split_kernel(PermParams p) {
    __shared__ u64 shN[64][32], shD[64][32];
    const u32 rl = threadIdx.x & 31, slot = threadIdx.x >> 5;
    const size_t r = (size_t)blockIdx.x * 32 + rl;
    const bool live = r < p.n;
    const u32 G = p.nch * p.nchunks;
    if (live) {
        const u64 x = gl_mul(p.xA[r & 2047], p.xB[r >> 11]);
        for (u32 g = slot; g < G; g += 8) {
            const u32 ci = g / p.nchunks, c = g % p.nchunks;
            const u64 beta = p.bg[ci], gamma = p.bg[p.nch + ci];
            u64 num = 1, den = 1;
            for (u32 t = 0; t < p.qdf; t++) {
                const u32 j = c * p.qdf + t;
                const u64 w = p.wires[(size_t)j * p.n + r];
                num = gl_mul(num, gl_add(gl_add(w, gl_mul(beta, gl_mul(p.kis[j], x))), gamma));
                den = gl_mul(den, gl_add(gl_add(w, gl_mul(beta, p.sigmas[(size_t)j * p.n + r])), gamma));
            }
            shN[g][rl] = num; shD[g][rl] = den;
        }
    }
    __syncthreads();
    if (!live || slot >= p.nch) return;
    const u32 ci = slot, g0 = ci * p.nchunks;
    u64 dall = 1;
    for (u32 c = 0; c < p.nchunks; c++) dall = gl_mul(dall, shD[g0 + c][rl]);
    if (dall == 0) { atomicCAS((unsigned long long *)p.err, 0ULL, (unsigned long long)(uint32_t)ZK_ERR_ZERO_INVERSE); }
    u64 dinv = plk_inv(dall);                            // = 1 / (D_0 .. D_{last})
    u64 a = 1;
    for (u32 c = 0; c < p.nchunks; c++) { a = gl_mul(a, shN[g0 + c][rl]); shN[g0 + c][rl] = a; }       // prefix numerators (this thread's own cells)
    for (u32 c = p.nchunks; c-- > 0;) {
        p.q[((size_t)ci * p.nchunks + c) * p.n + r] = gl_mul(shN[g0 + c][rl], dinv);
        dinv = gl_mul(dinv, shD[g0 + c][rl]);
    }
}
// the same for any number of chunks (more than PLK_MAX_CHUNKS: routed-wire counts beyond 16 x quotient_degree_factor): prefix numerators
// go to q, the chunk denominators to a scratch column block `den` [chunk][row], both re-read on the way down
__global__ __launch_bounds__(256) void plonk_perm_rows_wide_kernel(PermParams p, u64 *den) {
    const size_t r = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= p.n) return;
    const u64 x = gl_mul(p.xA[r & 2047], p.xB[r >> 11]);
    for (u32 ci = 0; ci < p.nch; ci++) {
        const u64 beta = p.bg[ci], gamma = p.bg[p.nch + ci];
        u64 pn = 1, dall = 1;
        for (u32 c = 0; c < p.nchunks; c++) {
            u64 num = 1, d = 1;
            for (u32 t = 0; t < p.qdf; t++) {
                const u32 j = c * p.qdf + t;
                const u64 w = p.wires[(size_t)j * p.n + r];
                num = gl_mul(num, gl_add(gl_add(w, gl_mul(beta, gl_mul(p.kis[j], x))), gamma));
                d = gl_mul(d, gl_add(gl_add(w, gl_mul(beta, p.sigmas[(size_t)j * p.n + r])), gamma));
            }
            pn = gl_mul(pn, num); dall = gl_mul(dall, d);
            p.q[((size_t)ci * p.nchunks + c) * p.n + r] = pn;
            den[(size_t)c * p.n + r] = d;
        }
        if (dall == 0) { atomicCAS((unsigned long long *)p.err, 0ULL, (unsigned long long)(uint32_t)ZK_ERR_ZERO_INVERSE); }
        u64 dinv = plk_inv(dall);
        for (u32 c = p.nchunks; c-- > 0;) {
            const size_t o = ((size_t)ci * p.nchunks + c) * p.n + r;
            p.q[o] = gl_mul(p.q[o], dinv);
            dinv = gl_mul(dinv, den[(size_t)c * p.n + r]);
        }
    }
}
// Z(w^r) = prod_{s < r} rowprod(s): exclusive prefix product over the rows, one workgroup per challenge
__global__ __launch_bounds__(1024) void plonk_prefix_product_kernel(const u64 *q, size_t n, u32 nchunks, u64 *z_out /* [challenge][row] */) {
    __shared__ u64 sh[1024];
    const u32 ci = blockIdx.x, t = threadIdx.x;
    const u64 *rowprod = q + ((size_t)ci * nchunks + (nchunks - 1)) * n;
    u64 *z = z_out + (size_t)ci * n;
    const size_t per = (n + 1023) / 1024, lo = (size_t)t * per, hi = lo + per < n ? lo + per : n;
    u64 a = 1;
    for (size_t r = lo; r < hi; r++) a = gl_mul(a, rowprod[r]);
    sh[t] = a;
    __syncthreads();
    for (u32 s = 1; s < 1024; s <<= 1) {                 // inclusive scan of the segment products
        u64 v = sh[t];
        if (t >= s) v = gl_mul(sh[t - s], v);
        __syncthreads();
        sh[t] = v;
        __syncthreads();
    }
    u64 run = t ? sh[t - 1] : 1;
    for (size_t r = lo; r < hi; r++) { z[r] = run; run = gl_mul(run, rowprod[r]); }
}
// zs_partial_products columns: [Z_0 .. Z_{nch-1}] then, per challenge, partial product c = Z * q_c  (c < nchunks - 1)
__global__ __launch_bounds__(256) void plonk_partial_products_kernel(const u64 *q, size_t n, u32 nchunks, u32 nch, u64 *zpp) {
    const size_t r = (size_t)blockIdx.x * 256 + threadIdx.x;
    const u32 ci = blockIdx.y;
    if (r >= n) return;
    const u64 z = zpp[(size_t)ci * n + r];
    const u32 npp = nchunks - 1;
    for (u32 c = 0; c < npp; c++) zpp[((size_t)nch + (size_t)ci * npp + c) * n + r] = gl_mul(z, q[((size_t)ci * nchunks + c) * n + r]);
}

// alpha_j^t for t < NT (reduce_with_powers): thread = (challenge, t)
__global__ void plonk_alpha_pows_kernel(const u64 *alphas, u32 nch, u32 NT, u64 *out) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nch * NT) return;
    out[i] = gl_pow(alphas[i / NT], (u64)(i % NT));
}
// eval_vanishing_poly_base_batch at every point of the coset: terms [L_0 (Z_i - 1)] [partial-product checks] [gate constraints],
// reduce_with_powers = sum_t term_t alpha_j^t, times 1 / Z_H.  Gate constraints of the same term index add up (every gate is evaluated at
// every point behind its selector filter), so the sum over terms is taken in whatever order the program emits them: the powers come
// from a table.  One thread = one coset point; gate registers live in LDS.
__global__ __launch_bounds__(256) void plonk_quotient_kernel(PlkQuotParams p) {
    extern __shared__ u64 regs[];           // [reg][thread]
    const size_t s = (size_t)blockIdx.x * 256 + threadIdx.x;
    const u32 t = threadIdx.x;
    if (s >= p.L) return;
    const u32 i = bitrev_u32((u32)s, p.lgL);
    const size_t sn = bitrev_u32((u32)((i + p.qdf) & (p.L - 1)), p.lgL);
    const u64 x = gl_mul(p.xA[i & 2047], p.xB[i >> 11]);
    u64 acc[4] = {0, 0, 0, 0};
    auto emit = [&](u32 term_index, u64 term) {
#pragma unroll
        for (int j = 0; j < 4; j++) if ((u32)j < p.nch) acc[j] = gl_add(acc[j], gl_mul(term, p.apow[(size_t)j * p.NT + term_index]));
    };
    const u32 npp = p.nchunks - 1;
    const u64 l0 = p.l0[s];
    u32 ti = 0;
    for (u32 ci = 0; ci < p.nch; ci++) emit(ti++, gl_mul(l0, gl_sub(p.z_lde[(size_t)ci * p.L + s], 1)));
    for (u32 ci = 0; ci < p.nch; ci++) {
        const u64 beta = p.bg[ci], gamma = p.bg[p.nch + ci];
        for (u32 c = 0; c < p.nchunks; c++) {
            u64 num = 1, den = 1;
            for (u32 u = 0; u < p.qdf; u++) {
                const u32 j = c * p.qdf + u;
                const u64 w = p.w_lde[(size_t)j * p.L + s];
                num = gl_mul(num, gl_add(gl_add(w, gl_mul(beta, gl_mul(p.kis[j], x))), gamma));
                den = gl_mul(den, gl_add(gl_add(w, gl_mul(beta, p.cs_lde[(size_t)(p.NC + j) * p.L + s])), gamma));
            }
            const u64 prev = c == 0 ? p.z_lde[(size_t)ci * p.L + s] : p.z_lde[((size_t)p.nch + (size_t)ci * npp + c - 1) * p.L + s];
            const u64 next = c == p.nchunks - 1 ? p.z_lde[(size_t)ci * p.L + sn] : p.z_lde[((size_t)p.nch + (size_t)ci * npp + c) * p.L + s];
            emit(ti++, gl_sub(gl_mul(prev, num), gl_mul(next, den)));
        }
    }
    auto fetch = [&](u32 o) -> u64 {
        const u32 k = o >> 24, idx = o & 0xFFFFFFu;
        if (k == ZK_OPND_REG) return regs[idx * 256 + t];
        if (k == ZK_OPND_LOCAL) return idx < p.W ? p.w_lde[(size_t)idx * p.L + s] : p.cs_lde[(size_t)(idx - p.W) * p.L + s];
        if (k == ZK_OPND_PUBLIC) return p.pih[idx];
        return p.gconst[idx];
    };
    for (u32 pc = 0; pc < p.n_instrs; pc++) {
        const zk_instr in = p.instrs[pc];
        const u64 a = fetch(in.a);
        if (in.op <= ZK_OP_MUL) {
            const u64 b = fetch(in.b);
            regs[(in.dst & 0xFFFFFFu) * 256 + t] = in.op == ZK_OP_ADD ? gl_add(a, b) : in.op == ZK_OP_SUB ? gl_sub(a, b) : gl_mul(a, b);
        } else emit(ti + in.dst, a);          // the device copy of the program carries explicit term indices (zk_plonk_setup)
    }
    u64 zi = p.zh_inv[0];                 // select over constant indices: see quotient_block (prover_quotient.hip)
    {
        const u32 zsel = i & (p.qdf - 1);
#pragma unroll
        for (u32 j = 1; j < 8; j++) zi = zsel == j ? p.zh_inv[j] : zi;
    }
#pragma unroll
    for (int j = 0; j < 4; j++) if ((u32)j < p.nch) p.out[(size_t)j * p.L + s] = gl_mul(acc[j], zi);
}

__global__ void plonk_zeta_kernel(u64 *zeta /* in: zeta; out: +2 = g_n * zeta */, u32 log_n, u64 g_n, u64 *err) {
    if (threadIdx.x || blockIdx.x) return;
    gl2 z = gl2_make(zeta[0], zeta[1]), pw = z;
    for (u32 i = 0; i < log_n; i++) pw = gl2_mul(pw, pw);
    if (pw.a == 1 && pw.b == 0) atomicCAS((unsigned long long *)err, 0ULL, (unsigned long long)(uint32_t)ZK_ERR_ZETA_IN_SUBGROUP);
    zeta[2] = gl_mul(z.a, g_n); zeta[3] = gl_mul(z.b, g_n);
}
__global__ void plonk_copy_kernel(const u64 *src, u64 *dst, u32 n) { u32 i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) dst[i] = src[i]; }
__global__ void plonk_canon_kernel(u64 *d, size_t n) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) d[i] = gl_canon(d[i]); }

static inline dim3 g1(size_t n) { return dim3((unsigned)((n + 255) / 256)); }

extern "C" int32_t zk_plonk_setup(zk_ctx *ctx, const zk_stark_config *cfg, const zk_plonk_circuit *c, zk_plonk_state **out) {
    if (!ctx || !out) return ZK_ERR_INVALID_ARG;
    *out = nullptr;
    if (plonk_circuit_check(cfg, c) != ZK_OK) return ctx->fail(ZK_ERR_INVALID_ARG, "plonk setup: malformed circuit or unsupported config");
    hipSetDevice(ctx->device);
    const size_t n = (size_t)1 << c->degree_bits;
    const u32 NC = c->num_constants, R = c->num_routed_wires, capw = 4u << cfg->cap_height;
    zk_plonk_state *st = new zk_plonk_state();
    st->ctx = ctx; st->c = *c; st->cfg = *cfg; st->c.gates = nullptr; st->c.constants = nullptr; st->c.sigmas = nullptr;
    st->n_instrs = c->gates->n_instrs; st->n_regs = c->gates->n_regs; st->n_gconst = c->gates->n_constants;
    st->n_gate_terms = plonk_gate_terms(c->gates);
    ZkOwned own(ctx);
    u64 *d_cols = (u64 *)own.get((size_t)(NC + R) * n * sizeof(u64));
    st->d_instrs = (zk_instr *)ctx->alloc((size_t)(st->n_instrs ? st->n_instrs : 1) * sizeof(zk_instr));
    st->d_gconst = (u64 *)ctx->alloc((size_t)(st->n_gconst ? st->n_gconst : 1) * sizeof(u64));
    st->d_kis = (u64 *)ctx->alloc((size_t)R * sizeof(u64));
    st->d_digest = (u64 *)ctx->alloc(4 * sizeof(u64));
    int32_t rc = (d_cols && st->d_instrs && st->d_gconst && st->d_kis && st->d_digest) ? ZK_OK : ZK_ERR_OOM;
    if (rc == ZK_OK) rc = zk_dev_upload(ctx, d_cols, c->constants, (size_t)NC * n * sizeof(u64));
    if (rc == ZK_OK) rc = zk_dev_upload(ctx, d_cols + (size_t)NC * n, c->sigmas, (size_t)R * n * sizeof(u64));
    if (rc == ZK_OK) ZK_LAUNCH(ctx, "plonk_canon", plonk_canon_kernel, g1((size_t)(NC + R) * n), dim3(256), 0, d_cols, (size_t)(NC + R) * n);
    if (rc == ZK_OK && st->n_instrs) {
        // device copy of the gate program: every emit carries its term index (ZK_OP_EMIT = the running count of EMITs)
        std::vector<zk_instr> prog(c->gates->instrs, c->gates->instrs + st->n_instrs);
        u32 seq = 0;
        for (auto &in : prog) if (in.op == ZK_OP_EMIT) { in.op = ZK_OP_EMIT_SLOT; in.dst = seq++; }
        rc = zk_dev_upload(ctx, st->d_instrs, prog.data(), prog.size() * sizeof(zk_instr));
    }
    if (rc == ZK_OK && st->n_gconst) {
        std::vector<u64> gc(c->gates->constants, c->gates->constants + st->n_gconst);
        for (auto &v : gc) v = gl_canon(v);
        rc = zk_dev_upload(ctx, st->d_gconst, gc.data(), gc.size() * sizeof(u64));
    }
    if (rc == ZK_OK) {
        std::vector<u64> ks(R); u64 x = 1;
        for (u32 j = 0; j < R; j++) { ks[j] = x; x = gl_mul(x, GL_GEN); }
        rc = zk_dev_upload(ctx, st->d_kis, ks.data(), ks.size() * sizeof(u64));
    }
    if (rc == ZK_OK) rc = zk_batch_commit_dev(ctx, d_cols, c->degree_bits, NC + R, cfg->rate_bits, cfg->cap_height, 1, &st->cs);
    if (rc == ZK_OK) {
        // CircuitBuilder::build: circuit digest = hash_no_pad(constants_sigmas_cap | hash_pad(domain_separator) | degree_bits);
        // hash_pad(x) = hash_no_pad(x | 1 | 0.. | 1), padded to a multiple of ZK_HASH_PAD_MULTIPLE (zkgl.h: the rate): [1,0,0,0,0,0,0,1] for the empty separator
        u32 nd = c->n_domain_separator, np = nd + 1;
        while ((np + 1) % ZK_HASH_PAD_MULTIPLE) np++;
        np++;
        u64 *d_in = (u64 *)own.get((size_t)(capw + 5) * sizeof(u64)), *d_pad = (u64 *)own.get((size_t)np * sizeof(u64));
        if (!d_in || !d_pad) rc = ZK_ERR_OOM;
        else {
            std::vector<u64> pad(np, 0);
            for (u32 i = 0; i < nd; i++) pad[i] = gl_canon(c->domain_separator[i]);
            pad[nd] = 1; pad[np - 1] = 1;
            u64 deg = c->degree_bits;
            ZK_LAUNCH(ctx, "plonk_copy", plonk_copy_kernel, g1(capw), dim3(256), 0, st->cs->cap(), d_in, capw);
            rc = zk_dev_upload(ctx, d_pad, pad.data(), pad.size() * sizeof(u64));
            if (rc == ZK_OK) rc = zk_dev_upload(ctx, d_in + capw + 4, &deg, sizeof deg);
            if (rc == ZK_OK) ZK_LAUNCH(ctx, "plonk_hash_no_pad", plonk_hash_no_pad_kernel, dim3(1), dim3(64), 0, (const u64 *)d_pad, np, d_in + capw);
            if (rc == ZK_OK) ZK_LAUNCH(ctx, "plonk_hash_no_pad", plonk_hash_no_pad_kernel, dim3(1), dim3(64), 0, (const u64 *)d_in, capw + 5, st->d_digest);
            if (rc == ZK_OK) rc = zk_dev_download(ctx, st->digest, st->d_digest, 4 * sizeof(u64));
        }
    }
    if (rc == ZK_OK) rc = zk_plonk_gates_kernel(ctx, c->gates, c->num_wires, NC, cfg->num_challenges, cfg->rate_bits, st->gates_fns, &st->gates_fn_is_module, &st->gates_n_state);
    if (rc != ZK_OK) { zk_plonk_state_free(st); return rc; }
    *out = st;
    return ZK_OK;
}
extern "C" int32_t zk_plonk_state_free(zk_plonk_state *st) {
    if (!st) return ZK_ERR_INVALID_ARG;
    if (st->cs) zk_batch_free(st->cs);
    st->ctx->release(st->d_instrs); st->ctx->release(st->d_gconst); st->ctx->release(st->d_kis); st->ctx->release(st->d_digest);
    delete st;
    return ZK_OK;
}
extern "C" int32_t zk_plonk_circuit_digest(zk_plonk_state *st, uint64_t digest_out[4]) {
    if (!st || !digest_out) return ZK_ERR_INVALID_ARG;
    memcpy(digest_out, st->digest, 4 * sizeof(u64));
    return ZK_OK;
}

extern "C" int32_t zk_prove_plonk(zk_ctx *ctx, const zk_stark_config *cfgp, zk_plonk_state *st, const uint64_t *wires, uint32_t wires_on_device,
                                  const uint64_t *public_inputs, uint32_t n_pub, uint64_t *proof_out, size_t cap_words, size_t *len_words) {
    // the prover data of a circuit is read-only after zk_plonk_setup (which ends with a host wait): any context of the same device may prove with it,
    // as every worker of the reference shares one P_STATE (/root/reference/common/src/prover_state/mod.rs:24-31)
    if (!ctx || !cfgp || !st || !wires || !proof_out || (n_pub && !public_inputs) || st->ctx->device != ctx->device) return ZK_ERR_INVALID_ARG;
    if (memcmp(cfgp, &st->cfg, sizeof(zk_stark_config)) != 0) return ctx->fail(ZK_ERR_INVALID_ARG, "prove_plonk: config differs from the one of zk_plonk_setup");
    const zk_stark_config &cfg = st->cfg;
    const zk_plonk_circuit &c = st->c;
    hipSetDevice(ctx->device);
    if (wires_on_device) ctx->wait_upload(wires);
    const u32 nch = cfg.num_challenges, qdf = c.quotient_degree_factor, R = c.num_routed_wires, W = c.num_wires, NC = c.num_constants;
    const u32 k = c.degree_bits, lgL = k + cfg.rate_bits, nchunks = R / qdf, npp = nchunks - 1, capw = 4u << cfg.cap_height, NZ = nch * (1 + npp), Q = nch * qdf;
    const size_t n = (size_t)1 << k, L = (size_t)1 << lgL;
    const PlonkLayout y = plonk_layout(cfg, c);
    if (len_words) *len_words = y.total;
    if (cap_words < y.total) return ctx->fail(ZK_ERR_BUFFER_TOO_SMALL, "prove_plonk: proof buffer too small");

    int32_t rc = ZK_OK;
    struct Guard {
        zk_ctx *ctx; std::vector<void *> owned; zk_batch *wb = nullptr, *zb = nullptr, *qb = nullptr;
        ~Guard() { hipStreamSynchronize(ctx->stream); if (wb) zk_batch_free(wb); if (zb) zk_batch_free(zb); if (qb) zk_batch_free(qb); for (void *q : owned) ctx->release(q); }
    } g{ctx, {}};
    auto A_ = [&](size_t bytes) { void *p = ctx->alloc(bytes); if (p) g.owned.push_back(p); return p; };
    u64 *d_proof = (u64 *)A_(y.total * sizeof(u64));
    DevChallenger *ch = (DevChallenger *)A_(sizeof(DevChallenger));
    u64 *d_bg = (u64 *)A_(8 * sizeof(u64)), *d_alphas = (u64 *)A_(4 * sizeof(u64)), *d_zeta = (u64 *)A_(8 * sizeof(u64)), *d_pih = (u64 *)A_(4 * sizeof(u64));
    if (!d_proof || !ch || !d_bg || !d_alphas || !d_zeta || !d_pih) return ctx->fail(ZK_ERR_OOM, "prove_plonk: out of device memory");
    ctx->h_stage_used = 0;
    zk_host_transcript_begin_proof(ctx);
    u64 *h_out = (u64 *)ctx->pinned_out((y.total + 1) * sizeof(u64));
    if (!h_out) return ctx->fail(ZK_ERR_OOM, "prove_plonk: pinned download buffer");
    {
        u64 h5[5] = {0x314B4C504C474B5AULL, k, W, R, NC};
        if ((rc = ctx->upload_small(d_proof, h5, sizeof h5)) != ZK_OK) return rc;
    }
    // ---- public_inputs_hash = hash_no_pad(public_inputs)
    if (n_pub) {
        u64 *d_pub = (u64 *)A_((size_t)n_pub * sizeof(u64));
        if (!d_pub) return ctx->fail(ZK_ERR_OOM, "prove_plonk: out of device memory");
        std::vector<u64> pc(public_inputs, public_inputs + n_pub);
        for (auto &v : pc) v = gl_canon(v);
        if ((rc = ctx->upload_small(d_pub, pc.data(), pc.size() * sizeof(u64))) != ZK_OK) return rc;
        ZK_LAUNCH(ctx, "plonk_hash_no_pad", plonk_hash_no_pad_kernel, dim3(1), dim3(64), 0, (const u64 *)d_pub, n_pub, d_pih);
    } else hipMemsetAsync(d_pih, 0, 4 * sizeof(u64), ctx->stream);

    // ---- wires commitment
    const u64 *d_wires = nullptr;
    if (wires_on_device) d_wires = (const u64 *)wires;
    else {
        u64 *d = (u64 *)A_((size_t)W * n * sizeof(u64));
        if (!d) return ctx->fail(ZK_ERR_OOM, "prove_plonk: out of device memory");
        if ((rc = ctx->upload_large(d, wires, (size_t)W * n * sizeof(u64), ctx->stream)) != ZK_OK) return rc;
        ZK_LAUNCH(ctx, "plonk_canon", plonk_canon_kernel, g1((size_t)W * n), dim3(256), 0, d, (size_t)W * n);
        d_wires = d;
    }
    if ((rc = zk_batch_commit_dev(ctx, d_wires, k, W, cfg.rate_bits, cfg.cap_height, 1, &g.wb)) != ZK_OK) return rc;
    ZK_LAUNCH(ctx, "plonk_copy", plonk_copy_kernel, g1(capw), dim3(256), 0, g.wb->cap(), d_proof + y.o_wcap, capw);
    {
        TOpList a = {}; a.n = 6;
        a.op[0].kind = TOP_INIT;
        a.op[1].kind = TOP_OBSERVE; a.op[1].src = st->d_digest; a.op[1].n = 4;
        a.op[2].kind = TOP_OBSERVE; a.op[2].src = d_pih; a.op[2].n = 4;
        a.op[3].kind = TOP_OBSERVE; a.op[3].src = d_proof + y.o_wcap; a.op[3].n = capw;
        a.op[4].kind = TOP_GET; a.op[4].dst = d_bg; a.op[4].n = nch;                 // betas
        a.op[5].kind = TOP_GET; a.op[5].dst = d_bg + nch; a.op[5].n = nch;           // gammas
        zk_transcript_run(ctx, ch, a);
    }
    // ---- partial products and Z of the routed-wire permutation
    {
        u64 *d_q = (u64 *)ctx->alloc((size_t)nch * nchunks * n * sizeof(u64));
        u64 *d_zpp = (u64 *)ctx->alloc((size_t)NZ * n * sizeof(u64));
        if (!d_q || !d_zpp) { ctx->release(d_q); ctx->release(d_zpp); return ctx->fail(ZK_ERR_OOM, "prove_plonk: out of device memory"); }
        PermParams pp = {};
        TwoLevel xs = zk_scale_table(ctx, gl_root(k), 1, k);
        pp.wires = d_wires; pp.sigmas = nullptr; pp.n = n; pp.kis = st->d_kis; pp.bg = d_bg; pp.xA = xs.A; pp.xB = xs.B;
        pp.R = R; pp.qdf = qdf; pp.nchunks = nchunks; pp.nch = nch; pp.q = d_q; pp.err = ctx->d_err;
        // the running product needs the sigma VALUES on H; the commitment keeps coefficients and the coset LDE, so they are
        // re-derived exactly with one forward NTT of the sigma coefficient columns (bit-reversed coefficients -> natural values)
        u64 *d_sig = (u64 *)ctx->alloc((size_t)R * n * sizeof(u64));
        if (!d_sig) { ctx->release(d_q); ctx->release(d_zpp); return ctx->fail(ZK_ERR_OOM, "prove_plonk: out of device memory"); }
        rc = zk_ntt_dit(ctx, st->cs->coeffs + (size_t)NC * n, n, d_sig, n, k, R, 0, nullptr, nullptr, 0, nullptr, 0);
        pp.sigmas = d_sig;
        if (rc == ZK_OK) {
            static const int perm_split = getenv("ZK_PLONK_PERM_SPLIT") ? atoi(getenv("ZK_PLONK_PERM_SPLIT")) : 1;
            if (perm_split && nch <= 8 && nch * nchunks <= 64 && n <= ((size_t)1 << 18))
                ZK_LAUNCH(ctx, "plonk_perm_rows", plonk_perm_rows_split_kernel, dim3((unsigned)((n + 31) / 32)), dim3(256), 0, pp);
            else if (nchunks <= PLK_MAX_CHUNKS) ZK_LAUNCH(ctx, "plonk_perm_rows", plonk_perm_rows_kernel, g1(n), dim3(256), 0, pp);
            else {
                u64 *d_den = (u64 *)A_((size_t)nchunks * n * sizeof(u64));
                if (!d_den) rc = ctx->fail(ZK_ERR_OOM, "prove_plonk: out of device memory");
                else ZK_LAUNCH(ctx, "plonk_perm_rows_wide", plonk_perm_rows_wide_kernel, g1(n), dim3(256), 0, pp, d_den);
            }
            if (rc) { ctx->release(d_sig); ctx->release(d_q); ctx->release(d_zpp); return rc; }
            ZK_LAUNCH(ctx, "plonk_prefix_product", plonk_prefix_product_kernel, dim3(nch), dim3(1024), 0, (const u64 *)d_q, n, nchunks, d_zpp);
            if (npp) ZK_LAUNCH(ctx, "plonk_partial_products", plonk_partial_products_kernel, dim3((unsigned)((n + 255) / 256), nch), dim3(256), 0, (const u64 *)d_q, n, nchunks, nch, d_zpp);
            rc = zk_batch_commit_dev(ctx, d_zpp, k, NZ, cfg.rate_bits, cfg.cap_height, 1, &g.zb);
        }
        ctx->release(d_sig); ctx->release(d_q); ctx->release(d_zpp);
        if (rc) return rc;
    }
    ZK_LAUNCH(ctx, "plonk_copy", plonk_copy_kernel, g1(capw), dim3(256), 0, g.zb->cap(), d_proof + y.o_zcap, capw);
    {
        TOpList a = {}; a.n = 2;
        a.op[0].kind = TOP_OBSERVE; a.op[0].src = d_proof + y.o_zcap; a.op[0].n = capw;
        a.op[1].kind = TOP_GET; a.op[1].dst = d_alphas; a.op[1].n = nch;
        zk_transcript_run(ctx, ch, a);
    }
    // ---- quotient polynomials
    {
        u64 *d_qv = (u64 *)ctx->alloc((size_t)nch * L * sizeof(u64));
        if (!d_qv) return ctx->fail(ZK_ERR_OOM, "prove_plonk: out of device memory");
        u64 *lag = nullptr;
        rc = zk_lagrange_table(ctx, k, cfg.rate_bits, &lag);
        if (rc == ZK_OK) {
            PlkQuotParams qp = {};
            qp.w_lde = g.wb->lde; qp.z_lde = g.zb->lde; qp.cs_lde = st->cs->lde; qp.l0 = lag;
            TwoLevel xs = zk_scale_table(ctx, gl_root(lgL), GL_GEN, lgL);
            const u32 NT = nch + nch * nchunks + st->n_gate_terms;
            u64 *d_apow = (u64 *)A_((size_t)nch * NT * sizeof(u64));
            if (!d_apow) { ctx->release(d_qv); return ctx->fail(ZK_ERR_OOM, "prove_plonk: out of device memory"); }
            ZK_LAUNCH(ctx, "plonk_alpha_pows", plonk_alpha_pows_kernel, g1((size_t)nch * NT), dim3(256), 0, (const u64 *)d_alphas, nch, NT, d_apow);
            qp.xA = xs.A; qp.xB = xs.B; qp.kis = st->d_kis; qp.bg = d_bg; qp.apow = d_apow; qp.NT = NT; qp.instrs = st->d_instrs; qp.gconst = st->d_gconst;
            const u64 gn = gl_pow(GL_GEN, n), wr = gl_root(cfg.rate_bits);
            for (u32 i = 0; i < qdf; i++) qp.zh_inv[i] = gl_inv(gl_sub(gl_mul(gn, gl_pow(wr, i)), 1));
            qp.L = L; qp.lgL = lgL; qp.W = W; qp.NC = NC; qp.R = R; qp.NZ = NZ; qp.qdf = qdf; qp.nchunks = nchunks; qp.nch = nch;
            qp.n_instrs = st->n_instrs; qp.n_regs = st->n_regs; qp.out = d_qv; qp.pih = d_pih;
            if (!st->gates_fns.empty()) {     // the circuit's gate program as compiled code (same results as the interpreter kernel below)
                if (st->gates_n_state) {
                    qp.state = (u64 *)A_((size_t)st->gates_n_state * L * sizeof(u64));
                    if (!qp.state) { ctx->release(d_qv); return ctx->fail(ZK_ERR_OOM, "prove_plonk: out of device memory"); }
                }
                void *args[] = {&qp};
                for (size_t kk = 0; kk < st->gates_fns.size() && rc == ZK_OK; kk++) {
                    hipError_t e;
                    ProfScope ps(ctx, "plonk_quotient");
                    if (ctx->sync_launch) fprintf(stderr, "[zk %p] plonk_quotient (compiled gate program, %s, kernel %zu of %zu) grid %u\n", (void *)ctx,
                                                  st->gates_fn_is_module ? "hiprtc" : "build-time", kk + 1, st->gates_fns.size(), (unsigned)((L + 255) / 256));
                    // kernel 0 (the permutation part) spreads a point's (challenge, chunk) pairs over 8 threads: 32 points per workgroup
                    const unsigned grid = kk == 0 ? (unsigned)((L + 31) / 32) : (unsigned)((L + 255) / 256);
                    ctx->note_launch(kk == 0 ? "plonk_quotient(compiled,0)" : "plonk_quotient(compiled,k)", grid, 256);
                    if (st->gates_fn_is_module) e = hipModuleLaunchKernel((hipFunction_t)st->gates_fns[kk], grid, 1, 1, 256, 1, 1, 0, ctx->stream, args, nullptr);
                    else e = hipLaunchKernel(st->gates_fns[kk], dim3(grid), dim3(256), args, 0, ctx->stream);
                    if (e != hipSuccess) rc = ctx->fail(ZK_ERR_HIP, "plonk quotient: launch of a compiled gate kernel failed");
                    if (ctx->sync_launch) hipStreamSynchronize(ctx->stream);
                }
            } else
                ZK_LAUNCH(ctx, "plonk_quotient", plonk_quotient_kernel, g1(L), dim3(256), (size_t)st->n_regs * 256 * sizeof(u64), qp);
            // coset_ifft(g): leaf-order values -> natural-order coefficients, scaled by L^-1 g^-j; in place
            TwoLevel post = zk_scale_table(ctx, gl_inv(GL_GEN), gl_inv((u64)L % GL_P), lgL);
            if (rc == ZK_OK) rc = zk_ntt_dit(ctx, d_qv, L, d_qv, L, lgL, nch, 1, nullptr, &post, 0, nullptr, 0);
        }
        // each challenge's L coefficients are qdf polynomials of n coefficients, back to back
        if (rc == ZK_OK) rc = zk_batch_commit_dev(ctx, d_qv, k, Q, cfg.rate_bits, cfg.cap_height, 0, &g.qb);
        ctx->release(d_qv);
        if (rc) return rc;
    }
    ZK_LAUNCH(ctx, "plonk_copy", plonk_copy_kernel, g1(capw), dim3(256), 0, g.qb->cap(), d_proof + y.o_qcap, capw);
    {
        TOpList a = {}; a.n = 2;
        a.op[0].kind = TOP_OBSERVE; a.op[0].src = d_proof + y.o_qcap; a.op[0].n = capw;
        a.op[1].kind = TOP_GET; a.op[1].dst = d_zeta; a.op[1].n = 2;
        zk_transcript_run(ctx, ch, a);
    }
    ZK_LAUNCH(ctx, "plonk_zeta", plonk_zeta_kernel, dim3(1), dim3(1), 0, d_zeta, k, gl_root(k), ctx->d_err);
    // ---- OpeningSet::new
    {
        zk_eval_job jobs[6]; u32 nj = 0;
        jobs[nj++] = {st->cs->coeffs, n, NC + R, d_zeta, d_proof + y.o_cs};
        jobs[nj++] = {g.wb->coeffs, n, W, d_zeta, d_proof + y.o_w};
        jobs[nj++] = {g.zb->coeffs, n, nch, d_zeta, d_proof + y.o_z};
        jobs[nj++] = {g.zb->coeffs, n, nch, d_zeta + 2, d_proof + y.o_zn};
        if (npp) jobs[nj++] = {g.zb->coeffs + (size_t)nch * n, n, nch * npp, d_zeta, d_proof + y.o_pp};
        jobs[nj++] = {g.qb->coeffs, n, Q, d_zeta, d_proof + y.o_q};
        rc = zk_eval_cols_multi(ctx, k, nj, jobs);
    }
    if (rc) return rc;
    {   // observe_openings(to_fri_openings): zeta batch (constants, sigmas, wires, zs, partial products, quotient), then zs_next
        TOpList a = {}; int q = 0;
        a.op[q].kind = TOP_OBSERVE; a.op[q].src = d_proof + y.o_cs; a.op[q].n = 2 * (NC + R + W + nch); q++;
        a.op[q].kind = TOP_OBSERVE; a.op[q].src = d_proof + y.o_pp; a.op[q].n = 2 * (nch * npp + Q); q++;
        a.op[q].kind = TOP_OBSERVE; a.op[q].src = d_proof + y.o_zn; a.op[q].n = 2 * nch; q++;
        a.n = q;
        zk_transcript_run(ctx, ch, a);
    }
    // ---- prove_openings over [constants_sigmas, wires, zs_partial_products, quotient]; zeta and g*zeta stay in HBM (d_zeta)
    {
        const zk_batch *oracles[4] = {st->cs, g.wb, g.zb, g.qb};
        std::vector<zk_fri_poly> p0, p1;
        for (u32 i = 0; i < NC + R; i++) p0.push_back({0, i});
        for (u32 i = 0; i < W; i++) p0.push_back({1, i});
        for (u32 i = 0; i < NZ; i++) p0.push_back({2, i});
        for (u32 i = 0; i < Q; i++) p0.push_back({3, i});
        for (u32 i = 0; i < nch; i++) p1.push_back({2, i});
        FriBatchDesc bt[2];
        bt[0].point[0] = bt[0].point[1] = 0; bt[0].n_polys = (u32)p0.size(); bt[0].polys = p0.data(); bt[0].d_point = d_zeta;
        bt[1].point[0] = bt[1].point[1] = 0; bt[1].n_polys = (u32)p1.size(); bt[1].polys = p1.data(); bt[1].d_point = d_zeta + 2;
        rc = zk_fri_prove_instance_dev(ctx, cfg, ch, oracles, 4, bt, 2, d_proof + y.o_fri, y.fri);
        if (rc) return rc;
    }
    // ---- fetch the proof and the device error word
    u64 err_word = 0;
    if (hipMemcpyAsync(h_out, d_proof, y.total * sizeof(u64), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) rc = ZK_ERR_HIP;
    if (hipMemcpyAsync(h_out + y.total, ctx->d_err, sizeof(u64), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) rc = ZK_ERR_HIP;
    hipError_t se = ctx->wait_stream();
    if (se != hipSuccess && rc == ZK_OK) rc = ctx->fail(ZK_ERR_HIP, std::string("stream sync: ") + hipGetErrorString(se));
    if (rc == ZK_OK) { memcpy(proof_out, h_out, y.total * sizeof(u64)); err_word = h_out[y.total]; }
    if (rc == ZK_OK && err_word) {
        rc = (int32_t)(uint32_t)err_word;
        ctx->fail(rc, "prove_plonk: device-side check failed");
        hipMemsetAsync(ctx->d_err, 0, sizeof(u64), ctx->stream);
    }
    if (const int32_t he = zk_host_transcript_take_error(ctx)) { if (rc == ZK_OK) { rc = he; ctx->fail(rc, "prove_plonk: host-side transcript check failed"); } }
    return rc;
}
