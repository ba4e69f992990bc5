// Device-resident PolynomialBatch: iNTT -> coset LDE -> Poseidon leaf hashing -> Merkle levels -> cap.
// Mirrors plonky2 0.2.2 fri::oracle::PolynomialBatch::{from_values, from_coeffs, get_lde_values} and
// hash::merkle_tree::MerkleTree::{new, prove} (crate pinned at /root/reference/Cargo.lock:3441-3443; SURVEY.md A5,
// 8a rows a5/a7/a11).  The reference reaches it from /root/reference/ops/src/lib.rs:52.
//
// HBM layout of one batch (n = 2^log_n rows, L = n << rate_bits, C columns):
//   coeffs  C x n   column-major, coefficient j at position bitrev(j)        (what the DIF inverse pass produces)
//   lde     C x L   column-major, row r = values at g*w_L^bitrev(r)          (= upstream's leaf order)
//   levels  (2L - 2^cap) x 4 words, level-order digests; the last 2^cap are the cap
// Upstream keeps leaves row-major and digests in a recursive layout; zk_batch_get_leaf / zk_batch_get_digests
// convert on demand, the prover itself never needs either.
#include "zk_internal.h"

__global__ void gather_rows_kernel(const u64 *lde, size_t L, u32 n_cols, const u32 *leaf_idx, u32 n_idx, u64 *out) {
    size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (size_t)n_idx * n_cols) return;
    u32 i = (u32)(g / n_cols), c = (u32)(g % n_cols);
    out[g] = lde[(size_t)c * L + leaf_idx[i]];
}
__global__ void gather_path_kernel(const u64 *levels, size_t L, u32 n_sib, u32 leaf, u64 *out) {
    u32 t = threadIdx.x;
    if (t >= n_sib * 4) return;
    u32 k = t >> 2, w = t & 3;
    size_t off = 0;
    for (u32 j = 0; j < k; j++) off += L >> j;
    out[t] = levels[4 * (off + ((leaf >> k) ^ 1)) + w];
}

// ---- polynomial evaluation at an extension point, coefficients in bit-reversed order.
// pos = B*2^(LT+LI) + i*2^LT + t  ->  j = bitrev(pos) = bitrev(t)*2^(k-LT) + bitrev(i)*2^LB + bitrev(B)
// so z^j = Zt[t] * Zi[i] * ZB[B]: three small power tables, 2 modmuls per coefficient, fully coalesced reads.
struct EvalGeom { int k, LT, LI, LB; };
static EvalGeom eval_geom(int k) {
    EvalGeom g; g.k = k;
    g.LT = k < 8 ? k : 8;
    g.LI = (k - g.LT) < 6 ? (k - g.LT) : 6;
    g.LB = k - g.LT - g.LI;
    return g;
}
__device__ inline gl2 ext_pow_bits(gl2 base, u32 e) {   // base^e
    gl2 r = gl2_make(1, 0);
    while (e) { if (e & 1) r = gl2_mul(r, base); base = gl2_mul(base, base); e >>= 1; }
    return r;
}
// tables: Zt[2^LT] | Zi[2^LI] | ZB[2^LB]  (ext elements)
__global__ void eval_tables_kernel(const u64 *point, EvalGeom g, u64 *tab) {
    u32 id = blockIdx.x * blockDim.x + threadIdx.x;
    u32 nT = 1u << g.LT, nI = 1u << g.LI, nB = 1u << g.LB;
    if (id >= nT + nI + nB) return;
    gl2 z = gl2_make(gl_canon(point[0]), gl_canon(point[1]));
    gl2 base; u32 e;
    if (id < nT) { base = z; for (int s = 0; s < g.k - g.LT; s++) base = gl2_mul(base, base); e = bitrev_u32(id, g.LT); }
    else if (id < nT + nI) { base = z; for (int s = 0; s < g.LB; s++) base = gl2_mul(base, base); e = bitrev_u32(id - nT, g.LI); }
    else { base = z; e = bitrev_u32(id - nT - nI, g.LB); }
    gl2 v = ext_pow_bits(base, e);
    tab[2 * id] = v.a; tab[2 * id + 1] = v.b;
}
__global__ __launch_bounds__(256) void eval_cols_kernel(const u64 *__restrict__ coeffs, size_t stride, EvalGeom g, const u64 *__restrict__ tab,
                                                        u64 *__restrict__ partial) {
    __shared__ u64 red[2 * 256];
    const u32 t = threadIdx.x, B = blockIdx.x, col = blockIdx.y;
    const u32 nT = 1u << g.LT, nI = 1u << g.LI;
    const u64 *zi = tab + 2 * nT, *zb = tab + 2 * (nT + nI);
    gl2 acc = gl2_make(0, 0);
    if (t < nT) {
        const u64 *p = coeffs + (size_t)col * stride + ((size_t)B << (g.LT + g.LI)) + t;
        for (u32 i = 0; i < nI; i++) {
            u64 c = p[(size_t)i << g.LT];
            acc.a = gl_add(acc.a, gl_mul(c, zi[2 * i]));
            acc.b = gl_add(acc.b, gl_mul(c, zi[2 * i + 1]));
        }
        acc = gl2_mul(acc, gl2_make(tab[2 * t], tab[2 * t + 1]));
    }
    red[2 * t] = acc.a; red[2 * t + 1] = acc.b;
    __syncthreads();
    for (u32 s = 128; s > 0; s >>= 1) {
        if (t < s) { red[2 * t] = gl_add(red[2 * t], red[2 * (t + s)]); red[2 * t + 1] = gl_add(red[2 * t + 1], red[2 * (t + s) + 1]); }
        __syncthreads();
    }
    if (t == 0) {
        gl2 r = gl2_mul(gl2_make(red[0], red[1]), gl2_make(zb[2 * B], zb[2 * B + 1]));
        size_t o = ((size_t)col << g.LB) + B;
        partial[2 * o] = r.a; partial[2 * o + 1] = r.b;
    }
}
__global__ void eval_reduce_kernel(const u64 *partial, int LB, int n_cols, u64 *out) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_cols) return;
    u64 a = 0, b = 0;
    for (u32 B = 0; B < (1u << LB); B++) { a = gl_add(a, partial[2 * (((size_t)c << LB) + B)]); b = gl_add(b, partial[2 * (((size_t)c << LB) + B) + 1]); }
    out[2 * c] = a; out[2 * c + 1] = b;
}

// Several evaluations of one proof in three launches: every column block of every job in ONE eval_cols launch, the power tables built once
// per distinct point (OpeningSet::new / StarkOpeningSet::new are six column blocks at two points: eighteen launches one by one).
struct EvalMulti {
    u32 n_jobs, n_points;
    u32 col0[ZK_EVAL_MAX_JOBS + 1];                      // first column (of the concatenation) of every job
    u32 point_of[ZK_EVAL_MAX_JOBS];
    const u64 *coeffs[ZK_EVAL_MAX_JOBS]; size_t stride[ZK_EVAL_MAX_JOBS];
    u64 *out[ZK_EVAL_MAX_JOBS];
    const u64 *points[ZK_EVAL_MAX_JOBS];
    // jobs that read the SAME coefficient columns at two points (the openings at zeta and at g * zeta) are evaluated in one pass over the
    // columns: the launch covers the columns of the primary jobs only, pair[j] = the job whose sums the same blocks produce as well
    u32 n_launch_jobs, launch_job[ZK_EVAL_MAX_JOBS], lcol0[ZK_EVAL_MAX_JOBS + 1], pair[ZK_EVAL_MAX_JOBS];
};
__global__ void eval_tables_multi_kernel(EvalMulti m, EvalGeom g, u32 ntab, u64 *tab) {
    const u32 pt = blockIdx.y, id = blockIdx.x * blockDim.x + threadIdx.x;
    const u32 nT = 1u << g.LT, nI = 1u << g.LI, nB = 1u << g.LB;
    if (id >= nT + nI + nB) return;
    const u64 *point = m.points[pt];
    gl2 z = gl2_make(gl_canon(point[0]), gl_canon(point[1]));
    gl2 base; u32 e;
    if (id < nT) { base = z; for (int s = 0; s < g.k - g.LT; s++) base = gl2_mul(base, base); e = bitrev_u32(id, g.LT); }
    else if (id < nT + nI) { base = z; for (int s = 0; s < g.LB; s++) base = gl2_mul(base, base); e = bitrev_u32(id - nT, g.LI); }
    else { base = z; e = bitrev_u32(id - nT - nI, g.LB); }
    gl2 v = ext_pow_bits(base, e);
    tab[2 * ((size_t)pt * ntab + id)] = v.a; tab[2 * ((size_t)pt * ntab + id) + 1] = v.b;
}
__global__ __launch_bounds__(256) void eval_cols_multi_kernel(EvalMulti m, EvalGeom g, u32 ntab, const u64 *__restrict__ tabs, u64 *__restrict__ partial) {
    __shared__ u64 red[2 * 256];
    const u32 t = threadIdx.x, B = blockIdx.x, lcol = blockIdx.y;
    u32 lj = 0;
    while (lj + 1 < m.n_launch_jobs && lcol >= m.lcol0[lj + 1]) lj++;
    const u32 job = m.launch_job[lj], col = lcol - m.lcol0[lj], job2 = m.pair[job];
    const bool two = job2 != 0xFFFFFFFFu;                                  // block-uniform
    const u64 *tab = tabs + 2 * (size_t)m.point_of[job] * ntab, *tab2 = tabs + 2 * (size_t)m.point_of[two ? job2 : job] * ntab;
    const u32 nT = 1u << g.LT, nI = 1u << g.LI;
    const u64 *zi = tab + 2 * nT, *zb = tab + 2 * (nT + nI), *zi2 = tab2 + 2 * nT, *zb2 = tab2 + 2 * (nT + nI);
    gl2 acc = gl2_make(0, 0), acc2 = gl2_make(0, 0);
    if (t < nT) {
        const u64 *p = m.coeffs[job] + (size_t)col * m.stride[job] + ((size_t)B << (g.LT + g.LI)) + t;
        if (two) {
            for (u32 i = 0; i < nI; i++) {
                const u64 c = p[(size_t)i << g.LT];
                acc.a = gl_add(acc.a, gl_mul(c, zi[2 * i]));
                acc.b = gl_add(acc.b, gl_mul(c, zi[2 * i + 1]));
                acc2.a = gl_add(acc2.a, gl_mul(c, zi2[2 * i]));
                acc2.b = gl_add(acc2.b, gl_mul(c, zi2[2 * i + 1]));
            }
            acc2 = gl2_mul(acc2, gl2_make(tab2[2 * t], tab2[2 * t + 1]));
        } else {
            for (u32 i = 0; i < nI; i++) {
                const u64 c = p[(size_t)i << g.LT];
                acc.a = gl_add(acc.a, gl_mul(c, zi[2 * i]));
                acc.b = gl_add(acc.b, gl_mul(c, zi[2 * i + 1]));
            }
        }
        acc = gl2_mul(acc, gl2_make(tab[2 * t], tab[2 * t + 1]));
    }
    for (int pass = 0; pass < (two ? 2 : 1); pass++) {
        const gl2 v = pass ? acc2 : acc;
        __syncthreads();
        red[2 * t] = v.a; red[2 * t + 1] = v.b;
        __syncthreads();
        for (u32 s = 128; s > 0; s >>= 1) {
            if (t < s) { red[2 * t] = gl_add(red[2 * t], red[2 * (t + s)]); red[2 * t + 1] = gl_add(red[2 * t + 1], red[2 * (t + s) + 1]); }
            __syncthreads();
        }
        if (t == 0) {
            const u64 *zz = pass ? zb2 : zb;
            gl2 r = gl2_mul(gl2_make(red[0], red[1]), gl2_make(zz[2 * B], zz[2 * B + 1]));
            const u32 vcol = m.col0[pass ? job2 : job] + col;
            size_t o = ((size_t)vcol << g.LB) + B;
            partial[2 * o] = r.a; partial[2 * o + 1] = r.b;
        }
    }
}
__global__ void eval_reduce_multi_kernel(EvalMulti m, const u64 *partial, int LB, u64 *unused) {
    const u32 vcol = blockIdx.x * blockDim.x + threadIdx.x;
    if (vcol >= m.col0[m.n_jobs]) return;
    u32 job = 0;
    while (job + 1 < m.n_jobs && vcol >= m.col0[job + 1]) job++;
    u64 a = 0, b = 0;
    for (u32 B = 0; B < (1u << LB); B++) { a = gl_add(a, partial[2 * (((size_t)vcol << LB) + B)]); b = gl_add(b, partial[2 * (((size_t)vcol << LB) + B) + 1]); }
    u64 *out = m.out[job] + 2 * (size_t)(vcol - m.col0[job]);
    out[0] = a; out[1] = b;
}
int32_t zk_eval_cols_multi(zk_ctx *ctx, int k, u32 n_jobs, const zk_eval_job *jobs) {
    if (n_jobs == 0) return ZK_OK;
    if (n_jobs > ZK_EVAL_MAX_JOBS) return ctx->fail(ZK_ERR_INVALID_ARG, "eval_cols_multi: too many jobs");
    EvalMulti m = {};
    m.n_jobs = n_jobs;
    u32 total = 0;
    for (u32 j = 0; j < n_jobs; j++) {
        m.col0[j] = total; total += jobs[j].n_cols;
        m.coeffs[j] = jobs[j].coeffs; m.stride[j] = jobs[j].stride; m.out[j] = jobs[j].d_out;
        u32 pt = 0;
        while (pt < m.n_points && m.points[pt] != jobs[j].d_point) pt++;
        if (pt == m.n_points) m.points[m.n_points++] = jobs[j].d_point;
        m.point_of[j] = pt;
    }
    m.col0[n_jobs] = total;
    if (total == 0) return ZK_OK;
    // pairs: job j and a later job over the same columns at another point (ZK_EVAL_PAIR=0: every job by itself, the A/B switch)
    static const int pairing = getenv("ZK_EVAL_PAIR") ? atoi(getenv("ZK_EVAL_PAIR")) : 1;
    bool secondary[ZK_EVAL_MAX_JOBS] = {};
    u32 ltotal = 0;
    for (u32 j = 0; j < n_jobs; j++) {
        m.pair[j] = 0xFFFFFFFFu;
        if (secondary[j]) continue;
        if (pairing)
            for (u32 q = j + 1; q < n_jobs; q++)
                if (!secondary[q] && jobs[q].coeffs == jobs[j].coeffs && jobs[q].stride == jobs[j].stride && jobs[q].n_cols == jobs[j].n_cols &&
                    m.point_of[q] != m.point_of[j]) { m.pair[j] = q; secondary[q] = true; break; }
        m.launch_job[m.n_launch_jobs] = j; m.lcol0[m.n_launch_jobs] = ltotal; m.n_launch_jobs++;
        ltotal += jobs[j].n_cols;
    }
    m.lcol0[m.n_launch_jobs] = ltotal;
    EvalGeom g = eval_geom(k);
    const u32 ntab = (1u << g.LT) + (1u << g.LI) + (1u << g.LB);
    u64 *tab = (u64 *)ctx->alloc((size_t)ntab * m.n_points * 2 * sizeof(u64));
    u64 *partial = (u64 *)ctx->alloc(((size_t)total << g.LB) * 2 * sizeof(u64));
    if (!tab || !partial) { ctx->release(tab); ctx->release(partial); return ctx->fail(ZK_ERR_OOM, "eval_cols_multi: device memory"); }
    ZK_LAUNCH(ctx, "eval_tables", eval_tables_multi_kernel, dim3((ntab + 255) / 256, m.n_points), dim3(256), 0, m, g, ntab, tab);
    ZK_LAUNCH(ctx, "eval_cols", eval_cols_multi_kernel, dim3(1u << g.LB, ltotal), dim3(256), 0, m, g, ntab, (const u64 *)tab, partial);
    ZK_LAUNCH(ctx, "eval_reduce", eval_reduce_multi_kernel, dim3((total + 255) / 256), dim3(256), 0, m, (const u64 *)partial, g.LB, (u64 *)nullptr);
    ctx->release(tab); ctx->release(partial);
    return ZK_OK;
}

int32_t zk_eval_cols_at(zk_ctx *ctx, const u64 *coeffs, size_t stride, int k, int n_cols, const u64 *d_point, u64 *d_out) {
    EvalGeom g = eval_geom(k);
    u32 ntab = (1u << g.LT) + (1u << g.LI) + (1u << g.LB);
    u64 *tab = (u64 *)ctx->alloc((size_t)ntab * 2 * sizeof(u64));
    u64 *partial = (u64 *)ctx->alloc(((size_t)n_cols << g.LB) * 2 * sizeof(u64));
    if (!tab || !partial) { ctx->release(tab); ctx->release(partial); return ctx->fail(ZK_ERR_OOM, "eval_cols_at: device memory"); }
    ZK_LAUNCH(ctx, "eval_tables", eval_tables_kernel, dim3((ntab + 255) / 256), dim3(256), 0, d_point, g, tab);
    ZK_LAUNCH(ctx, "eval_cols", eval_cols_kernel, dim3(1u << g.LB, n_cols), dim3(256), 0, coeffs, stride, g, (const u64 *)tab, partial);
    ZK_LAUNCH(ctx, "eval_reduce", eval_reduce_kernel, dim3((n_cols + 255) / 256), dim3(256), 0, (const u64 *)partial, g.LB, n_cols, d_out);
    ctx->release(tab); ctx->release(partial);   // stream-ordered: later users of these blocks run after the kernels above
    return ZK_OK;
}

// ------------------------------------------------------------------ commit
int32_t zk_batch_commit_dev(zk_ctx *ctx, const u64 *d_cols, uint32_t log_n, uint32_t n_cols, uint32_t rate_bits,
                            uint32_t cap_height, int values_natural, zk_batch **out) {
    if (!ctx || !d_cols || !out || n_cols == 0 || log_n > 28 || rate_bits > 4 || cap_height > log_n + rate_bits)
        return ctx ? ctx->fail(ZK_ERR_INVALID_ARG, "batch commit: bad arguments") : ZK_ERR_INVALID_ARG;
    hipSetDevice(ctx->device);
    zk_batch *b = new zk_batch();
    b->ctx = ctx; b->log_n = log_n; b->n_cols = n_cols; b->rate_bits = rate_bits; b->cap_height = cap_height;
    size_t n = b->n(), L = b->L();
    size_t n_dig = zk_level_offset(L, b->n_levels());
    b->coeffs = (u64 *)ctx->alloc(n * n_cols * sizeof(u64));
    b->lde = (u64 *)ctx->alloc(L * n_cols * sizeof(u64));
    b->levels = (u64 *)ctx->alloc(n_dig * 4 * sizeof(u64));
    u64 *tmp = (u64 *)ctx->alloc(n * n_cols * sizeof(u64));
    int32_t rc = (b->coeffs && b->lde && b->levels && tmp) ? ZK_OK : ZK_ERR_OOM;
    if (rc == ZK_OK) {
        if (values_natural)   // ifft: natural-order values -> bit-reversed-order coefficients, scaled by n^-1
            rc = zk_ntt_dif(ctx, d_cols, n, b->coeffs, n, log_n, n_cols, 1, nullptr, gl_inv((u64)n % GL_P), 1);
        else
            rc = zk_bitrev_copy(ctx, d_cols, n, b->coeffs, n, log_n, n_cols);
    }
    // lde(rate_bits).coset_fft(g): the size-L transform of zero-padded coefficients splits into 2^rate_bits size-n
    // transforms on the cosets g*w_L^par; output half bitrev(par) of the bit-reversed LDE.
    if (rc == ZK_OK) rc = zk_ntt_lde(ctx, b->coeffs, n, b->lde, L, log_n, n_cols, rate_bits, tmp, n);
    if (rc == ZK_OK) rc = zk_hash_leaves_colmajor(ctx, b->lde, L, n_cols, L, b->levels);
    if (rc == ZK_OK) rc = zk_merkle_levels(ctx, b->levels, L, cap_height);
    ctx->release(tmp);
    if (rc != ZK_OK) { ctx->release(b->coeffs); ctx->release(b->lde); ctx->release(b->levels); delete b; return rc; }
    *out = b;
    return ZK_OK;
}

static int32_t commit_host(zk_ctx *ctx, const uint64_t *cols, uint32_t log_n, uint32_t n_cols, uint32_t rate_bits,
                           uint32_t cap_height, int values_natural, zk_batch **out, uint64_t *cap_out) {
    if (!ctx || !cols || !out || n_cols == 0 || log_n > 28) return ZK_ERR_INVALID_ARG;
    hipSetDevice(ctx->device);
    size_t bytes = ((size_t)n_cols << log_n) * sizeof(u64);
    u64 *d = (u64 *)ctx->alloc(bytes);
    if (!d) return ZK_ERR_OOM;
    int32_t rc = zk_dev_upload(ctx, d, cols, bytes);
    if (rc == ZK_OK) rc = zk_batch_commit_dev(ctx, d, log_n, n_cols, rate_bits, cap_height, values_natural, out);
    ctx->release(d);
    if (rc == ZK_OK && cap_out) rc = zk_batch_get_cap(*out, cap_out);
    return rc;
}

extern "C" {

int32_t zk_batch_from_values(zk_ctx *ctx, const uint64_t *cols, uint32_t log_n, uint32_t n_cols, uint32_t rate_bits,
                             uint32_t cap_height, zk_batch **out, uint64_t *cap_out) {
    return commit_host(ctx, cols, log_n, n_cols, rate_bits, cap_height, 1, out, cap_out);
}
int32_t zk_batch_from_coeffs(zk_ctx *ctx, const uint64_t *cols, uint32_t log_n, uint32_t n_cols, uint32_t rate_bits,
                             uint32_t cap_height, zk_batch **out, uint64_t *cap_out) {
    return commit_host(ctx, cols, log_n, n_cols, rate_bits, cap_height, 0, out, cap_out);
}
int32_t zk_batch_from_values_dev(zk_ctx *ctx, const uint64_t *d_cols, uint32_t log_n, uint32_t n_cols, uint32_t rate_bits,
                                 uint32_t cap_height, zk_batch **out, uint64_t *cap_out) {
    if (ctx && d_cols) ctx->wait_upload(d_cols);      // a zk_trace_upload_async prefetch into this buffer lands first (device-side wait)
    int32_t rc = zk_batch_commit_dev(ctx, (const u64 *)d_cols, log_n, n_cols, rate_bits, cap_height, 1, out);
    if (rc == ZK_OK && cap_out) rc = zk_batch_get_cap(*out, cap_out);
    return rc;
}
int32_t zk_batch_from_coeffs_dev(zk_ctx *ctx, const uint64_t *d_cols, uint32_t log_n, uint32_t n_cols, uint32_t rate_bits,
                                 uint32_t cap_height, zk_batch **out, uint64_t *cap_out) {
    if (ctx && d_cols) ctx->wait_upload(d_cols);
    int32_t rc = zk_batch_commit_dev(ctx, (const u64 *)d_cols, log_n, n_cols, rate_bits, cap_height, 0, out);
    if (rc == ZK_OK && cap_out) rc = zk_batch_get_cap(*out, cap_out);
    return rc;
}

int32_t zk_batch_free(zk_batch *b) {
    if (!b) return ZK_ERR_INVALID_ARG;
    b->ctx->release(b->coeffs); b->ctx->release(b->lde); b->ctx->release(b->levels);
    delete b;
    return ZK_OK;
}

int32_t zk_batch_info(const zk_batch *b, uint32_t *log_n, uint32_t *n_cols, uint32_t *rate_bits, uint32_t *cap_height) {
    if (!b) return ZK_ERR_INVALID_ARG;
    if (log_n) *log_n = b->log_n;
    if (n_cols) *n_cols = b->n_cols;
    if (rate_bits) *rate_bits = b->rate_bits;
    if (cap_height) *cap_height = b->cap_height;
    return ZK_OK;
}

int32_t zk_batch_get_cap(zk_batch *b, uint64_t *cap_out) {
    if (!b || !cap_out) return ZK_ERR_INVALID_ARG;
    return zk_dev_download(b->ctx, cap_out, b->cap(), ((size_t)4 << b->cap_height) * sizeof(u64));
}

static int32_t gather_leaves(zk_batch *b, const std::vector<u32> &leaf, uint64_t *out) {
    zk_ctx *ctx = b->ctx;
    hipSetDevice(ctx->device);
    size_t n_idx = leaf.size();
    u32 *d_idx = (u32 *)ctx->alloc(n_idx * sizeof(u32));
    u64 *d_out = (u64 *)ctx->alloc(n_idx * b->n_cols * sizeof(u64));
    if (!d_idx || !d_out) return ZK_ERR_OOM;
    int32_t rc = zk_dev_upload(ctx, d_idx, leaf.data(), n_idx * sizeof(u32));
    if (rc == ZK_OK) {
        size_t tot = n_idx * b->n_cols;
        ZK_LAUNCH(ctx, "gather_rows", gather_rows_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0,
                  (const u64 *)b->lde, b->L(), b->n_cols, (const u32 *)d_idx, (u32)n_idx, d_out);
        rc = zk_dev_download(ctx, out, d_out, tot * sizeof(u64));
    }
    ctx->release(d_idx); ctx->release(d_out);
    return rc;
}

int32_t zk_batch_get_lde_rows(zk_batch *b, const uint32_t *idx, uint32_t n_idx, uint32_t step, uint64_t *out) {
    if (!b || !idx || !out || n_idx == 0) return ZK_ERR_INVALID_ARG;
    u32 lgL = b->log_n + b->rate_bits;
    std::vector<u32> leaf(n_idx);
    for (u32 i = 0; i < n_idx; i++) {
        uint64_t v = (uint64_t)idx[i] * step;
        if (v >= b->L()) return ZK_ERR_INVALID_ARG;
        leaf[i] = bitrev_u32((u32)v, lgL);
    }
    return gather_leaves(b, leaf, out);
}

int32_t zk_batch_get_leaf(zk_batch *b, uint32_t leaf_idx, uint64_t *out) {
    if (!b || !out || leaf_idx >= b->L()) return ZK_ERR_INVALID_ARG;
    std::vector<u32> leaf(1, leaf_idx);
    return gather_leaves(b, leaf, out);
}

int32_t zk_batch_merkle_prove(zk_batch *b, uint32_t leaf_idx, uint64_t *siblings_out) {
    if (!b || !siblings_out || leaf_idx >= b->L()) return ZK_ERR_INVALID_ARG;
    zk_ctx *ctx = b->ctx;
    hipSetDevice(ctx->device);
    u32 n_sib = b->n_levels() - 1;
    if (n_sib == 0) return ZK_OK;
    u64 *d = (u64 *)ctx->alloc((size_t)n_sib * 4 * sizeof(u64));
    if (!d) return ZK_ERR_OOM;
    ZK_LAUNCH(ctx, "gather_path", gather_path_kernel, dim3(1), dim3(256), 0, (const u64 *)b->levels, b->L(), n_sib, leaf_idx, d);
    int32_t rc = zk_dev_download(ctx, siblings_out, d, (size_t)n_sib * 4 * sizeof(u64));
    ctx->release(d);
    return rc;
}

int32_t zk_batch_get_digests(zk_batch *b, uint64_t *digests_out) {
    if (!b || !digests_out) return ZK_ERR_INVALID_ARG;
    zk_ctx *ctx = b->ctx;
    hipSetDevice(ctx->device);
    size_t total = 2 * (b->L() - ((size_t)1 << b->cap_height));
    if (!total) return ZK_OK;
    u64 *d = (u64 *)ctx->alloc(total * 4 * sizeof(u64));
    if (!d) return ZK_ERR_OOM;
    int32_t rc = zk_digests_upstream_layout(ctx, b->levels, b->L(), b->n_levels(), d);
    if (rc == ZK_OK) rc = zk_dev_download(ctx, digests_out, d, total * 4 * sizeof(u64));
    ctx->release(d);
    return rc;
}

int32_t zk_batch_get_coeffs(zk_batch *b, uint32_t col, uint64_t *out) {
    if (!b || !out || col >= b->n_cols) return ZK_ERR_INVALID_ARG;
    zk_ctx *ctx = b->ctx;
    hipSetDevice(ctx->device);
    size_t n = b->n();
    u64 *d = (u64 *)ctx->alloc(n * sizeof(u64));
    if (!d) return ZK_ERR_OOM;
    int32_t rc = zk_bitrev_copy(ctx, b->coeffs + (size_t)col * n, n, d, n, b->log_n, 1);
    if (rc == ZK_OK) rc = zk_dev_download(ctx, out, d, n * sizeof(u64));
    ctx->release(d);
    return rc;
}

int32_t zk_batch_eval_at(zk_batch *b, const uint64_t point[2], uint64_t *out) {
    if (!b || !point || !out) return ZK_ERR_INVALID_ARG;
    zk_ctx *ctx = b->ctx;
    hipSetDevice(ctx->device);
    u64 *d_pt = (u64 *)ctx->alloc(2 * sizeof(u64)), *d_out = (u64 *)ctx->alloc((size_t)b->n_cols * 2 * sizeof(u64));
    if (!d_pt || !d_out) return ZK_ERR_OOM;
    int32_t rc = zk_dev_upload(ctx, d_pt, point, 2 * sizeof(u64));
    if (rc == ZK_OK) rc = zk_eval_cols_at(ctx, b->coeffs, b->n(), b->log_n, b->n_cols, d_pt, d_out);
    if (rc == ZK_OK) rc = zk_dev_download(ctx, out, d_out, (size_t)b->n_cols * 2 * sizeof(u64));
    ctx->release(d_pt); ctx->release(d_out);
    return rc;
}

}  // extern "C"
