// Auxiliary (logUp / cross-table-lookup shaped) columns and the synthetic witness generator.
// Mirrors starky 0.4.0 lookup::lookup_helper_columns and cross_table_lookup::partial_sums (crate pinned at
// /root/reference/Cargo.lock:4529-4531): helper columns are sums of reciprocals
// of challenge-combined trace columns (batch inversion), Z columns are running sums.  Here
//   h_m[row] = 1/(t[a0] + beta t[b0] + gamma) + 1/(t[a1] + beta t[b1] + gamma),   Z[row] = sum_{s >= row} sum_m h_m[s].
// One thread = one row; reciprocals are batched 8 helper columns at a time (Montgomery's trick, one Fermat inversion).
#include <stdlib.h>
#include "zk_internal.h"
#include "prover.h"

// a^(p-2) with p-2 = 2^64 - 2^32 - 1 = 0xFFFFFFFEFFFFFFFF
__device__ __forceinline__ u64 gl_inv_dev(u64 a) {
    // exponent bits (msb first): 31 ones, a zero, 32 ones
    u64 t = a;                       // a^(2^1 - 1)
    u64 x2 = gl_mul(gl_sqr(t), a);   // 2 ones
    u64 x4 = gl_mul(gl_sqr(gl_sqr(x2)), x2);
    u64 x8 = x4; for (int i = 0; i < 4; i++) x8 = gl_sqr(x8); x8 = gl_mul(x8, x4);
    u64 x16 = x8; for (int i = 0; i < 8; i++) x16 = gl_sqr(x16); x16 = gl_mul(x16, x8);
    u64 x24 = x16; for (int i = 0; i < 8; i++) x24 = gl_sqr(x24); x24 = gl_mul(x24, x8);
    u64 x28 = x24; for (int i = 0; i < 4; i++) x28 = gl_sqr(x28); x28 = gl_mul(x28, x4);
    u64 x30 = gl_mul(gl_sqr(gl_sqr(x28)), x2);
    u64 x31 = gl_mul(gl_sqr(x30), a);
    u64 x32 = gl_mul(gl_sqr(x31), a);
    u64 r = gl_sqr(x31);             // 31 ones followed by a zero
    for (int i = 0; i < 32; i++) r = gl_sqr(r);
    return gl_mul(r, x32);
}

#define AUX_CHUNK 8
// combine(cols) + gamma of one pair: t[a] + beta t[b] (+ beta^2 t[e0] + ..) + gamma, Horner from the last column
__device__ __forceinline__ u64 ctl_combine(const u64 *__restrict__ trace, size_t n, size_t r, u32 a, u32 b, const zk_helper_ext *x, int k, u64 beta, u64 gamma) {
    if (!x || x->n_extra[k] == 0) return gl_add(gl_mul_add(beta, trace[(size_t)b * n + r], trace[(size_t)a * n + r]), gamma);
    u64 acc = trace[(size_t)x->extra[k][x->n_extra[k] - 1] * n + r];
    for (u32 i = x->n_extra[k] - 1; i-- > 0;) acc = gl_mul_add(acc, beta, trace[(size_t)x->extra[k][i] * n + r]);
    acc = gl_mul_add(acc, beta, trace[(size_t)b * n + r]);
    acc = gl_mul_add(acc, beta, trace[(size_t)a * n + r]);
    return gl_add(acc, gamma);
}
__global__ __launch_bounds__(256) void aux_helpers_kernel(const u64 *__restrict__ trace, size_t n, const zk_helper_desc *__restrict__ helpers,
                                                          const zk_helper_ext *__restrict__ ext, u32 H, u32 nch, const u64 *__restrict__ chal,
                                                          u64 *__restrict__ aux, u64 *__restrict__ rowsum, u64 *err) {
    size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    u32 j = blockIdx.y;
    if (r >= n) return;
    u64 beta = chal[2 * j], gamma = chal[2 * j + 1];
    u64 total = 0;
    for (u32 m0 = 0; m0 < H; m0 += AUX_CHUNK) {
        u64 num[AUX_CHUNK], den[AUX_CHUNK], pre[AUX_CHUNK];
        u64 acc = 1;
#pragma unroll
        for (int k = 0; k < AUX_CHUNK; k++) {
            if (m0 + k < H) {
                zk_helper_desc d = helpers[m0 + k];
                const zk_helper_ext *x = ext ? ext + (m0 + k) : nullptr;
                u64 x1 = ctl_combine(trace, n, r, d.a0, d.b0, x, 0, beta, gamma);
                u64 x2 = ctl_combine(trace, n, r, d.a1, d.b1, x, 1, beta, gamma);
                // f_0 / x1 + f_1 / x2 = (f_0 x2 + f_1 x1) / (x1 x2); a filter is a 0/1 trace column, or absent (constant 1)
                const u64 f0 = (x && x->filter[0] != ZK_NO_FILTER) ? trace[(size_t)x->filter[0] * n + r] : 1;
                const u64 f1 = (x && x->filter[1] != ZK_NO_FILTER) ? trace[(size_t)x->filter[1] * n + r] : 1;
                num[k] = x ? gl_add(gl_mul(f0, x2), gl_mul(f1, x1)) : gl_add(x1, x2); den[k] = gl_mul(x1, x2);
            } else { num[k] = 0; den[k] = 1; }
            pre[k] = acc;
            acc = gl_mul(acc, den[k]);
        }
        if (acc == 0) atomicCAS((unsigned long long *)err, 0ULL, (unsigned long long)(uint32_t)ZK_ERR_ZERO_INVERSE);
        u64 inv = gl_inv_dev(acc);
#pragma unroll
        for (int k = AUX_CHUNK - 1; k >= 0; k--) {
            u64 dinv = gl_mul(inv, pre[k]);
            inv = gl_mul(inv, den[k]);
            if (m0 + k < H) {
                u64 h = gl_mul(num[k], dinv);
                aux[(size_t)(j * H + m0 + k) * n + r] = h;
                total = gl_add(total, h);
            }
        }
    }
    rowsum[(size_t)j * n + r] = total;
}

// The same for NCH challenge sets in ONE thread (round 4): the trace cells a row's helpers need -- four per helper -- are loaded once and
// every challenge set's reciprocals are formed from them.  The kernel above runs a grid row per challenge set and fetched every column
// pair once per set (FETCH 2 x 623 MB for a 93-column trace, profiles/r03e_pmc...: >= 1.6 x the trace + aux bytes).  Plain helpers only
// (no zk_helper_ext: those keep the kernel above); same field operations on the same values, so the same words.
template <int NCH>
__global__ __launch_bounds__(256) void aux_helpers_multi_kernel(const u64 *__restrict__ trace, size_t n, const zk_helper_desc *__restrict__ helpers, u32 H,
                                                                const u64 *__restrict__ chal, u64 *__restrict__ aux, u64 *__restrict__ rowsum, u64 *err) {
    const size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    u64 total[NCH];
#pragma unroll
    for (int j = 0; j < NCH; j++) total[j] = 0;
    for (u32 m0 = 0; m0 < H; m0 += AUX_CHUNK) {
        u64 ta0[AUX_CHUNK], tb0[AUX_CHUNK], ta1[AUX_CHUNK], tb1[AUX_CHUNK];
#pragma unroll
        for (int k = 0; k < AUX_CHUNK; k++) {
            if (m0 + k < H) {
                const zk_helper_desc d = helpers[m0 + k];
                ta0[k] = trace[(size_t)d.a0 * n + r]; tb0[k] = trace[(size_t)d.b0 * n + r];
                ta1[k] = trace[(size_t)d.a1 * n + r]; tb1[k] = trace[(size_t)d.b1 * n + r];
            } else { ta0[k] = tb0[k] = ta1[k] = tb1[k] = 0; }
        }
#pragma unroll
        for (int j = 0; j < NCH; j++) {
            const u64 beta = chal[2 * j], gamma = chal[2 * j + 1];
            u64 num[AUX_CHUNK], den[AUX_CHUNK], pre[AUX_CHUNK];
            u64 acc = 1;
#pragma unroll
            for (int k = 0; k < AUX_CHUNK; k++) {
                if (m0 + k < H) {
                    const u64 x1 = gl_add(gl_mul_add(beta, tb0[k], ta0[k]), gamma), x2 = gl_add(gl_mul_add(beta, tb1[k], ta1[k]), gamma);
                    num[k] = gl_add(x1, x2); den[k] = gl_mul(x1, x2);
                } else { num[k] = 0; den[k] = 1; }
                pre[k] = acc;
                acc = gl_mul(acc, den[k]);
            }
            if (acc == 0) atomicCAS((unsigned long long *)err, 0ULL, (unsigned long long)(uint32_t)ZK_ERR_ZERO_INVERSE);
            u64 inv = gl_inv_dev(acc);
#pragma unroll
            for (int k = AUX_CHUNK - 1; k >= 0; k--) {
                const u64 dinv = gl_mul(inv, pre[k]);
                inv = gl_mul(inv, den[k]);
                if (m0 + k < H) {
                    const u64 h = gl_mul(num[k], dinv);
                    aux[(size_t)(j * H + m0 + k) * n + r] = h;
                    total[j] = gl_add(total[j], h);
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < NCH; j++) rowsum[(size_t)j * n + r] = total[j];
}

// ---- suffix sums over rows: Z[r] = sum_{s >= r} rowsum[s]   (three phases, 2048 rows per workgroup)
#define SCAN_BLOCK 2048
__global__ __launch_bounds__(256) void scan_block_sums_kernel(const u64 *rowsum, size_t n, u64 *blocksum) {
    __shared__ u64 red[256];
    u32 j = blockIdx.y; size_t b = blockIdx.x;
    u64 acc = 0;
    for (u32 i = threadIdx.x; i < SCAN_BLOCK; i += 256) { size_t r = b * SCAN_BLOCK + i; if (r < n) acc = gl_add(acc, rowsum[(size_t)j * n + r]); }
    red[threadIdx.x] = acc; __syncthreads();
    for (u32 s = 128; s > 0; s >>= 1) { if (threadIdx.x < s) red[threadIdx.x] = gl_add(red[threadIdx.x], red[threadIdx.x + s]); __syncthreads(); }
    if (threadIdx.x == 0) blocksum[(size_t)j * gridDim.x + b] = red[0];
}
__global__ void scan_block_suffix_kernel(u64 *blocksum, u32 nb) {   // exclusive suffix: out[b] = sum_{b' > b}
    if (threadIdx.x != 0) return;
    u32 j = blockIdx.x;
    u64 run = 0;
    for (u32 b = nb; b-- > 0;) { u64 v = blocksum[(size_t)j * nb + b]; blocksum[(size_t)j * nb + b] = run; run = gl_add(run, v); }
}
__global__ __launch_bounds__(256) void scan_final_kernel(const u64 *rowsum, size_t n, const u64 *blocksum, u64 *Z /* nch columns of n */) {
    __shared__ u64 sh[SCAN_BLOCK];
    __shared__ u64 part[256];
    u32 j = blockIdx.y; size_t b = blockIdx.x;
    const u32 per = SCAN_BLOCK / 256;
    // thread t owns rows [t*per, (t+1)*per) of this block
    for (u32 i = threadIdx.x; i < SCAN_BLOCK; i += 256) { size_t r = b * SCAN_BLOCK + i; sh[i] = r < n ? rowsum[(size_t)j * n + r] : 0; }
    __syncthreads();
    u64 acc = 0;
    for (u32 i = 0; i < per; i++) acc = gl_add(acc, sh[threadIdx.x * per + i]);
    part[threadIdx.x] = acc; __syncthreads();
    if (threadIdx.x == 0) { u64 run = blocksum[(size_t)j * gridDim.x + b]; for (u32 t = 256; t-- > 0;) { u64 v = part[t]; part[t] = run; run = gl_add(run, v); } }
    __syncthreads();
    u64 run = part[threadIdx.x];
    for (u32 i = per; i-- > 0;) { run = gl_add(run, sh[threadIdx.x * per + i]); sh[threadIdx.x * per + i] = run; }
    __syncthreads();
    for (u32 i = threadIdx.x; i < SCAN_BLOCK; i += 256) { size_t r = b * SCAN_BLOCK + i; if (r < n) Z[(size_t)j * n + r] = sh[i]; }
}

// ---- logUp lookup argument inside a table (starky::lookup::lookup_helper_columns)
// helper_b = sum over a batch of two looked columns of filter / (f + beta); delta = sum_b helper_b - m / (t + beta).  One thread = one
// row and challenge; the reciprocals of a row (two per helper + the table's) are batched four helpers at a time.
struct LookupParams { const zk_lookup_col *cols; u32 n_cols, table_col, freq_col, LH; };
__global__ __launch_bounds__(256) void lookup_helpers_kernel(const u64 *__restrict__ trace, size_t n, LookupParams lk, const u64 *__restrict__ chal,
                                                            u64 *__restrict__ aux, u64 *__restrict__ rowdelta, u64 *err) {
    const size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const u32 j = blockIdx.y;
    if (r >= n) return;
    const u64 beta = chal[2 * j];
    u64 *base = aux + (size_t)j * (lk.LH + 1) * n;
    u64 total = 0;
    for (u32 b0 = 0; b0 < lk.LH; b0 += 4) {
        u64 num[4], den[4], pre[4];
        u64 acc = 1;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            num[k] = 0; den[k] = 1;
            const u32 b = b0 + k;
            if (b < lk.LH) {
                const zk_lookup_col c0 = lk.cols[2 * b];
                const u64 d0 = gl_add(trace[(size_t)c0.col * n + r], beta);
                const u64 f0 = c0.filter == ZK_NO_FILTER ? 1 : gl_canon(trace[(size_t)c0.filter * n + r]);
                if (2 * b + 1 < lk.n_cols) {
                    const zk_lookup_col c1 = lk.cols[2 * b + 1];
                    const u64 d1 = gl_add(trace[(size_t)c1.col * n + r], beta);
                    const u64 f1 = c1.filter == ZK_NO_FILTER ? 1 : gl_canon(trace[(size_t)c1.filter * n + r]);
                    num[k] = gl_add(gl_mul(f0, d1), gl_mul(f1, d0)); den[k] = gl_mul(d0, d1);
                } else { num[k] = f0; den[k] = d0; }
            }
            pre[k] = acc;
            acc = gl_mul(acc, den[k]);
        }
        if (acc == 0) atomicCAS((unsigned long long *)err, 0ULL, (unsigned long long)(uint32_t)ZK_ERR_ZERO_INVERSE);
        u64 inv = gl_inv_dev(acc);
#pragma unroll
        for (int k = 3; k >= 0; k--) {
            const u64 dinv = gl_mul(inv, pre[k]);
            inv = gl_mul(inv, den[k]);
            if (b0 + k < lk.LH) { const u64 h = gl_mul(num[k], dinv); base[(size_t)(b0 + k) * n + r] = h; total = gl_add(total, h); }
        }
    }
    const u64 dt = gl_add(trace[(size_t)lk.table_col * n + r], beta);
    if (dt == 0) atomicCAS((unsigned long long *)err, 0ULL, (unsigned long long)(uint32_t)ZK_ERR_ZERO_INVERSE);
    rowdelta[(size_t)j * n + r] = gl_sub(total, gl_mul(gl_canon(trace[(size_t)lk.freq_col * n + r]), gl_inv_dev(dt)));
}
// forward exclusive prefix sum from the inclusive suffix sums S: Z[r] = sum_{s < r} delta[s] = S[0] - S[r]
__global__ __launch_bounds__(256) void lookup_prefix_from_suffix_kernel(const u64 *S, size_t n, u32 LH, u64 *aux) {
    const size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const u32 j = blockIdx.y;
    if (r >= n) return;
    aux[((size_t)j * (LH + 1) + LH) * n + r] = gl_sub(S[(size_t)j * n], S[(size_t)j * n + r]);
}

static int32_t lookup_columns_dev(zk_ctx *ctx, const zk_lookup *lk, u32 nch, const u64 *d_chal, const u64 *d_trace, size_t n, u64 *d_aux) {
    const u32 LH = (lk->n_cols + 1) / 2;
    ZkOwned own(ctx);
    zk_lookup_col *d_cols = (zk_lookup_col *)own.get((size_t)lk->n_cols * sizeof(zk_lookup_col));
    u64 *rowdelta = (u64 *)own.get(n * nch * sizeof(u64)), *S = (u64 *)own.get(n * nch * sizeof(u64));
    const u32 nb = (u32)((n + SCAN_BLOCK - 1) / SCAN_BLOCK);
    u64 *blocksum = (u64 *)own.get((size_t)nb * nch * sizeof(u64));
    if (!d_cols || !rowdelta || !S || !blocksum) return ctx->fail(ZK_ERR_OOM, "lookup columns: out of device memory");
    int32_t rc = ctx->upload_small(d_cols, lk->cols, (size_t)lk->n_cols * sizeof(zk_lookup_col));
    if (rc) return rc;
    LookupParams p = {d_cols, lk->n_cols, lk->table_col, lk->freq_col, LH};
    ZK_LAUNCH(ctx, "lookup_helpers", lookup_helpers_kernel, dim3((unsigned)((n + 255) / 256), nch), dim3(256), 0, d_trace, n, p, d_chal, d_aux, rowdelta, ctx->d_err);
    ZK_LAUNCH(ctx, "aux_scan_sums", scan_block_sums_kernel, dim3(nb, nch), dim3(256), 0, (const u64 *)rowdelta, n, blocksum);
    ZK_LAUNCH(ctx, "aux_scan_suffix", scan_block_suffix_kernel, dim3(nch), dim3(64), 0, blocksum, nb);
    ZK_LAUNCH(ctx, "aux_scan_final", scan_final_kernel, dim3(nb, nch), dim3(256), 0, (const u64 *)rowdelta, n, (const u64 *)blocksum, S);
    ZK_LAUNCH(ctx, "lookup_prefix", lookup_prefix_from_suffix_kernel, dim3((unsigned)((n + 255) / 256), nch), dim3(256), 0, (const u64 *)S, n, LH, d_aux);
    return ZK_OK;
}

int32_t zk_aux_columns_dev(zk_ctx *ctx, const zk_air *air, const zk_helper_desc *d_helpers, const zk_helper_ext *d_helpers_ext, u32 nch, const u64 *d_chal,
                           const u64 *d_trace, u32 log_n, u64 *d_aux) {
    u32 H = air->n_helpers;
    size_t n = (size_t)1 << log_n;
    if (air->lookup && air->lookup->n_cols) {        // lookup columns first, the CTL-shaped ones behind them (upstream's order)
        int32_t rc = lookup_columns_dev(ctx, air->lookup, nch, d_chal, d_trace, n, d_aux);
        if (rc) return rc;
        d_aux += (size_t)nch * ((air->lookup->n_cols + 1) / 2 + 1) * n;
    }
    if (!H) return ZK_OK;
    u64 *rowsum = (u64 *)ctx->alloc(n * nch * sizeof(u64));
    u32 nb = (u32)((n + SCAN_BLOCK - 1) / SCAN_BLOCK);
    u64 *blocksum = (u64 *)ctx->alloc((size_t)nb * nch * sizeof(u64));
    if (!rowsum || !blocksum) return ZK_ERR_OOM;
    static const int multi = getenv("ZK_AUX_MULTI") ? atoi(getenv("ZK_AUX_MULTI")) : 1;        // 0: a grid row per challenge set (A/B switch)
    if (multi && nch == 2 && !d_helpers_ext)
        ZK_LAUNCH(ctx, "aux_helpers", aux_helpers_multi_kernel<2>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, d_trace, n, d_helpers, H, d_chal, d_aux, rowsum, ctx->d_err);
    else
        ZK_LAUNCH(ctx, "aux_helpers", aux_helpers_kernel, dim3((unsigned)((n + 255) / 256), nch), dim3(256), 0, d_trace, n, d_helpers, d_helpers_ext, H, nch, d_chal,
                  d_aux, rowsum, ctx->d_err);
    ZK_LAUNCH(ctx, "aux_scan_sums", scan_block_sums_kernel, dim3(nb, nch), dim3(256), 0, (const u64 *)rowsum, n, blocksum);
    ZK_LAUNCH(ctx, "aux_scan_suffix", scan_block_suffix_kernel, dim3(nch), dim3(64), 0, blocksum, nb);
    ZK_LAUNCH(ctx, "aux_scan_final", scan_final_kernel, dim3(nb, nch), dim3(256), 0, (const u64 *)rowsum, n, (const u64 *)blocksum,
              d_aux + (size_t)nch * H * n);
    ctx->release(rowsum); ctx->release(blocksum);
    return ZK_OK;
}

// ------------------------------------------------------------------ synthetic witness
__device__ __forceinline__ u64 splitmix_field(u64 st) {
    for (;;) {
        st += 0x9E3779B97F4A7C15ULL;
        u64 z = st;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
        z ^= z >> 31;
        if (z < GL_P) return z;
    }
}
// one thread per column triple (rows are a recurrence: inherently serial, like the EVM interpreter upstream)
__global__ void synth_trace_kernel(u32 n_cols, u32 log_n, u64 seed, u64 *t, u64 *pub) {
    size_t n = (size_t)1 << log_n;
    u32 T = n_cols / 3;
    u32 k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= T + (n_cols - 3 * T)) return;
    if (k < T) {
        u64 *a = t + (size_t)(3 * k) * n, *b = a + n, *c = b + n;
        u64 va = splitmix_field(seed + 3 * k), vb = splitmix_field(seed + 3 * k + 1), vc = splitmix_field(seed + 3 * k + 2);
        for (size_t r = 0; r < n; r++) {
            a[r] = va; b[r] = vb; c[r] = vc;
            u64 ab = gl_mul(va, vb);
            u64 na = gl_add(ab, vc), nb = gl_add(gl_mul(vb, vb), va), nc = gl_add(gl_mul(ab, vc), 1);
            va = na; vb = nb; vc = nc;
        }
    } else {
        u32 col = 3 * T + (k - T);
        u64 *a = t + (size_t)col * n;
        u64 v = splitmix_field(seed + col);
        for (size_t r = 0; r < n; r++) { a[r] = v; v = gl_add(v, 1); }
    }
}
__global__ void synth_pub_kernel(u32 n_cols, u32 log_n, const u64 *t, u64 *pub) {
    size_t n = (size_t)1 << log_n;
    pub[0] = t[0]; pub[1] = n_cols > 1 ? t[n] : 0; pub[2] = t[n - 1];
}

extern "C" int32_t zk_synth_trace_dev(zk_ctx *ctx, uint32_t n_cols, uint32_t log_n, uint64_t seed, uint64_t *d_trace, uint64_t *public_inputs_out) {
    if (!ctx || !d_trace || n_cols == 0 || log_n > 28) return ZK_ERR_INVALID_ARG;
    hipSetDevice(ctx->device);
    u32 work = n_cols / 3 + (n_cols % 3);
    u64 *d_pub = (u64 *)ctx->alloc(3 * sizeof(u64));
    if (!d_pub) return ZK_ERR_OOM;
    ZK_LAUNCH(ctx, "synth_trace", synth_trace_kernel, dim3((work + 63) / 64), dim3(64), 0, n_cols, log_n, seed, (u64 *)d_trace, d_pub);
    ZK_LAUNCH(ctx, "synth_pub", synth_pub_kernel, dim3(1), dim3(1), 0, n_cols, log_n, (const u64 *)d_trace, d_pub);
    int32_t rc = ZK_OK;
    if (public_inputs_out) rc = zk_dev_download(ctx, public_inputs_out, d_pub, 3 * sizeof(u64));
    ctx->release(d_pub);
    return rc;
}
