// Goldilocks radix-2 NTT family for gfx950: batched over column-major polynomials, digit passes staged through LDS.
// Mirrors plonky2_field 0.2.2 fft::{fft_classic, ifft_with_options}, polynomial::{coset_fft, coset_ifft, lde}
// (crate pinned at /root/reference/Cargo.lock:3466-3468; conventions: SURVEY.md A3), reached from the reference at
// /root/reference/ops/src/lib.rs:52 through PolynomialBatch::from_values.
//
// Decomposition.  A size-2^k transform is split into digit passes of width W <= 10.  Each pass loads, per
// workgroup, either T whole rows of 2^W contiguous elements (ROW pass, stride bits S == 0) or a tile of T adjacent
// positions x 2^W strided elements (TILE pass), runs the 2^W-point butterflies in LDS and stores back.
//   DIF engine: natural order in  -> bit-reversed order out  (digit outputs stored bit-reversed, post-twiddles)
//   DIT engine: bit-reversed in   -> natural order out        (digit inputs read bit-reversed, pre-twiddles);
//               its last pass can instead scatter whole rows to bit-reversed positions, which turns a
//               "bit-reversed coefficients -> bit-reversed evaluations" LDE into coalesced row writes.
// No pass ever does a global bit-reversal gather: every global access is a run of >= T*8 contiguous bytes.
#include "zk_internal.h"
#include <stdlib.h>

#define NTT_MAX_W 10
#define NTT_LDS_ELEMS 8192   // 64 KiB of LDS per workgroup -> 2 workgroups per CU

#include "ntt_pass.h"

template <bool DIT, bool ROW>
__global__ __launch_bounds__(256) void ntt_pass_kernel(PassParams p) {
    extern __shared__ u64 sh[];
    const int W = p.W, S = p.S, TL = p.TL;
    const u32 T = 1u << TL, nd = 1u << W;
    const u32 tid = threadIdx.x;
    const size_t n = (size_t)1 << p.k;

    // ---- which instances does this workgroup own
    size_t col = 0, base = 0;            // TILE: base = (outer << (W+S)) + inner0
    size_t row0 = 0, rows_per_col = 0, total_rows = 0;
    if (ROW) {
        rows_per_col = n >> W;
        total_rows = rows_per_col * (size_t)p.n_cols;
        row0 = (size_t)blockIdx.x * T;
    } else {
        size_t tiles_per_block = ((size_t)1 << S) >> TL;                 // tiles across `inner`
        size_t tiles_per_col = (n >> (W + S)) * tiles_per_block;
        col = blockIdx.x / tiles_per_col;
        size_t rem = blockIdx.x % tiles_per_col;
        size_t outer = rem / tiles_per_block, inner0 = (rem % tiles_per_block) << TL;
        base = (outer << (W + S)) + inner0;
    }

    // ---- load (+ optional scale, + DIT pre-twiddle)
    for (u32 idx = tid; idx < (nd << TL); idx += 256) {
        u32 d, t; size_t c, pos;
        if (ROW) {
            t = idx >> W; d = idx & (nd - 1);
            size_t rid = row0 + t;
            if (rid >= total_rows) { sh[idx] = 0; continue; }
            c = rid / rows_per_col;
            pos = ((rid % rows_per_col) << W) + d;
        } else {
            d = idx >> TL; t = idx & (T - 1);
            c = col; pos = base + ((size_t)d << S) + t;
        }
        u64 v = p.in[c * p.in_stride + pos];
        if (p.canon_in) v = gl_canon(v);
        if (p.preA) {
            u32 j = p.pre_bitrev ? bitrev_u32((u32)pos, p.k) : (u32)pos;
            v = gl_mul(v, tab2(p.preA, p.preB, j));
        }
        if (DIT && S > 0) {
            u32 inner = (u32)(pos & (((size_t)1 << S) - 1));
            u32 x = bitrev_u32(d, W) * inner;
            v = gl_mul(v, tab2(p.twA, p.twB, x));
        }
        sh[ROW ? idx : ((d << TL) + t)] = v;
    }
    __syncthreads();

    // ---- 2^W-point butterflies in LDS.  index of (digit slot d, instance t): ROW t*nd + d, TILE d*T + t
    const u32 nbf = (nd >> 1) << TL;
    for (int st = 0; st < W; st++) {
        const int h = DIT ? st : (W - 1 - st);
        const u32 half = 1u << h;
        for (u32 q = tid; q < nbf; q += 256) {
            u32 bq, t;
            if (ROW) { t = q >> (W - 1); bq = q & ((nd >> 1) - 1); } else { bq = q >> TL; t = q & (T - 1); }
            u32 low = bq & (half - 1), high = bq >> h;
            u32 d0 = (high << (h + 1)) | low, d1 = d0 + half;
            u32 i0 = ROW ? (t << W) + d0 : (d0 << TL) + t;
            u32 i1 = ROW ? (t << W) + d1 : (d1 << TL) + t;
            u64 tw = p.tw_digit[low << (W - 1 - h)];
            u64 u = sh[i0], v = sh[i1];
            if (DIT) {
                v = gl_mul(v, tw);
                sh[i0] = gl_add(u, v); sh[i1] = gl_sub(u, v);
            } else {
                sh[i0] = gl_add(u, v); sh[i1] = gl_mul(gl_sub(u, v), tw);
            }
        }
        __syncthreads();
    }

    // ---- store (+ DIF post-twiddle, + optional scale)
    for (u32 idx = tid; idx < (nd << TL); idx += 256) {
        u32 slot, t, e; size_t c, pos; u32 lds;
        if (ROW) {
            t = idx >> W; slot = idx & (nd - 1);
            size_t rid = row0 + t;
            if (rid >= total_rows) continue;
            c = rid / rows_per_col;
            e = DIT ? slot : bitrev_u32(slot, W);
            lds = idx;
            if (DIT && p.permute_out) {
                // single-pass transform (S == 0, W == k): natural index e -> position bitrev(e) == slot' where we read lds e
                pos = ((rid % rows_per_col) << W) + slot; e = bitrev_u32(slot, W); lds = (t << W) + e;
            } else pos = ((rid % rows_per_col) << W) + slot;
        } else {
            c = col;
            if (DIT && p.permute_out) {
                // K = e*2^S + inner  ->  position bitrev_S(inner)*2^W + bitrev_W(e); iterate with bitrev_W(e) fastest
                t = idx >> W; u32 pe = idx & (nd - 1);
                e = bitrev_u32(pe, W);
                u32 inner = (u32)(base & (((size_t)1 << S) - 1)) + t;
                pos = ((size_t)bitrev_u32(inner, S) << W) + pe;
                lds = (e << TL) + t;
            } else {
                slot = idx >> TL; t = idx & (T - 1);
                e = DIT ? slot : bitrev_u32(slot, W);
                pos = base + ((size_t)slot << S) + t;
                lds = idx;
            }
        }
        u64 v = sh[lds];
        if (!DIT && S > 0) {
            u32 inner = (u32)(pos & (((size_t)1 << S) - 1));
            v = gl_mul(v, tab2(p.twA, p.twB, e * inner));
        }
        if (p.postA) {
            // natural output index of this element
            u32 K;
            if (ROW) K = (u32)(((pos >> W) << W) + e);
            else K = (u32)(((base >> (W + S)) << (W + S)) + ((size_t)e << S) + ((pos & (((size_t)1 << S) - 1))));
            if (DIT && p.permute_out && !ROW) K = (u32)(((size_t)e << S) + (u32)(base & (((size_t)1 << S) - 1)) + t);
            v = gl_mul(v, tab2(p.postA, p.postB, K));
        }
        if (p.post_const) v = gl_mul(v, p.post_const);
        p.out[c * p.out_stride + pos] = v;
    }
}

__global__ void bitrev_copy_kernel(const u64 *in, size_t in_stride, u64 *out, size_t out_stride, int k, int n_cols) {
    size_t n = (size_t)1 << k;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * (size_t)n_cols) return;
    size_t c = i >> k, pos = i & (n - 1);
    out[c * out_stride + pos] = gl_canon(in[c * in_stride + bitrev_u32((u32)pos, k)]);
}

// ------------------------------------------------------------------ host: tables
static u64 *upload(zk_ctx *ctx, const std::vector<u64> &h) {
    u64 *d = (u64 *)ctx->alloc(h.size() * sizeof(u64));
    if (!d) return nullptr;
    // tables are built rarely; a synchronous copy (through the context's pinned bounce buffer) keeps the host vector's lifetime trivial
    if (ctx->copy_h2d_sync(d, h.data(), h.size() * sizeof(u64)) != ZK_OK) { ctx->release(d); return nullptr; }
    return d;
}

u64 *zk_full_digit_table(zk_ctx *ctx, int W, int inverse) {
    auto key = std::make_pair(W + 100, inverse);
    auto it = ctx->digit_tw.find(key);
    if (it != ctx->digit_tw.end()) return it->second;
    u64 w = gl_root(W);
    if (inverse) w = gl_inv(w);
    std::vector<u64> h((size_t)1 << W);
    u64 x = 1;
    for (size_t i = 0; i < h.size(); i++) { h[i] = x; x = gl_mul(x, w); }
    u64 *d = upload(ctx, h);
    ctx->digit_tw[key] = d;
    return d;
}

u64 *zk_digit_twiddles(zk_ctx *ctx, int W, int inverse) {
    auto key = std::make_pair(W, inverse);
    auto it = ctx->digit_tw.find(key);
    if (it != ctx->digit_tw.end()) return it->second;
    u64 w = gl_root(W);
    if (inverse) w = gl_inv(w);
    size_t cnt = W ? ((size_t)1 << (W - 1)) : 1;
    std::vector<u64> h(cnt);
    u64 x = 1;
    for (size_t i = 0; i < cnt; i++) { h[i] = x; x = gl_mul(x, w); }
    u64 *d = upload(ctx, h);
    ctx->digit_tw[key] = d;
    return d;
}

static TwoLevel make_two_level(zk_ctx *ctx, u64 base, u64 premul, size_t count) {
    // value(x) = premul * base^x = A[x & 2047] * B[x >> 11]
    size_t na = count < 2048 ? count : 2048, nb = (count + 2047) >> 11;
    if (na == 0) na = 1;
    if (nb == 0) nb = 1;
    std::vector<u64> A(2048, 0), B(nb);
    u64 x = 1;
    for (size_t i = 0; i < 2048; i++) { A[i] = x; x = gl_mul(x, base); }   // x ends as base^2048
    u64 step = x, y = premul;
    for (size_t i = 0; i < nb; i++) { B[i] = y; y = gl_mul(y, step); }
    TwoLevel t; t.A = upload(ctx, A); t.B = upload(ctx, B);
    return t;
}

TwoLevel zk_root_table(zk_ctx *ctx, int m, int inverse) {
    auto key = std::make_pair(m, inverse);
    auto it = ctx->root_tab.find(key);
    if (it != ctx->root_tab.end()) return it->second;
    u64 w = gl_root(m);
    if (inverse) w = gl_inv(w);
    TwoLevel t = make_two_level(ctx, w, 1, (size_t)1 << m);
    ctx->root_tab[key] = t;
    return t;
}

TwoLevel zk_scale_table(zk_ctx *ctx, u64 base, u64 premul, int log_count) {
    auto key = std::make_tuple(base, premul, log_count);
    auto it = ctx->scale_tab.find(key);
    if (it != ctx->scale_tab.end()) return it->second;
    TwoLevel t = make_two_level(ctx, base, premul, (size_t)1 << log_count);
    ctx->scale_tab[key] = t;
    return t;
}

// ------------------------------------------------------------------ host: pass schedules
// k <= 12: ONE pass over whole rows (the register-radix kernel does 2^11 / 2^12 points as three radix-16 phases; ZK_NTT_ROW_W=10 brings
// back the two-pass split for an A/B run); beyond that, passes of at most 10 bits.
static void digit_widths(int k, int *w, int *np) {
    static const int row_w = getenv("ZK_NTT_ROW_W") ? atoi(getenv("ZK_NTT_ROW_W")) : 12;
    if (k > NTT_MAX_W && k <= row_w && k <= 12) { w[0] = k; *np = 1; return; }
    int passes = (k + NTT_MAX_W - 1) / NTT_MAX_W;
    if (passes < 1) passes = 1;
    int base = k / passes, extra = k % passes;
    for (int i = 0; i < passes; i++) w[i] = base + (i < extra ? 1 : 0);
    *np = passes;
    // experiment switch: ZK_NTT_W0 = digit width of the TILE pass of a two-pass transform (the ROW pass takes the rest, up to 12 bits:
    // 9 + 11 at k = 20 gives the TILE pass runs of 16 elements = whole 128-byte lines instead of 8)
    static const int w0 = getenv("ZK_NTT_W0") ? atoi(getenv("ZK_NTT_W0")) : 0;
    if (w0 >= 4 && passes == 2 && k - w0 >= 4 && k - w0 <= 12 && w0 <= NTT_MAX_W && k - w0 >= 13 - w0) { w[0] = w0; w[1] = k - w0; }
}

template <bool DIT, bool ROW>
static void launch_pass(zk_ctx *ctx, PassParams &p, const char *name) {
    if (p.W >= 4 && !getenv("ZK_NTT_SLOW")) {
        p.tw_full = zk_full_digit_table(ctx, p.W, p.inverse);
        if (zk_ntt_fast_pass(ctx, p, DIT, ROW, name)) return;
    }
    size_t n = (size_t)1 << p.k;
    size_t blocks;
    if (ROW) {
        size_t rows = (n >> p.W) * (size_t)p.n_cols;
        blocks = (rows + ((size_t)1 << p.TL) - 1) >> p.TL;
    } else blocks = ((n >> p.W) >> p.TL) * (size_t)p.n_cols;
    size_t shmem = ((size_t)1 << (p.W + p.TL)) * sizeof(u64);
    ZK_LAUNCH(ctx, name, (ntt_pass_kernel<DIT, ROW>), dim3((unsigned)blocks), dim3(256), shmem, p);
}

static int tile_log(int W, int S, bool row, size_t rows_total) {
    int tl = 13 - W;                      // NTT_LDS_ELEMS = 2^13
    if (!row && tl > S) tl = S;
    if (row) { while (tl > 0 && ((size_t)1 << tl) > rows_total) tl--; }
    if (tl < 0) tl = 0;
    return tl;
}

int32_t zk_ntt_dif(zk_ctx *ctx, const u64 *in, size_t in_stride, u64 *out, size_t out_stride, int k, int n_cols,
                   int inverse, const TwoLevel *pre, u64 post_const, int canon_in) {
    if (k < 0 || k > 30 || n_cols <= 0) return ctx->fail(ZK_ERR_INVALID_ARG, "ntt_dif: bad size");
    int w[4], np; digit_widths(k, w, &np);
    // experiment switch (VERDICT r2 #3b): both passes of a column GROUP back to back, so that the second pass finds the first one's output in
    // the 256 MiB Infinity Cache.  Off by default: the passes are bound by instruction issue, not by HBM.
    static const int col_group = getenv("ZK_NTT_COL_GROUP") ? atoi(getenv("ZK_NTT_COL_GROUP")) : 0;
    if (col_group > 0 && np > 1 && n_cols > col_group) {
        for (int c0 = 0; c0 < n_cols; c0 += col_group) {
            int32_t rc = zk_ntt_dif(ctx, in + (size_t)c0 * in_stride, in_stride, out + (size_t)c0 * out_stride, out_stride, k, n_cols - c0 < col_group ? n_cols - c0 : col_group,
                                    inverse, pre, post_const, canon_in);
            if (rc) return rc;
        }
        return ZK_OK;
    }
    int S = k;
    for (int i = 0; i < np; i++) {
        S -= w[i];
        PassParams p = {};
        p.in = i == 0 ? in : out; p.in_stride = i == 0 ? in_stride : out_stride;
        p.out = out; p.out_stride = out_stride;
        p.k = k; p.W = w[i]; p.S = S; p.n_cols = n_cols;
        bool row = S == 0;
        p.TL = tile_log(w[i], S, row, (((size_t)1 << k) >> w[i]) * (size_t)n_cols);
        p.tw_digit = zk_digit_twiddles(ctx, w[i], inverse); p.inverse = inverse;
        if (S > 0) { TwoLevel t = zk_root_table(ctx, w[i] + S, inverse); p.twA = t.A; p.twB = t.B; }
        if (i == 0 && pre) { p.preA = pre->A; p.preB = pre->B; }
        if (i == np - 1) p.post_const = post_const;
        p.canon_in = (i == 0) ? canon_in : 0;
        if (row) launch_pass<false, true>(ctx, p, "ntt_dif_row"); else launch_pass<false, false>(ctx, p, "ntt_dif_tile");
    }
    return ZK_OK;
}

int32_t zk_ntt_dit(zk_ctx *ctx, const u64 *in, size_t in_stride, u64 *out, size_t out_stride, int k, int n_cols,
                   int inverse, const TwoLevel *pre, const TwoLevel *post, int permute_out, u64 *tmp, size_t tmp_stride) {
    if (k < 0 || k > 30 || n_cols <= 0) return ctx->fail(ZK_ERR_INVALID_ARG, "ntt_dit: bad size");
    int w[4], np; digit_widths(k, w, &np);
    if (permute_out && np > 1 && !tmp) return ctx->fail(ZK_ERR_INVALID_ARG, "ntt_dit: permuted output needs a scratch buffer");
    static const int col_group = getenv("ZK_NTT_COL_GROUP") ? atoi(getenv("ZK_NTT_COL_GROUP")) : 0;      // see zk_ntt_dif
    if (col_group > 0 && np > 1 && n_cols > col_group) {
        for (int c0 = 0; c0 < n_cols; c0 += col_group) {
            int32_t rc = zk_ntt_dit(ctx, in + (size_t)c0 * in_stride, in_stride, out + (size_t)c0 * out_stride, out_stride, k, n_cols - c0 < col_group ? n_cols - c0 : col_group,
                                    inverse, pre, post, permute_out, tmp ? tmp + (size_t)c0 * tmp_stride : nullptr, tmp_stride);
            if (rc) return rc;
        }
        return ZK_OK;
    }
    // DIT processes the lowest position digit first; use the reversed width order so the last (TILE) pass is the widest-stride one
    int S = 0;
    for (int i = 0; i < np; i++) {
        int W = w[np - 1 - i];
        PassParams p = {};
        bool last = i == np - 1;
        // data flow: in -> (tmp if permuting and more than one pass, else out) ... last pass -> out
        u64 *mid = (permute_out && np > 1) ? tmp : out;
        size_t mid_stride = (permute_out && np > 1) ? tmp_stride : out_stride;
        p.in = i == 0 ? in : mid; p.in_stride = i == 0 ? in_stride : mid_stride;
        p.out = last ? out : mid; p.out_stride = last ? out_stride : mid_stride;
        p.k = k; p.W = W; p.S = S; p.n_cols = n_cols;
        bool row = S == 0;
        p.TL = tile_log(W, S, row, (((size_t)1 << k) >> W) * (size_t)n_cols);
        p.tw_digit = zk_digit_twiddles(ctx, W, inverse); p.inverse = inverse;
        if (S > 0) { TwoLevel t = zk_root_table(ctx, W + S, inverse); p.twA = t.A; p.twB = t.B; }
        if (i == 0 && pre) { p.preA = pre->A; p.preB = pre->B; p.pre_bitrev = 1; }
        if (last && post) { p.postA = post->A; p.postB = post->B; }
        p.permute_out = last ? permute_out : 0;
        if (row) launch_pass<true, true>(ctx, p, "ntt_dit_row"); else launch_pass<true, false>(ctx, p, "ntt_dit_tile");
        S += W;
    }
    return ZK_OK;
}

// lde(rate_bits).coset_fft(g) of bit-reversed coefficient columns: the size-L transform of zero-padded coefficients is 2^rate_bits size-n
// transforms on the cosets g * w_L^par, coset par filling block bitrev(par) of every (bit-reversed) LDE column.  Small batches -- the
// Plonk half's circuits, the narrow tables -- run ALL cosets in one launch per pass (virtual columns, ntt_fast_kernel<.., CS>): a single
// coset of 135 wires x 2^12 rows is 67 workgroups on 256 compute units, and an LDE with rate_bits 3 used to be 16 such launches.
// tmp (n_cols x n, stride tmp_stride): scratch of the coset-by-coset path; the merged path brings its own.
int32_t zk_ntt_lde(zk_ctx *ctx, const u64 *coeffs, size_t in_stride, u64 *lde, size_t out_stride, int log_n, int n_cols, int rate_bits,
                   u64 *tmp, size_t tmp_stride) {
    if (log_n < 0 || log_n > 30 || n_cols <= 0 || rate_bits < 0 || rate_bits > 4) return ctx->fail(ZK_ERR_INVALID_ARG, "ntt_lde: bad size");
    const size_t n = (size_t)1 << log_n;
    const u32 P = 1u << rate_bits;
    const u64 wL = gl_root(log_n + rate_bits);
    int w[4], np; digit_widths(log_n, w, &np);
    static const int merge = getenv("ZK_NTT_COSET_MERGE") ? atoi(getenv("ZK_NTT_COSET_MERGE")) : 1;
    bool merged = merge && rate_bits >= 1 && log_n >= 4 && log_n <= 20 && np <= 2 && !getenv("ZK_NTT_SLOW") &&
                  ((n >> 13) * (size_t)n_cols < 4096 || merge >= 2);   // a launch per coset already fills the chip beyond this (2: merge anyway, experiment)
    if (merged && np == 2 && w[1] < 13 - w[0]) merged = false;          // the TILE pass would not take the register-radix kernel
    if (merged) {
        u64 *mid = nullptr;
        const size_t mid_stride = n * P;
        if (np > 1) { mid = (u64 *)ctx->alloc(mid_stride * (size_t)n_cols * sizeof(u64)); if (!mid) return ZK_ERR_OOM; }
        int S = 0;
        int32_t rc = ZK_OK;
        for (int i = 0; i < np && rc == ZK_OK; i++) {
            const int W = w[np - 1 - i];
            const bool last = i == np - 1, row = S == 0;
            PassParams p = {};
            p.in = i == 0 ? coeffs : mid; p.in_stride = i == 0 ? in_stride : mid_stride; p.in_sub = i == 0 ? 0 : n;
            p.out = last ? lde : mid; p.out_stride = last ? out_stride : mid_stride; p.out_sub = n;
            p.k = log_n; p.W = W; p.S = S; p.n_cols = n_cols; p.cs_bits = rate_bits;
            p.tw_digit = zk_digit_twiddles(ctx, W, 0); p.tw_full = zk_full_digit_table(ctx, W, 0);
            if (S > 0) { TwoLevel t = zk_root_table(ctx, W + S, 0); p.twA = t.A; p.twB = t.B; }
            NttCosetPre cpre = {};
            if (i == 0) {
                for (u32 h = 0; h < P; h++) {
                    TwoLevel pre = zk_scale_table(ctx, gl_mul(GL_GEN, gl_pow(wL, bitrev_u32(h, rate_bits))), 1, log_n);
                    cpre.A[h] = pre.A; cpre.B[h] = pre.B;
                }
                p.pre_bitrev = 1;
            }
            p.permute_out = last ? 1 : 0;
            if (!zk_ntt_fast_pass(ctx, p, true, row, row ? "ntt_dit_row" : "ntt_dit_tile", i == 0 ? &cpre : nullptr)) rc = ctx->fail(ZK_ERR_OOM, "ntt_lde: merged pass not launched");
            S += W;
        }
        if (mid) ctx->release(mid);
        return rc;
    }
    int32_t rc = ZK_OK;
    for (u32 par = 0; rc == ZK_OK && par < P; par++) {
        TwoLevel pre = zk_scale_table(ctx, gl_mul(GL_GEN, gl_pow(wL, par)), 1, log_n);
        rc = zk_ntt_dit(ctx, coeffs, in_stride, lde + (size_t)bitrev_u32(par, rate_bits) * n, out_stride, log_n, n_cols, 0, &pre, nullptr, 1, tmp, tmp_stride);
    }
    return rc;
}

int32_t zk_bitrev_copy(zk_ctx *ctx, const u64 *in, size_t in_stride, u64 *out, size_t out_stride, int k, int n_cols) {
    size_t total = ((size_t)1 << k) * (size_t)n_cols;
    ZK_LAUNCH(ctx, "bitrev_copy", bitrev_copy_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, in, in_stride, out, out_stride, k, n_cols);
    return ZK_OK;
}

__global__ void scale_cols_kernel(u64 *data, size_t stride, int k, int n_cols, const u64 *A, const u64 *B) {
    size_t n = (size_t)1 << k;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * (size_t)n_cols) return;
    size_t c = i >> k; u32 j = (u32)(i & (n - 1));
    data[c * stride + j] = gl_mul(data[c * stride + j], tab2(A, B, j));
}
int32_t zk_scale_cols(zk_ctx *ctx, u64 *data, size_t stride, int k, int n_cols, const TwoLevel *sc) {
    size_t total = ((size_t)1 << k) * (size_t)n_cols;
    ZK_LAUNCH(ctx, "scale_cols", scale_cols_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, data, stride, k, n_cols, (const u64 *)sc->A, (const u64 *)sc->B);
    return ZK_OK;
}

// ------------------------------------------------------------------ C ABI: fft_with_options / ifft_with_options / coset variants
__global__ void ntt_zero_tail_kernel(u64 *d, size_t n, size_t keep, size_t total) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total && (i & (n - 1)) >= keep) d[i] = 0;
}
// fft_with_options / ifft_with_options with zero_factor = Some(r): the caller promises that only the first n >> r inputs of every column are
// non-zero, and upstream then skips the first r butterfly layers (they would only copy) -- i.e. it computes the transform of the input
// with its tail TAKEN AS zero.  Forward: the size-n transform of zero-padded coefficients is 2^r transforms of size n >> r on the cosets
// shift * w_n^par (the same identity the commit path uses for its LDE), each written where the bit-reversed order wants it.
extern "C" int32_t zk_ntt_batch_opts_dev(zk_ctx *ctx, uint64_t *d_data, uint32_t log_n, uint32_t n_cols, int32_t inverse, uint64_t shift, uint32_t zero_factor) {
    if (!ctx || !d_data || log_n > 30 || n_cols == 0 || zero_factor > log_n) return ZK_ERR_INVALID_ARG;
    hipSetDevice(ctx->device);
    size_t n = (size_t)1 << log_n;
    shift = gl_canon(shift);
    u64 *tmp = (u64 *)ctx->alloc(n * n_cols * sizeof(u64));
    if (!tmp) return ZK_ERR_OOM;
    int32_t rc = ZK_OK;
    if (zero_factor && inverse) {       // rare: give the ignored tail its promised value, then the full inverse transform
        const size_t total = n * n_cols;
        ZK_LAUNCH(ctx, "ntt_zero_tail", ntt_zero_tail_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (u64 *)d_data, n, n >> zero_factor, total);
    }
    if (!inverse && zero_factor) {
        const u32 r = zero_factor, ks = log_n - r;
        const size_t ns = n >> r;
        const u64 w = gl_root(log_n), sh = shift ? shift : 1;
        for (u32 par = 0; rc == ZK_OK && par < (1u << r); par++) {
            TwoLevel pre = zk_scale_table(ctx, gl_mul(sh, gl_pow(w, par)), 1, ks);
            rc = zk_ntt_dif(ctx, (const u64 *)d_data, n, tmp + (size_t)bitrev_u32(par, r) * ns, n, ks, n_cols, 0, &pre, 0, 1);
        }
        if (rc == ZK_OK) rc = zk_bitrev_copy(ctx, tmp, n, (u64 *)d_data, n, log_n, n_cols);
    } else if (!inverse) {
        TwoLevel pre; bool has = shift > 1;
        if (has) pre = zk_scale_table(ctx, shift, 1, log_n);
        rc = zk_ntt_dif(ctx, (const u64 *)d_data, n, tmp, n, log_n, n_cols, 0, has ? &pre : nullptr, 0, 1);
        if (rc == ZK_OK) rc = zk_bitrev_copy(ctx, tmp, n, (u64 *)d_data, n, log_n, n_cols);
    } else {
        // c[j] = shift^-j n^-1 sum_i v[i] w^-ij : DIF with inverse roots, then un-permute, scaling folded into a DIT-free post step
        u64 ninv = gl_inv((u64)n % GL_P);
        rc = zk_ntt_dif(ctx, (const u64 *)d_data, n, tmp, n, log_n, n_cols, 1, nullptr, (shift > 1) ? 0 : ninv, 1);
        if (rc == ZK_OK) rc = zk_bitrev_copy(ctx, tmp, n, (u64 *)d_data, n, log_n, n_cols);
        if (rc == ZK_OK && shift > 1) {
            // multiply natural-order coefficient j by ninv * shift^-j: run it as a 0-bit "transform" would be wasteful; reuse DIF pre-scale on a copy
            TwoLevel sc = zk_scale_table(ctx, gl_inv(shift), ninv, log_n);
            rc = zk_scale_cols(ctx, (u64 *)d_data, n, log_n, n_cols, &sc);
        }
    }
    ctx->release(tmp);
    return rc;
}
extern "C" int32_t zk_ntt_batch_dev(zk_ctx *ctx, uint64_t *d_data, uint32_t log_n, uint32_t n_cols, int32_t inverse, uint64_t shift) {
    return zk_ntt_batch_opts_dev(ctx, d_data, log_n, n_cols, inverse, shift, 0);
}

extern "C" int32_t zk_ntt_batch_opts(zk_ctx *ctx, uint64_t *data, uint32_t log_n, uint32_t n_cols, int32_t inverse, uint64_t shift, uint32_t zero_factor) {
    if (!ctx || !data || log_n > 30 || n_cols == 0 || zero_factor > log_n) return ZK_ERR_INVALID_ARG;
    hipSetDevice(ctx->device);
    size_t bytes = ((size_t)n_cols << log_n) * sizeof(u64);
    u64 *d = (u64 *)ctx->alloc(bytes);
    if (!d) return ZK_ERR_OOM;
    int32_t rc = zk_dev_upload(ctx, d, data, bytes);
    if (rc == ZK_OK) rc = zk_ntt_batch_opts_dev(ctx, d, log_n, n_cols, inverse, shift, zero_factor);
    if (rc == ZK_OK) rc = zk_dev_download(ctx, data, d, bytes);
    ctx->release(d);
    return rc;
}
extern "C" int32_t zk_ntt_batch(zk_ctx *ctx, uint64_t *data, uint32_t log_n, uint32_t n_cols, int32_t inverse, uint64_t shift) {
    return zk_ntt_batch_opts(ctx, data, log_n, n_cols, inverse, shift, 0);
}
