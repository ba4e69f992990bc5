// Register-radix NTT digit pass: host side (pass tables, shape checks) and the DIF kernel family.  Kernel: ntt_fast_kernel.cuh.
#include "ntt_fast_kernel.cuh"
#include <map>
#include <tuple>
#include <stdlib.h>

bool zk_ntt_fast_launch_dif(zk_ctx *ctx, const PassParams &p, bool row, const char *name) { return ntt_fast_launch_family<false, false>(ctx, p, row, name); }

// per-position factor tables: exactly the factors the kernel would otherwise form from two-level tables at run time
__global__ void build_pass_table_kernel(PassParams p, int dit, int store, u64 *out, int tab_bits) {
    size_t pos = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t n = (size_t)1 << tab_bits;                         // the column, or one period 2^(W+S) of the inter-pass factors
    if (pos >= n) return;
    const int W = p.W, S = p.S;
    u32 slot = (u32)((pos >> S) & (((size_t)1 << W) - 1)), inner = (u32)(pos & (((size_t)1 << S) - 1));
    u64 f = 1;
    if (!store) {
        if (p.preA) { u32 j = p.pre_bitrev ? bitrev_u32(pos, p.k) : pos; f = gl_mul(f, tab2(p.preA, p.preB, j)); }
        if (dit && S > 0) f = gl_mul(f, tab2(p.twA, p.twB, bitrev_u32(slot, W) * inner));
    } else {
        u32 e = dit ? slot : bitrev_u32(slot, W);
        if (!dit && S > 0) f = gl_mul(f, tab2(p.twA, p.twB, e * inner));
        if (p.postA) f = gl_mul(f, tab2(p.postA, p.postB, (u32)pos));     // DIT output position == natural index
        if (p.post_const) f = gl_mul(f, p.post_const);
    }
    out[pos] = f;
}
// Without an input / output scale the factors of a pass depend on the position modulo 2^(W+S) only: the table is that long.  (Up to round 4
// every table had 2^k entries and transforms beyond 2^24 were refused here -- 128 MiB per table and more -- and fell to the LDS radix-2
// kernel.  Now the middle pass of a tall transform reads a table of 2^18 entries or so, which stays in L2; its first and last pass
// still need 2^k entries: 2 GiB at k = 28, a few of them per shape, out of 288 GB.)
static int pass_table_bits(const PassParams &p, bool store) {
    const bool scaled = store ? (p.postA != nullptr || p.post_const != 0) : p.preA != nullptr;
    return scaled ? p.k : (p.W + p.S < p.k ? p.W + p.S : p.k);
}
static const u64 *pass_table(zk_ctx *ctx, const PassParams &p, bool dit, bool store) {
    if (p.k > 28) return nullptr;
    char key[192];
    snprintf(key, sizeof key, "%d.%d.%d.%d.%d.%p.%d.%p.%llx.%d", p.k, p.W, p.S, (int)dit, p.inverse, (const void *)p.preA, p.pre_bitrev,
             (const void *)p.postA, (unsigned long long)p.post_const, (int)store);
    auto &cache = ctx->table_cache;
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    const int bits = pass_table_bits(p, store);
    size_t n = (size_t)1 << bits;
    u64 *t = (u64 *)ctx->alloc(n * sizeof(u64));
    if (!t) return nullptr;
    ZK_LAUNCH(ctx, "ntt_build_table", build_pass_table_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, p, (int)dit, (int)store, t, bits);
    cache[key] = t;
    return t;
}
// the input factors of a coset-merged first pass: slice h = the table the pass would use for the input scale of slot h
static const u64 *pass_table_cosets(zk_ctx *ctx, const PassParams &p, bool dit, const NttCosetPre &pre) {
    if (p.k > 20) return nullptr;
    char key[256];
    snprintf(key, sizeof key, "cs.%d.%d.%d.%d.%d.%p.%p.%d", p.k, p.W, p.S, (int)dit, p.cs_bits, (const void *)pre.A[0], (const void *)pre.A[1], p.pre_bitrev);
    auto &cache = ctx->table_cache;
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    size_t n = (size_t)1 << p.k;
    u64 *t = (u64 *)ctx->alloc((n << p.cs_bits) * sizeof(u64));
    if (!t) return nullptr;
    for (u32 h = 0; h < (1u << p.cs_bits); h++) {
        PassParams q = p;
        q.preA = pre.A[h]; q.preB = pre.B[h];
        ZK_LAUNCH(ctx, "ntt_build_table", build_pass_table_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, q, (int)dit, 0, t + (size_t)h * n, p.k);
    }
    cache[key] = t;
    return t;
}

bool zk_ntt_fast_pass(zk_ctx *ctx, const PassParams &p0, bool dit, bool row, const char *name, const NttCosetPre *cs_pre) {
    if (p0.W < 4 || p0.W > (row ? 12 : 10) || !p0.tw_full || p0.k > 30) return false;
    if (!row && p0.S < 13 - p0.W) return false;            // a TILE pass needs 2^(13-W) adjacent positions: smaller transforms take the LDS radix-2 kernel
    const size_t vcols = (size_t)p0.n_cols << p0.cs_bits;
    if (((((size_t)1 << p0.k) >> p0.W) * vcols + 64) >> 32) return false;      // the kernel numbers rows in 32 bits
    if (p0.cs_bits && (!dit || p0.inverse || p0.postA || p0.post_const || p0.cs_bits > 4)) return false;
    PassParams p = p0;
    p.TL = 13 - p.W;                                        // fixed tile shape (ROW tiles past the last row are skipped inside the kernel)
    p.force_redo = (int)ctx->force_redo;
    bool need_load = p.preA || cs_pre || (dit && p.S > 0);
    bool need_store = !p.permute_out && ((!dit && p.S > 0) || p.postA || p.post_const);
    // an output scale that is one constant for every position (the 1/n of an inverse transform's last pass) needs no table
    const bool store_is_const = need_store && !(!dit && p.S > 0) && !p.postA;
    p.load_tab_mask = p.store_tab_mask = ~0u;
    // the DIT post table is indexed by the natural output index, which equals the output position only when not permuting
    static const int use_tab = getenv("ZK_NTT_TAB") ? atoi(getenv("ZK_NTT_TAB")) : 3;       // bit0: ROW passes, bit1: TILE passes
    static const int use_colfast = getenv("ZK_NTT_COLFAST") ? atoi(getenv("ZK_NTT_COLFAST")) : 3;
    bool tab_on = row ? (use_tab & 1) : (use_tab & 2);
    if (p.cs_bits) {
        // per-slot input factors live in one table of 2^cs_bits slices (the kernel has no two-level path for them)
        if (cs_pre) { p.load_tab = pass_table_cosets(ctx, p, dit, *cs_pre); p.tab_sub = (size_t)1 << p.k; if (!p.load_tab) return false; }
        else if (need_load) { p.load_tab = pass_table(ctx, p, dit, false); p.tab_sub = 0; if (!p.load_tab) return false; }
    } else if (need_load && tab_on) p.load_tab = pass_table(ctx, p, dit, false);
    if (store_is_const) { p.store_const = p.post_const; need_store = false; }
    if (need_store && tab_on) p.store_tab = pass_table(ctx, p, dit, true);
    if (!row) {
        if (p.load_tab && !p.cs_bits) p.load_tab_mask = (unsigned)(((size_t)1 << pass_table_bits(p, false)) - 1);
        if (p.store_tab) p.store_tab_mask = (unsigned)(((size_t)1 << pass_table_bits(p, true)) - 1);
    }
    // the kernel takes its per-position factors from ONE coalesced table per side or not at all: a pass whose table is not available
    // (k > 28; ZK_NTT_TAB switched off) and a permuting pass with an output scale go to the LDS radix-2 kernel
    if ((need_load && !p.load_tab) || (need_store && !p.store_tab)) return false;
    if (p.permute_out && (p.postA || p.post_const)) return false;
    size_t rows_per_col = ((size_t)1 << p.k) >> p.W;
    {
        // the kernel's addressing (ntt_fast_kernel.cuh): a tile is either inside one column (TILE passes; ROW passes with at least T rows
        // per column) or holds T whole one-row columns (k == W); the per-thread part of an address is a 32-bit byte offset
        const size_t T = (size_t)1 << p.TL;
        if (row && rows_per_col != 1 && rows_per_col < T) return false;
        if (p.k > 28) return false;
        if (row && rows_per_col == 1) {
            const size_t st = p.in_stride > p.out_stride ? p.in_stride : p.out_stride;
            size_t sub = p.in_sub > p.out_sub ? p.in_sub : p.out_sub;
            if (p.tab_sub > sub) sub = p.tab_sub;
            if (((T * st + (((size_t)1 << p.cs_bits) - 1) * sub + ((size_t)1 << p.W)) >> 29) != 0) return false;
        }
    }
    // passes that move more than the Infinity Cache holds write whole lines with the non-temporal hint (kernel: ntt_st)
    static const int use_nt = getenv("ZK_NTT_NT") ? atoi(getenv("ZK_NTT_NT")) : 1;
    p.nt_store = (use_nt && ((((size_t)1 << p.k) * vcols * sizeof(u64)) >> 29) != 0) ? 1u : 0u;
    bool cf_on = row ? (use_colfast & 1) : (use_colfast & 2);
    p.col_fast = cf_on ? (row ? (rows_per_col >= ((size_t)1 << p.TL) ? 1 : 0) : 1) : 0;
    if (!row && p.col_fast) {
        // runs shorter than a 128-byte line: pair the two workgroups that share lines on one XCD (kernel comment); needs whole groups of 16
        static const int xcd_pair = getenv("ZK_NTT_XCD_PAIR") ? atoi(getenv("ZK_NTT_XCD_PAIR")) : 1;
        const size_t tiles_per_col = (((size_t)1 << p.k) >> p.W) >> p.TL, blocks = tiles_per_col * vcols;
        if (xcd_pair && p.TL < 4 && p.S > p.TL && (tiles_per_col % 2) == 0 && (blocks % 16) == 0) p.col_fast = 2;
    }
    if (p.cs_bits) return zk_ntt_fast_launch_dit_cs(ctx, p, row, name);
    if (dit) return zk_ntt_fast_launch_dit(ctx, p, row, name);
    return zk_ntt_fast_launch_dif(ctx, p, row, name);
}
