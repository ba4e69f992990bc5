// Parameters of one NTT digit pass, shared by the LDS radix-2 kernel (ntt.hip) and the register-radix kernel (ntt_fast.hip).
#pragma once
#include "zk_internal.h"

struct PassParams {
    const u64 *in; u64 *out;
    size_t in_stride, out_stride;   // column strides in elements
    int k, W, S, TL;                // transform bits, digit bits, stride bits, log2 tile
    int n_cols;
    const u64 *tw_digit;            // w_{2^W}^{+-i}, i < 2^(W-1)
    const u64 *tw_full;             // w_{2^W}^{+-i}, i < 2^W  (register-radix kernel)
    const u64 *twA, *twB;           // two-level table of w_{2^(W+S)}^{+-x}
    const u64 *preA, *preB;         // optional input scale table (indexed by coefficient index)
    const u64 *postA, *postB;       // optional output scale table (indexed by natural output index)
    u64 post_const;                 // multiply every output (0 = none)
    int pre_bitrev;                 // pre-scale index = bitrev_k(pos) instead of pos
    int permute_out;                // DIT last pass: scatter rows to bit-reversed positions
    int canon_in;                   // canonicalise loaded values
    int inverse;                    // direction (selects the twiddle tables)
    // register-radix kernel only: every per-position factor of the pass folded into one coalesced table each
    const u64 *load_tab;            // multiply the element loaded from position pos by load_tab[pos]  (or NULL)
    const u64 *store_tab;           // multiply the element stored to position pos by store_tab[pos]  (or NULL)
    int force_redo;                 // test hook: take the exact-redo branch in every workgroup (ctx->force_redo)
    int col_fast;                   // blockIdx -> (tile, column) with the column fastest, so table slices stay in L2
    int col_group;                  // TILE passes: > 0 = groups of this many columns, tile position fastest inside a group (kernel comment)
    // coset-merged LDE (DIT, register-radix kernel only): cs_bits > 0 makes the launch cover n_cols << cs_bits virtual columns;
    // virtual column v = (column v >> cs_bits, slot v & mask) sits at column * stride + slot * sub; its factors at load_tab + slot * tab_sub
    int cs_bits;
    size_t in_sub, out_sub, tab_sub;
    // tall transforms (k > 24): a table whose factors repeat with period 2^(W+S) (inter-pass factors of a middle pass) holds one period,
    // and the kernel reduces the tile's base position by this mask before it indexes the table (all ones: the table covers the column)
    unsigned load_tab_mask, store_tab_mask;
    u64 store_const;                // multiply every stored element by this constant instead of a table (0: none)
    unsigned nt_store;              // whole-line data stores with the non-temporal hint (set by the launcher for passes over more data than the caches hold)
    unsigned n_blocks;              // register-radix kernel, persistent launch: number of tiles (0: one workgroup per tile, the grid says it)
};

__device__ __forceinline__ u64 tab2(const u64 *A, const u64 *B, u32 x) { return gl_mul(A[x & 2047], B[x >> 11]); }


// register-radix pass (ntt_fast.hip); returns false when the shape is not covered (caller falls back to the LDS radix-2 kernel)
// cs_pre (coset-merged first pass only): the input scale table of every coset slot, folded into one load table of 2^cs_bits slices
struct NttCosetPre { const u64 *A[16], *B[16]; };
bool zk_ntt_fast_pass(zk_ctx *ctx, const PassParams &p, bool dit, bool row, const char *name, const NttCosetPre *cs_pre = nullptr);
u64 *zk_full_digit_table(zk_ctx *ctx, int W, int inverse);
