// Register-radix NTT digit pass for gfx950 (the hot NTT kernel).
// Same contract as ntt.hip's LDS radix-2 pass (PassParams), different execution shape:
//   * 512 threads, 16 elements per thread held in VGPRs, tile = 8192 elements = 64 KiB of LDS -> 4 waves per SIMD;
//   * the 2^W-point digit transform (4 <= W <= 10) is done as up to three register phases of radix 16 / 16 / 2^(W-8),
//     with ONE LDS exchange between phases instead of one LDS round trip + barrier per radix-2 stage;
//   * global loads/stores are issued in the mapping whose lanes walk contiguous addresses (>= T*8 bytes per run).
// Phase bookkeeping (slot = position of an element inside the digit, W bits):
//   H phase: slot bits [W-4, W)   registers i <-> slot i*2^(W-4) + r
//   M phase: slot bits [P3, P3+P2)
//   L phase: slot bits [0, P3)
// DIF (natural digit in, bit-reversed digit out) runs H, M, L; DIT (bit-reversed in, natural out) runs L, M, H.
// Twiddle between phases: DIF  w_{2^hb}^{e * r}  (hb = slot bits not yet transformed before the phase, r = lower slot bits),
//                         DIT  w_{2^(W-lo)}^{e * bitrev(upper slot bits)}  (lo = first slot bit of the phase).
#pragma once
#include "ntt_pass.h"
#include <type_traits>

#include "ntt_radix.cuh"

template <bool DIT, bool ROW, int W, bool INV, bool CS = false>
// Two 512-thread workgroups per CU (64-78 KiB of LDS each) need <= 128 VGPRs: with the tile shape a constant hipcc hoists every
// address of the pass and asks for ~140; bounded to 128 it spills 10-28 of them to scratch and is still the fastest variant measured
// (from_values NTT at 2^20 x 93: 2.95 ms; unbounded 140 VGPRs: 3.98 ms; run-time tile shape, 84 VGPRs, +25 instructions: 3.3 ms).
// CS (coset-merged LDE): the launch covers n_cols << cs_bits VIRTUAL columns; virtual column v = (real column v >> cs_bits, coset slot
// v & mask): its data sits at column * stride + slot * sub on either side, and its input factors at load_tab + slot * tab_sub.
__global__ __launch_bounds__(512, 4) void ntt_fast_kernel(PassParams p) {
    static_assert(W >= 4 && W <= 12 && (ROW || W <= 10), "digit width");
    constexpr int P2 = (W - 4) < 4 ? (W - 4) : 4;
    constexpr int P3 = W - 4 - P2;
    constexpr int R = 1 << (W - 4);
    constexpr int G2 = 16 >> P2, G3 = 16 >> P3;
    constexpr bool SWZ = ROW && W >= 8;
    constexpr int RS = (1 << W) + (SWZ ? (16 << P3) : 0);         // LDS row stride of a ROW tile, in elements
    extern __shared__ u64 sh[];
    // the tile always holds 2^13 elements (T = 2^(13-W) rows or adjacent positions): with the tile shape a compile-time constant
    // every LDS index of an exchange is the thread's base plus an immediate offset (the launcher routes other shapes elsewhere)
    constexpr int TL = 13 - W;
    const int S = p.S;
    const u32 T = 1u << TL, tid = threadIdx.x;
    constexpr u32 nact = 512;
    constexpr bool act = true, wact = true;
    const size_t n = (size_t)1 << p.k;
    const u64 *__restrict__ RW = p.tw_full;

    size_t col = 0, base = 0, row0 = 0, rows_per_col = 1, total_rows = 0;
    if (ROW) {
        rows_per_col = n >> W; total_rows = rows_per_col * ((size_t)p.n_cols << (CS ? p.cs_bits : 0));
        const u32 vcols = (u32)p.n_cols << (CS ? p.cs_bits : 0);
        if (p.col_fast) row0 = (size_t)(blockIdx.x % vcols) * rows_per_col + ((size_t)(blockIdx.x / vcols) << TL);
        else row0 = (size_t)blockIdx.x << TL;
    } else {
        size_t tiles_per_block = ((size_t)1 << S) >> TL, tiles_per_col = (n >> (W + S)) * tiles_per_block;
        size_t rem;
        const u32 vcols = (u32)p.n_cols << (CS ? p.cs_bits : 0);
        if (p.col_fast == 2) {
            // XCD-paired order.  A TILE run is T = 2^TL adjacent elements; at W = 10 that is 8 elements = 64 bytes, HALF a 128-byte
            // line, and the workgroup holding the other half used to come n_cols workgroups later on another XCD (another L2): the
            // TILE passes fetched every line twice (FETCH_SIZE 2.14x the data against 1.09x for ROW passes and for TILE passes at
            // W = 9, profiles/r03e_pmc_tile_fetch_ab_and_whole_proof.json).  Workgroups go to the XCDs round robin (blockIdx % 8), so
            // blockIdx b and b + 8 are consecutive on one XCD: give those two the two halves of the same lines.
            const u32 b = blockIdx.x, xcd = b & 7u, m = b >> 3;
            const u32 l = (m & 1u) + 2u * (xcd + 8u * (m >> 1));
            const u32 pair = l >> 1;
            col = pair % vcols; rem = 2 * (size_t)(pair / vcols) + (l & 1u);
        } else if (p.col_fast) { col = blockIdx.x % vcols; rem = blockIdx.x / vcols; }
        else { col = blockIdx.x / tiles_per_col; rem = blockIdx.x % tiles_per_col; }
        base = ((rem / tiles_per_block) << (W + S)) + ((rem % tiles_per_block) << TL);
    }
    // the 2^W digit roots go to LDS behind the tile: the inter-phase twiddles are per-lane gathers, and an LDS gather costs a
    // fraction of an L1-hit global load's latency
    u64 *shRW = sh + (ROW ? ((size_t)RS << TL) : ((size_t)1 << (W + TL)));
    for (u32 i = tid; i < (1u << W); i += 512) shRW[i] = RW[i];

    // ROW tiles: 2^P3 padding elements after every block of R = 2^(W-4) slots (instead of an XOR swizzle): the M-phase lanes, which
    // walk slots R apart, then spread over the LDS banks, and -- unlike an XOR -- the padded index of register q is the thread's
    // base plus a compile-time constant in all three mappings, so an exchange costs no address arithmetic per element.
    auto lds_idx = [&](u32 t, u32 slot) -> u32 {
        if (ROW) return SWZ ? t * (u32)RS + slot + ((slot >> (W - 4)) << P3) : (t << W) + slot;
        return (slot << TL) + t;
    };
    // ---- thread <-> (t, slot) for register q in each mapping
    auto map_H = [&](int q, u32 &t, u32 &slot) {
        u32 r;
        if (ROW) { r = tid & (R - 1); t = tid >> (W - 4); } else { t = tid & (T - 1); r = tid >> TL; }
        slot = (u32)q * R + r;
    };
    auto map_M = [&](int q, u32 &t, u32 &slot) {
        const int g = q >> P2, i = q & ((1 << P2) - 1);
        u32 c = tid + nact * g, hi, lo;
        if (ROW) { lo = c & ((1u << P3) - 1); hi = (c >> P3) & 15; t = c >> (P3 + 4); }
        else { t = c & (T - 1); u32 rest = c >> TL; lo = rest & ((1u << P3) - 1); hi = rest >> P3; }
        slot = (hi << (P2 + P3)) | ((u32)i << P3) | lo;
    };
    auto map_L = [&](int q, u32 &t, u32 &slot) {
        const int g = q >> P3, i = q & ((1 << P3) - 1);
        u32 c = tid + nact * g, hb;
        if (ROW) { hb = c & ((1u << (W - P3)) - 1); t = c >> (W - P3); } else { t = c & (T - 1); hb = c >> TL; }
        slot = (hb << P3) | (u32)i;
    };
    // mapping ids: 0 = H, 1 = M, 2 = L
    auto map_any = [&](int m, int q, u32 &t, u32 &slot) { if (m == 0) map_H(q, t, slot); else if (m == 1) map_M(q, t, slot); else map_L(q, t, slot); };
    // global position of (t, slot); returns false when the row is past the end (ROW tail block).  Positions and row numbers fit
    // 32 bits (k <= 30 and the launcher bounds rows * columns), so only the final address is formed in 64 bits.
    const u32 base32 = (u32)base, row032 = (u32)row0, rpc_mask = (u32)rows_per_col - 1u, kW = (u32)(p.k - W), total32 = (u32)total_rows;
    auto gpos = [&](u32 t, u32 slot, size_t &c, u32 &pos) -> bool {
        if (ROW) {
            u32 rid = row032 + t;
            if (rid >= total32) return false;
            c = rid >> kW; pos = ((rid & rpc_mask) << W) + slot;      // rows_per_col = 2^(k-W)
        } else { c = col; pos = base32 + (slot << S) + t; }
        return true;
    };

    // element offset of (virtual) column c on the input / output side, and of its slice of the load table
    const u32 cs_mask = CS ? ((1u << p.cs_bits) - 1u) : 0u;
    auto in_off = [&](size_t c) -> size_t { return CS ? (c >> p.cs_bits) * p.in_stride + (c & cs_mask) * p.in_sub : c * p.in_stride; };
    auto out_off = [&](size_t c) -> size_t { return CS ? (c >> p.cs_bits) * p.out_stride + (c & cs_mask) * p.out_sub : c * p.out_stride; };
    auto tab_off = [&](size_t c) -> size_t { return CS ? (c & cs_mask) * p.tab_sub : 0; };

    u64 x[16];
    constexpr int FIRST_DIT = P3 > 0 ? 2 : (P2 > 0 ? 1 : 0);     // first phase mapping of DIT == last phase mapping of DIF
    const int load_map = DIT ? (ROW ? 0 : FIRST_DIT) : 0;

    // everything from the global loads to the last butterfly, for either arithmetic (glx<false> fast / glx<true> exact)
    auto to_lds = [&](int m) {
        if (act) {
#pragma unroll
            for (int q = 0; q < 16; q++) { u32 t, slot; map_any(m, q, t, slot); sh[lds_idx(t, slot)] = x[q]; }
        }
    };
    auto from_lds = [&](int m) {
        if (act) {
#pragma unroll
            for (int q = 0; q < 16; q++) { u32 t, slot; map_any(m, q, t, slot); x[q] = sh[lds_idx(t, slot)]; }
        }
    };

    auto compute = [&](auto &fx) {
    // the in-network twiddles as table values: only the exact arithmetic reads them (the fast arithmetic shifts), so they are
    // loaded here rather than held in 16 registers through the fast path
    u64 tw16[8];
#pragma unroll
    for (int j = 0; j < 8; j++) tw16[j] = std::remove_reference_t<decltype(fx)>::exact ? RW[j << (W - 4)] : 0;
    // ---------------------------------------------------------------- load (+ input scale, + DIT pass pre-twiddle)
    // Loads (and the factor loads) are guarded per lane and issued back to back; the products run after them in wave-uniform
    // control flow, because the sticky word of glx<false> is a scalar: inactive lanes multiply 0 by 1.
    {
        // which factor an element gets is a property of the PASS: decided once here, so that the unrolled per-element code below
        // carries no run-time branch (mode 0: none, 1: one coalesced table read, 2: the two-level tables)
        const bool need_mul = p.load_tab || p.preA || (DIT && S > 0);
        u64 f[16];
#pragma unroll
        for (int q = 0; q < 16; q++) { x[q] = 0; f[q] = 1; }
        auto load_all = [&](auto MODE, auto CANON) {
            if (!act) return;
#pragma unroll
            for (int q = 0; q < 16; q++) {
                u32 t, slot; map_any(load_map, q, t, slot);
                size_t c; u32 pos;
                if (gpos(t, slot, c, pos)) {
                    u64 v = p.in[in_off(c) + pos];
                    if (decltype(CANON)::value) v = gl_canon(v);
                    x[q] = v;
                    if (decltype(MODE)::value == 1) f[q] = p.load_tab[tab_off(c) + pos];
                    else if (decltype(MODE)::value == 2) {
                        u64 g1 = 1;
                        if (p.preA) { u32 j = p.pre_bitrev ? bitrev_u32(pos, p.k) : pos; g1 = tab2(p.preA, p.preB, j); }
                        if (DIT && S > 0) {
                            u32 inner = pos & ((1u << S) - 1);
                            u64 g2 = tab2(p.twA, p.twB, bitrev_u32(slot, W) * inner);
                            g1 = p.preA ? gl_mul(g1, g2) : g2;
                        }
                        f[q] = g1;
                    }
                }
            }
        };
        using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
        if (p.canon_in) {
            if (p.load_tab) load_all(I1{}, std::true_type{}); else if (need_mul) load_all(I2{}, std::true_type{}); else load_all(I0{}, std::true_type{});
        } else {
            if (p.load_tab) load_all(I1{}, std::false_type{}); else if (need_mul) load_all(I2{}, std::false_type{}); else load_all(I0{}, std::false_type{});
        }
        if (wact && need_mul) {
#pragma unroll
            for (int q = 0; q < 16; q++) x[q] = fx.mulw(x[q], f[q]);
        }
    }
    __syncthreads();            // shRW is complete (the data loads above were in flight meanwhile)
    if (!DIT) {
        // ------------------------------------------------------------ DIF: H, M, L
        if (wact) {
            radix_net<4, false, INV>(fx, x, 0, tw16);
            if (W > 4) {
                u32 t, slot; map_H(0, t, slot); const u32 r = slot;          // slot of register 0 == r
#pragma unroll
                for (int q = 1; q < 16; q++) x[q] = fx.mulw(x[q], shRW[act ? BR<4>::rev(q) * r : 0]);
            }
        }
        if (P2 > 0) {
            to_lds(0); __syncthreads(); from_lds(1);
            if (wact) {
#pragma unroll
                for (int g = 0; g < G2; g++) radix_net<P2, false, INV>(fx, x, g << P2, tw16);
                if (P3 > 0) {
#pragma unroll
                    for (int q = 0; q < 16; q++) {
                        const int i = q & ((1 << P2) - 1);
                        if (i == 0) continue;
                        u32 t, slot; map_M(q, t, slot);
                        u32 lo = slot & ((1u << P3) - 1);
                        x[q] = fx.mulw(x[q], shRW[act ? (BR<P2>::rev(i) * lo) << 4 : 0]);
                    }
                }
            }
        }
        if (P3 > 0) {
            to_lds(1); __syncthreads(); from_lds(2);
            if (wact) {
#pragma unroll
                for (int g = 0; g < G3; g++) radix_net<P3, false, INV>(fx, x, g << P3, tw16);
            }
        }
    } else {
        // ------------------------------------------------------------ DIT: L, M, H
        if (ROW && FIRST_DIT != 0) { to_lds(0); __syncthreads(); from_lds(FIRST_DIT); }
        if (P3 > 0) {
            if (wact) {
#pragma unroll
                for (int g = 0; g < G3; g++) radix_net<P3, true, INV>(fx, x, g << P3, tw16);
#pragma unroll
                for (int q = 0; q < 16; q++) {
                    const int i = q & ((1 << P3) - 1);
                    if (i == 0) continue;
                    u32 t, slot; map_L(q, t, slot);
                    x[q] = fx.mulw(x[q], shRW[act ? (u32)i * bitrev_u32(slot >> P3, W - P3) : 0]);
                }
            }
            to_lds(2); __syncthreads(); from_lds(1);
        }
        if (P2 > 0) {
            if (wact) {
#pragma unroll
                for (int g = 0; g < G2; g++) radix_net<P2, true, INV>(fx, x, g << P2, tw16);
#pragma unroll
                for (int q = 0; q < 16; q++) {
                    const int i = q & ((1 << P2) - 1);
                    if (i == 0) continue;
                    u32 t, slot; map_M(q, t, slot);
                    u32 hi = slot >> (P2 + P3);
                    x[q] = fx.mulw(x[q], shRW[act ? ((u32)i * bitrev_u32(hi, 4)) << P3 : 0]);
                }
            }
            to_lds(1); __syncthreads(); from_lds(0);
        }
        if (wact) radix_net<4, true, INV>(fx, x, 0, tw16);
    }
    // the networks work on weak residues: anything that leaves this function must be canonical (or flag the tile for the exact redo)
    if (wact) {
#pragma unroll
        for (int q = 0; q < 16; q++) fx.check_canon(x[q]);
    }

    };
    {
        glx<false> gf; gf.fl = 0;
        compute(gf);
        // a skipped rare repair anywhere in the workgroup (waves exchange through LDS): redo the tile with every repair in place
#ifndef ZK_ISA_COUNT_FAST_ONLY      // analysis builds (tools/isa_mix.py) drop the exact twin so that the static mix is the fast path's
        if (__syncthreads_or((gf.fl != 0) | p.force_redo)) { glx<true> ge; ge.fl = 0; compute(ge); }
#endif
    }

    // ---------------------------------------------------------------- store
    constexpr int LAST_DIF = FIRST_DIT;
    int smap = DIT ? 0 : LAST_DIF;
    if (DIT && p.permute_out) {
        // scatter whole rows to bit-reversed positions: go through LDS so that lanes walk consecutive output positions
        // Exchange layout for the TILE case: element (t, slot) sits at t*2^W + swz(bitrev(slot)), so the read side
        // (consecutive output positions pe) is linear, and the XOR with the top m = 6-TL address bits and with t spreads the
        // write side (lanes = t x low bits of r, which land in the TOP bits of bitrev(slot)) over all 64 banks.
        const int m_sw = 6 - TL;
        const bool pswz = !ROW && TL <= 6 && (W - m_sw) >= 6;
        auto pidx = [&](u32 t, u32 a) -> u32 {
            u32 a2 = a ^ ((a >> (W - m_sw)) & ((1u << m_sw) - 1)) ^ ((t << m_sw) & 63u);
            return (t << W) + a2;
        };
        __syncthreads();
        if (pswz) {
            if (act) {
#pragma unroll
                for (int q = 0; q < 16; q++) { u32 t, slot; map_H(q, t, slot); sh[pidx(t, bitrev_u32(slot, W))] = x[q]; }
            }
        } else to_lds(0);
        __syncthreads();
        auto scatter = [&](auto POST) {
            if (!act) return;
#pragma unroll
            for (int q = 0; q < 16; q++) {
                u32 lin = tid + nact * q;
                u32 t = lin >> W, pe = lin & ((1u << W) - 1), e = bitrev_u32(pe, W);
                u64 v = pswz ? sh[pidx(t, pe)] : sh[lds_idx(t, e)];
                size_t c; u32 pos, K;
                if (ROW) {
                    u32 rid = row032 + t;
                    if (rid >= total32) continue;
                    c = rid >> kW; pos = ((rid & rpc_mask) << W) + pe; K = ((rid & rpc_mask) << W) + e;
                } else {
                    c = col;
                    u32 inner = (base32 & ((1u << S) - 1)) + t;
                    pos = (bitrev_u32(inner, S) << W) + pe; K = (e << S) + inner;
                }
                if (decltype(POST)::value) {
                    if (p.postA) v = gl_mul(v, tab2(p.postA, p.postB, K));      // exact products: the permuting pass rarely has an output scale
                    if (p.post_const) v = gl_mul(v, p.post_const);
                }
                p.out[out_off(c) + pos] = v;
            }
        };
        if (p.postA || p.post_const) scatter(std::true_type{}); else scatter(std::false_type{});
        return;
    }
    if (!DIT && ROW && LAST_DIF != 0) { __syncthreads(); to_lds(LAST_DIF); __syncthreads(); from_lds(0); smap = 0; }
    // same structure as the load side: guarded batched factor loads, products in wave-uniform control flow, guarded stores;
    // x[] is left intact, so a flagged wave runs the loop again with exact products and overwrites what it stored
    auto store_plain = [&](auto &fx) {
        const bool need_mul = p.store_tab || (!DIT && S > 0) || p.postA || p.post_const;
        u64 f[16];
#pragma unroll
        for (int q = 0; q < 16; q++) f[q] = 1;
        // as on the load side: the factor source is decided once per workgroup (mode 1: table, 2: two-level tables)
        auto factors = [&](auto MODE) {
            if (!act) return;
#pragma unroll
            for (int q = 0; q < 16; q++) {
                u32 t, slot; map_any(smap, q, t, slot);
                size_t c; u32 pos;
                if (!gpos(t, slot, c, pos)) continue;
                if (decltype(MODE)::value == 1) f[q] = p.store_tab[pos];
                else {
                    const u32 e = DIT ? slot : bitrev_u32(slot, W);
                    u64 g1 = 1; bool have = false;
                    if (!DIT && S > 0) { u32 inner = pos & ((1u << S) - 1); g1 = tab2(p.twA, p.twB, e * inner); have = true; }
                    if (p.postA) {
                        u32 K = ROW ? (((pos >> W) << W) + e) : (u32)(((base >> (W + S)) << (W + S)) + ((size_t)e << S) + (pos & ((1u << S) - 1)));
                        u64 g2 = tab2(p.postA, p.postB, K);
                        g1 = have ? gl_mul(g1, g2) : g2; have = true;
                    }
                    if (p.post_const) g1 = have ? gl_mul(g1, p.post_const) : p.post_const;
                    f[q] = g1;
                }
            }
        };
        if (need_mul) { if (p.store_tab) factors(std::integral_constant<int, 1>{}); else factors(std::integral_constant<int, 2>{}); }
        if (wact && need_mul) {
#pragma unroll
            for (int q = 0; q < 16; q++) f[q] = fx.mul(x[q], f[q]);            // f now holds the values to store
        }
        if (act) {
#pragma unroll
            for (int q = 0; q < 16; q++) {
                u32 t, slot; map_any(smap, q, t, slot);
                size_t c; u32 pos;
                if (!gpos(t, slot, c, pos)) continue;
                p.out[out_off(c) + pos] = need_mul ? f[q] : x[q];
            }
        }
    };
    {
        glx<false> gf; gf.fl = 0;
        store_plain(gf);
#ifndef ZK_ISA_COUNT_FAST_ONLY
        if ((gf.fl != 0) | p.force_redo) { glx<true> ge; ge.fl = 0; store_plain(ge); }
#endif
    }
}


// ---- launchers (one translation unit per family: ntt_fast.hip DIF, ntt_fast_dit.hip DIT, ntt_fast_cs.hip coset-merged DIT -- the
// families compile side by side under `make -j`)
template <bool DIT, bool ROW, int W, bool INV, bool CS>
static void ntt_fast_launch_wi(zk_ctx *ctx, const PassParams &p, const char *name) {
    size_t n = (size_t)1 << p.k, blocks;
    const size_t vcols = (size_t)p.n_cols << (CS ? p.cs_bits : 0);
    if (ROW) { size_t rows = (n >> p.W) * vcols; blocks = (rows + ((size_t)1 << p.TL) - 1) >> p.TL; }
    else blocks = ((n >> p.W) >> p.TL) * vcols;
    constexpr int P2 = (W - 4) < 4 ? (W - 4) : 4, P3 = W - 4 - P2;
    const size_t row_stride = ((size_t)1 << W) + ((ROW && W >= 8) ? ((size_t)16 << P3) : 0);      // == RS in the kernel
    size_t shmem = ((ROW ? (row_stride << p.TL) : ((size_t)1 << (p.W + p.TL))) + ((size_t)1 << p.W)) * sizeof(u64);
    ZK_LAUNCH(ctx, name, (ntt_fast_kernel<DIT, ROW, W, INV, CS>), dim3((unsigned)blocks), dim3(512), shmem, p);
}
template <bool DIT, bool ROW, bool INV, bool CS>
static bool ntt_fast_launch_w(zk_ctx *ctx, const PassParams &p, const char *name) {
    switch (p.W) {
    case 4: ntt_fast_launch_wi<DIT, ROW, 4, INV, CS>(ctx, p, name); return true;
    case 5: ntt_fast_launch_wi<DIT, ROW, 5, INV, CS>(ctx, p, name); return true;
    case 6: ntt_fast_launch_wi<DIT, ROW, 6, INV, CS>(ctx, p, name); return true;
    case 7: ntt_fast_launch_wi<DIT, ROW, 7, INV, CS>(ctx, p, name); return true;
    case 8: ntt_fast_launch_wi<DIT, ROW, 8, INV, CS>(ctx, p, name); return true;
    case 9: ntt_fast_launch_wi<DIT, ROW, 9, INV, CS>(ctx, p, name); return true;
    case 10: ntt_fast_launch_wi<DIT, ROW, 10, INV, CS>(ctx, p, name); return true;
    // whole rows of 2^11 / 2^12 elements (three radix-16 phases): single-pass transforms of small polynomials; ROW tiles only
    case 11: if constexpr (ROW) { ntt_fast_launch_wi<DIT, ROW, 11, INV, CS>(ctx, p, name); return true; } return false;
    case 12: if constexpr (ROW) { ntt_fast_launch_wi<DIT, ROW, 12, INV, CS>(ctx, p, name); return true; } return false;
    default: return false;
    }
}
template <bool DIT, bool CS>
static bool ntt_fast_launch_family(zk_ctx *ctx, const PassParams &p, bool row, const char *name) {
    if (p.inverse) {
        if constexpr (CS) return false;       // the coset-merged form is the forward LDE only
        else return row ? ntt_fast_launch_w<DIT, true, true, false>(ctx, p, name) : ntt_fast_launch_w<DIT, false, true, false>(ctx, p, name);
    }
    return row ? ntt_fast_launch_w<DIT, true, false, CS>(ctx, p, name) : ntt_fast_launch_w<DIT, false, false, CS>(ctx, p, name);
}
bool zk_ntt_fast_launch_dif(zk_ctx *ctx, const PassParams &p, bool row, const char *name);
bool zk_ntt_fast_launch_dit(zk_ctx *ctx, const PassParams &p, bool row, const char *name);
bool zk_ntt_fast_launch_dit_cs(zk_ctx *ctx, const PassParams &p, bool row, const char *name);
