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
#include <stdlib.h>

#include "ntt_radix.cuh"

// Global addressing (round 4).  For every register mapping the pair (t, slot) of register q is the pair of register 0 -- a function of
// the thread id only -- PLUS a pair that depends on q alone (NttGeom, compile-time): the thread bits and the register bits occupy
// disjoint bit fields in all three mappings.  The element offset of an access is therefore
//     [wave-uniform: column base + tile base + the register's part]  +  [per thread: ONE 32-bit byte offset per mapping and side],
// which is the addressing mode of the hardware (global_load  vdst, voffset32, sbase64): the uniform part lives in scalar registers /
// immediates and costs no vector instruction, the thread part is one VGPR instead of a hoisted 64-bit address per register.  Before,
// hipcc formed 16 + 16 + 16 full addresses (v_lshl_add_u64 each) per thread and, bounded to 128 VGPRs, spilled up to 64 of them to
// scratch in every one of the 47 instantiations; now no instantiation uses scratch (tools/kernel_resources.py, tests/test_abi_cpu.py).
// Two shapes of tile: (A) all rows of the tile in one column (every TILE pass; ROW passes with at least T rows per column) and
// (B) one row per column (ROW passes of whole transforms, k = W <= 12: the tile holds T columns).  The launcher routes anything else
// to the LDS radix-2 kernel.
#ifndef ZK_NTT_PERSIST_DEFAULT
#define ZK_NTT_PERSIST_DEFAULT 0
#endif
#ifndef NTT_SH64            // 0 (tools/ntt_floor.hip only): the round-4 form, every inter-phase twiddle a general product
#define NTT_SH64 1
#endif
template <bool ROW, int W> struct NttGeom {
    static constexpr int P2 = (W - 4) < 4 ? (W - 4) : 4, P3 = W - 4 - P2, TL = 13 - W, R = 1 << (W - 4);
    // W = 9, 10: the M and L phases together are a 2^(4+P3) <= 64-point transform, whose root w_32 = 2^6 / w_64 = 2^3 is a power of two:
    // the twiddle between the two phases is a SHIFT, by an amount that depends on the register (compile time) and on the P3 low slot
    // bits -- which the M mapping of these widths takes from the WAVE number, so that the amount is uniform per wave (round 5)
    // (W = 9 TILE tiles spilled 6-9 registers with the shift form until the exchange indices were spelled out as thread base + constant,
    // lds_off below: that freed ~20 registers in every instantiation)
    static constexpr bool SH64 = NTT_SH64 && (P2 == 4) && (P3 == 2 || P3 == 1);
    // LDS image of the tile.  ROW: row stride RS; SH64 tiles pad one element per 32 slots (+1 per row at W = 10), the others 2^P3 per
    // 2^(W-4) slots.  TILE: element (t, slot) at slot * T + t; SH64 tiles skip T elements after every 2^(W-4) slots.  With these every
    // exchange of the SH64 mappings is free of bank conflicts on ds_read_b64 (32-lane groups, 64 banks) except the TILE L reads.
    static constexpr int RS = !ROW ? 0 : SH64 ? (1 << W) + (1 << (W - 5)) + (W == 10 ? 1 : 0) : (1 << W) + (W >= 8 ? (16 << P3) : 0);
    static constexpr size_t tile_elems = ROW ? ((size_t)RS << TL) : ((size_t)((1 << W) + (SH64 ? 16 : 0)) << TL);
    // index of element (t, slot) in the LDS image.  ADDITIVE over disjoint bit fields of (t, slot): the index of register q's element is
    // the index of the thread's part plus lds_idx(t_uni, slot_uni) of the register's part -- a compile-time constant (lds_off below)
    static constexpr u32 lds_idx(u32 t, u32 slot) {
        if (SH64) return ROW ? t * (u32)RS + slot + (slot >> 5) : ((slot + (slot >> (W - 4))) << TL) + t;
        if (ROW) return W >= 8 ? t * (u32)RS + slot + ((slot >> (W - 4)) << P3) : (t << W) + slot;
        return (slot << TL) + t;
    }
    static constexpr u32 lds_off(int m, int q) { return lds_idx(t_uni(m, q), slot_uni(m, q)); }
    // the part of (t, slot) of mapping m (0 = H, 1 = M, 2 = L) that register q contributes; map_X(q) = map_X(0) + these
    static constexpr u32 t_uni(int m, int q) {
        if (!ROW || m == 0) return 0;
        return m == 1 ? (u32)(q >> P2) << (5 - P3) : (u32)(q >> P3) << (5 - P2);
    }
    static constexpr u32 slot_uni(int m, int q) {
        if (m == 0) return (u32)q * R;
        if (m == 1) { const u32 g = q >> P2, i = q & ((1 << P2) - 1); return ROW ? (i << P3) : (((g << P2) << (P2 + P3)) + (i << P3)); }
        const u32 g = q >> P3, i = q & ((1 << P3) - 1);
        return ROW ? i : ((g << (W - 4 + P3)) + i);
    }
};
// 2^e mod p for the exact twin's in-network twiddles: 2^64 = 2^32 - 1, 2^96 = -1, 2^192 = 1
__host__ __device__ constexpr u64 ntt_pow2(int e) {
    e %= 192;
    if (e >= 96) { const u64 r = ntt_pow2(e - 96); return 0xFFFFFFFF00000001ull - r; }
    return e < 64 ? (1ull << e) : (((1ull << 32) - 1) << (e - 64));
}
static_assert(ntt_pow2(0) == 1 && ntt_pow2(72) == (1ull << 40) - 256 && ntt_pow2(96) == 0xFFFFFFFF00000000ull && ntt_pow2(191) == 0xFFFFFFFF00000001ull - (0xFFFFFFFFull << 31), "2^e mod p");
// x[sbase + thread byte offset]: sbase wave-uniform (scalar registers), the offset one 32-bit VGPR
// (readfirstlane makes the base one opaque scalar value: left visible, hipcc re-associates  base + (uniform + thread)  into a 64-bit
// vector sum and is back to one v_lshl_add_u64 and two VGPRs per access; an empty asm with an "s" operand does the same for loads OR
// stores but, used for both, ends in "illegal VGPR to SGPR copy".  The access is typed as GLOBAL memory: behind the intrinsic the
// pointer's address space is no longer inferred and the access would become flat_load / flat_store.)
__device__ __forceinline__ u64 ntt_uniform(u64 v) {
    return ((u64)(u32)__builtin_amdgcn_readfirstlane((int)(u32)(v >> 32)) << 32) | (u32)__builtin_amdgcn_readfirstlane((int)(u32)v);
}
// NTT_FLOOR (tools/ntt_floor.hip only; never defined for the library): 1 = the pass without its global data traffic -- a load is a value
// made from its address, a store happens only for a value that never occurs -- 2 = without the LDS exchanges between phases as well:
// what is left is the instruction stream of the butterflies, twiddles and factor products, whose run time is the MEASURED issue-bound floor.
#ifndef NTT_FLOOR
#define NTT_FLOOR 0
#endif
// NTT_NT (experiment): 1 = the pass's data stores carry the non-temporal hint (written once, read by a later kernel: keep them out of the way
// of the half-used input lines that the paired workgroup of a TILE pass still needs from L2), 2 = its data loads as well
#ifndef NTT_NT
#define NTT_NT 0
#endif
__device__ __forceinline__ u64 ntt_ld(const u64 *sbase, u32 thr_bytes) {
    const u64 b = ntt_uniform((u64)sbase);
#if NTT_FLOOR && NTT_FLOOR != 3      // (3: real loads, no stores; 4: made-up loads, real stores -- which side of the traffic is exposed?)
    return ((b + thr_bytes) * 0x9E3779B97F4A7C15ull) >> 1;     // a well-mixed value below 2^63 (structured values meet the rare-repair paths far too often)
#else
#if NTT_NT >= 2
    return __builtin_nontemporal_load((const __attribute__((address_space(1))) u64 *)(b + thr_bytes));
#else
    return *(const __attribute__((address_space(1))) u64 *)(b + thr_bytes);
#endif
#endif
}
__device__ __forceinline__ void ntt_st(u64 *sbase, u32 thr_bytes, u64 v, bool nt = false) {
    const u64 b = ntt_uniform((u64)sbase);
#if NTT_FLOOR && NTT_FLOOR != 4
    if (v != 0x5A5A5A5A5A5A5A5Aull) return;
#endif
    // nt (wave-uniform): whole-line stores of a pass over far more data than the caches hold go out with the non-temporal hint -- they are
    // read again only by a later kernel, and they no longer push the lines other workgroups still read out of L2 (round 5: -1.5...-2.5 % on
    // the ROW passes and on the permuted row store at 93 x 2^20; NOT for the 64-byte runs of a plain TILE store, which must meet their
    // other half in L2: +3 % there; profiles/r05r_ntt_nontemporal.txt)
    if (nt || NTT_NT >= 1) __builtin_nontemporal_store(v, (__attribute__((address_space(1))) u64 *)(b + thr_bytes));
    else *(__attribute__((address_space(1))) u64 *)(b + thr_bytes) = v;
}

template <bool DIT, bool ROW, int W, bool INV, bool CS = false>
// Two 512-thread workgroups per CU (64-78 KiB of LDS each) need <= 128 VGPRs.
// CS (coset-merged LDE): the launch covers n_cols << cs_bits VIRTUAL columns; virtual column v = (real column v >> cs_bits, coset slot
// v & mask): its data sits at column * stride + slot * sub on either side, and its input factors at load_tab + slot * tab_sub.
__global__ __launch_bounds__(512, 4) void ntt_fast_kernel(PassParams p_arg) {
    static_assert(W >= 4 && W <= 12 && (ROW || W <= 10), "digit width");
    constexpr int P2 = (W - 4) < 4 ? (W - 4) : 4;
    constexpr int P3 = W - 4 - P2;
    constexpr int R = 1 << (W - 4);
    constexpr int G2 = 16 >> P2, G3 = 16 >> P3;
    constexpr bool SWZ = ROW && W >= 8;
    constexpr bool SH64 = NttGeom<ROW, W>::SH64;
    constexpr int RS = ROW ? NttGeom<ROW, W>::RS : (1 << W);      // LDS row stride of a ROW tile, in elements
    extern __shared__ u64 sh[];
    __shared__ u32 redo_vote;
    // the tile always holds 2^13 elements (T = 2^(13-W) rows or adjacent positions): with the tile shape a compile-time constant
    // every LDS index of an exchange is the thread's base plus an immediate offset (the launcher routes other shapes elsewhere)
    constexpr int TL = 13 - W;
    const u32 T = 1u << TL, tid0 = threadIdx.x;
    constexpr u32 nact = 512;
    constexpr bool act = true, wact = true;
    const u64 *__restrict__ RW = p_arg.tw_full;

    // the 2^W digit roots go to LDS behind the tile: the inter-phase twiddles are per-lane gathers, and an LDS gather costs a
    // fraction of an L1-hit global load's latency
    u64 *shRW = sh + NttGeom<ROW, W>::tile_elems;
    for (u32 i = tid0; i < (1u << W); i += 512) shRW[i] = RW[i];
    // (Requesting the roots BEHIND the first tile's data loads -- so that a workgroup does not wait out an L2 round trip before it asks for
    // its data -- was measured in round 5: 2-4 % SLOWER on all four passes of tools/ntt_floor.hip; the fill stays here.)
    // Persistent launch (round 5, p.n_blocks > 0): the grid is a fixed number of workgroups and workgroup b takes the tiles b, b + grid,
    // b + 2 grid, ... -- no wait for a tile's stores to drain and for the next workgroup to be set up between two tiles of a CU slot, the
    // roots are staged once.  grid % 8 == 0 keeps a workgroup's tiles on "its" XCD numbering (the pairing of col_fast == 2 holds).
    const u32 nblk = p_arg.n_blocks ? p_arg.n_blocks : gridDim.x;
    const u32 wave_id0 = (u32)__builtin_amdgcn_readfirstlane((int)(tid0 >> 6));
    for (u32 bid = blockIdx.x; bid < nblk; bid += gridDim.x) {
    // every wave is done with the LDS image of the previous tile.  An LDS-only barrier: __syncthreads() also waits for the previous
    // tile's global STORES to drain (its fence covers global memory), and this tile's loads would queue up behind that wait
    if (bid != blockIdx.x) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    }
    // the thread id is made opaque once per tile: everything below that depends on the thread alone (exchange indices, byte offsets of
    // the three mappings, row limits) would otherwise be hoisted out of the tile loop and kept alive across it -- 30 more registers
    // than the 128 that two workgroups per CU allow, i.e. scratch
    // (rebuilt from the wave number -- a scalar -- and the lane number, so that not even the id itself occupies a vector register
    // across the loop)
    u32 tid = (wave_id0 << 6) + __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    asm volatile("" : "+v"(tid));
    if (tid == 0) redo_vote = 0;                                      // set (after the barrier inside compute) by any wave that skipped a rare repair
    // ... and the pass parameters are read from the kernel-argument segment again in every tile (scalar loads that hit the scalar cache)
    // instead of staying in ~50 scalar registers across the loop, which spilled scalars into vector lanes
    const __attribute__((address_space(4))) PassParams *kp = (const __attribute__((address_space(4))) PassParams *)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kp));
    const __attribute__((address_space(4))) PassParams &p = *kp;
    const int S = p.S;
    const size_t n = (size_t)1 << p.k;
    size_t rows_per_col = 1, total_rows = 0;
    if (ROW) { rows_per_col = n >> W; total_rows = rows_per_col * ((size_t)p.n_cols << (CS ? p.cs_bits : 0)); }
    // launch index -> the tile's place: (column, base position) of a TILE tile, first row of a ROW tile
    auto tile_geom = [&](const u32 bid, size_t &col, size_t &base, size_t &row0) {
    col = 0; base = 0; row0 = 0;
    if (ROW) {
        const u32 vcols = (u32)p.n_cols << (CS ? p.cs_bits : 0);
        if (p.col_fast) row0 = (size_t)(bid % vcols) * rows_per_col + ((size_t)(bid / vcols) << TL);
        else row0 = (size_t)bid << TL;
    } else {
        size_t tiles_per_block = ((size_t)1 << S) >> TL, tiles_per_col = (n >> (W + S)) * tiles_per_block;
        size_t rem;
        const u32 vcols = (u32)p.n_cols << (CS ? p.cs_bits : 0);
        // blockIdx -> (column, tile of the column).  Column fastest (col_fast) keeps a tile position's slice of the factor table in L2
        // while every column passes through it, but then the workgroups in flight at one time (two per CU) touch only ~5 adjacent runs
        // of each 2^S-element stripe of a column: short bursts scattered over DRAM pages.  p.col_group = G > 0 orders the launch by groups
        // of G columns, tile position fastest inside a group: 512 / G adjacent tiles of a column are in flight together, so whole
        // stripes stream out of open DRAM pages, and the table slice is still shared by G columns (tools/ntt_floor.hip measures G).
        auto split = [&](u32 unit, u32 units_per_col, u32 &c, u32 &u) {
            if (p.col_group > 0) {
                const u32 G = (u32)p.col_group, gsz = G * units_per_col, g = unit / gsz, idx = unit % gsz, c0 = g * G;
                const u32 Gl = vcols - c0 < G ? vcols - c0 : G;
                c = c0 + idx % Gl; u = idx / Gl;
            } else { c = unit % vcols; u = unit / vcols; }
        };
        if (p.col_fast == 2) {
            // XCD-paired order.  A TILE run is T = 2^TL adjacent elements; at W = 10 that is 8 elements = 64 bytes, HALF a 128-byte
            // line, and the workgroup holding the other half used to come n_cols workgroups later on another XCD (another L2): the
            // TILE passes fetched every line twice (FETCH_SIZE 2.14x the data against 1.09x for ROW passes and for TILE passes at
            // W = 9, profiles/r03e_pmc_tile_fetch_ab_and_whole_proof.json).  Workgroups go to the XCDs round robin (blockIdx % 8), so
            // blockIdx b and b + 8 are consecutive on one XCD: give those two the two halves of the same lines.
            const u32 b = bid, xcd = b & 7u, m = b >> 3;
            const u32 l = (m & 1u) + 2u * (xcd + 8u * (m >> 1));
            u32 c, u; split(l >> 1, (u32)(tiles_per_col >> 1), c, u);
            col = c; rem = 2 * (size_t)u + (l & 1u);
        } else if (p.col_fast) { u32 c, u; split(bid, (u32)tiles_per_col, c, u); col = c; rem = u; }
        else { col = bid / tiles_per_col; rem = bid % tiles_per_col; }
        base = ((rem / tiles_per_block) << (W + S)) + ((rem % tiles_per_block) << TL);
    }
    };
    size_t col, base, row0;
    tile_geom(bid, col, base, row0);
    // ROW tiles: 2^P3 padding elements after every block of R = 2^(W-4) slots (instead of an XOR swizzle): the M-phase lanes, which
    // walk slots R apart, then spread over the LDS banks, and -- unlike an XOR -- the padded index of register q is the thread's
    // base plus a compile-time constant in all three mappings, so an exchange costs no address arithmetic per element.
    auto lds_idx = [&](u32 t, u32 slot) -> u32 { return NttGeom<ROW, W>::lds_idx(t, slot); };
    // SH64: the low slot bits of the M mapping, uniform per wave (a scalar: the shift twiddles branch on it)
    u32 wave_id = wave_id0;
    asm volatile("" : "+s"(wave_id));                                 // opaque per tile, like tid: nothing derived from it is kept across the loop
    const u32 lo_uni = wave_id & ((1u << P3) - 1);
    // ---- thread <-> (t, slot) for register q in each mapping
    auto map_H = [&](int q, u32 &t, u32 &slot) {
        u32 r;
        if (ROW) { r = tid & (R - 1); t = tid >> (W - 4); } else { t = tid & (T - 1); r = tid >> TL; }
        slot = (u32)q * R + r;
    };
    auto map_M = [&](int q, u32 &t, u32 &slot) {
        const int g = q >> P2, i = q & ((1 << P2) - 1);
        u32 c = tid + nact * g, hi, lo;
        if (SH64) {
            // lo from the wave number; the lanes run over hi and the low bits of t in the order that keeps each 32-lane group on 32
            // different banks of the padded image (tools/ntt_lds_banks.py): g == 0 here (P2 == 4)
            const u32 l = tid & 63u, w = wave_id;
            lo = lo_uni;
            if (ROW) {
                if (W == 10) { hi = (l >> 1) & 15u; t = ((w >> 2) << 2) | ((l >> 5) << 1) | (l & 1u); }
                else { hi = l & 15u; t = ((w >> 1) << 2) | (l >> 4); }
            } else {
                if (W == 10) { t = l & 7u; hi = (l >> 3) | ((w >> 2) << 3); }
                else { t = l & 15u; hi = (l >> 4) | ((w >> 1) << 2); }
            }
            slot = (hi << (P2 + P3)) | ((u32)i << P3) | lo;
            return;
        }
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
    // ---- global addressing (see the top of the file): per side a uniform element base and one per-thread byte offset per mapping
    using GM = NttGeom<ROW, W>;
    const u32 base32 = (u32)base, row032 = (u32)row0, rpc_mask = (u32)rows_per_col - 1u, total32 = (u32)total_rows;
    const u32 cs_bits = CS ? (u32)p.cs_bits : 0u, cs_mask = CS ? ((1u << cs_bits) - 1u) : 0u;
    const bool modeB = ROW && rows_per_col == 1;                       // one row per column: the tile's rows are T different columns
    // offset of (virtual) column v on a side with column stride `stride` and coset-slot stride `sub`
    auto col_off = [&](size_t v, size_t stride, size_t sub) -> size_t { return CS ? (v >> cs_bits) * stride + (v & cs_mask) * sub : v * stride; };
    const size_t vcolA = ROW ? (size_t)(row032 >> (u32)(p.k - W)) : col;                  // shape A: the tile's column
    const u32 rr0 = row032 & rpc_mask;                                                   // shape A, ROW: first row of the tile inside its column
    // uniform element offset of register q's access in mapping m: side = 0 input, 1 output, 2 load table, 3 store table
    auto uni_pos = [&](int m, int q) -> u32 {                                            // position inside the column (shape A)
        return ROW ? ((rr0 + GM::t_uni(m, q)) << W) + GM::slot_uni(m, q) : base32 + (GM::slot_uni(m, q) << S);
    };
    auto uni_off = [&](int side, int m, int q) -> size_t {
        const size_t stride = side == 0 ? p.in_stride : side == 1 ? p.out_stride : 0, sub = side == 0 ? p.in_sub : side == 1 ? p.out_sub : (side == 2 ? p.tab_sub : 0);
        if (modeB) return col_off((size_t)row032 + GM::t_uni(m, q), stride, sub) + GM::slot_uni(m, q);
        if (!ROW && side >= 2) {
            // a factor table may hold ONE period of 2^(W+S) positions (tall transforms): the tile's base position is reduced, the rest
            // of the position (slot << S, t) lies below the period anyway
            const u32 mask = side == 2 ? p.load_tab_mask : p.store_tab_mask;
            return (side == 3 ? 0 : col_off(vcolA, stride, sub)) + (size_t)((base32 & mask) + (GM::slot_uni(m, q) << S));
        }
        return (side == 3 ? 0 : col_off(vcolA, stride, sub)) + uni_pos(m, q);
    };
    // per-thread byte offset of mapping m on a side (the launcher has checked that it fits 32 bits)
    auto thr_pos = [&](int m) -> u32 {                                                   // position part (shape A), in elements
        u32 t, slot; map_any(m, 0, t, slot);
        return ROW ? (t << W) + slot : (slot << S) + t;
    };
    auto thr_off = [&](int side, int m) -> u32 {
        u32 t, slot; map_any(m, 0, t, slot);
        if (modeB) {
            const size_t stride = side == 0 ? p.in_stride : side == 1 ? p.out_stride : 0, sub = side == 0 ? p.in_sub : side == 1 ? p.out_sub : (side == 2 ? p.tab_sub : 0);
            return (u32)(col_off(t, stride, sub) + slot) << 3;
        }
        return thr_pos(m) << 3;
    };
    // shape B only: rows past the last column are skipped (t of register q = thread part + GM::t_uni); shape A tiles are always whole
    auto row_lim = [&](int m) -> int { u32 t, slot; map_any(m, 0, t, slot); return modeB ? (int)(total32 - row032) - (int)t : 0x7fffffff; };

    u64 x[16];
    constexpr int FIRST_DIT = P3 > 0 ? 2 : (P2 > 0 ? 1 : 0);     // first phase mapping of DIT == last phase mapping of DIF
    constexpr int load_map = DIT ? (ROW ? 0 : FIRST_DIT) : 0;
    const u32 ld_thr = thr_off(0, load_map), ldtab_thr = thr_off(2, load_map);
    const int ld_lim = row_lim(load_map);

    // everything from the global loads to the last butterfly, for either arithmetic (glx<false> fast / glx<true> exact)
    auto to_lds = [&](int m) {
        if (act && NTT_FLOOR != 2) {
            // thread part once + a constant per register (round 5: spelled out -- with the one-pad-per-32-slots image hipcc no longer found
            // the split by itself and spent four vector instructions per element and exchange on the index)
            u32 t0, s0; map_any(m, 0, t0, s0);
            const u32 b = lds_idx(t0, s0);
#pragma unroll
            for (int q = 0; q < 16; q++) sh[b + NttGeom<ROW, W>::lds_off(m, q)] = x[q];
        }
    };
    auto from_lds = [&](int m) {
        if (act && NTT_FLOOR != 2) {
            u32 t0, s0; map_any(m, 0, t0, s0);
            const u32 b = lds_idx(t0, s0);
#pragma unroll
            for (int q = 0; q < 16; q++) x[q] = sh[b + NttGeom<ROW, W>::lds_off(m, q)];
        }
    };

    auto compute = [&](auto &fx) {
    // the in-network twiddles as values: only the exact arithmetic multiplies by them (the fast arithmetic shifts).  They are powers of
    // w_16 = 2^12, i.e. compile-time constants (ntt_pow2) -- literals at their uses instead of eight table loads and 16 live registers
    u64 tw16[8];
#pragma unroll
    for (int j = 0; j < 8; j++) tw16[j] = std::remove_reference_t<decltype(fx)>::exact ? ntt_pow2(INV ? (j ? 96 - 12 * j : 0) : 12 * j) : 0;
    // ---------------------------------------------------------------- load (+ input scale, + DIT pass pre-twiddle)
    // Loads (and the factor loads) are guarded per lane and issued back to back; the products run after them in wave-uniform
    // control flow, because the sticky word of glx<false> is a scalar: inactive lanes multiply 0 by 1.
    {
        // every per-position factor of a pass (input scale, coset shift, inter-pass twiddle) comes folded into ONE coalesced table
        // (p.load_tab, built once per shape by the launcher); a pass whose factors would have to be formed from the two-level tables
        // at run time (k > 24) is not taken by this kernel.  Decided once here: the unrolled code below carries no run-time branch.
        const bool need_mul = p.load_tab != nullptr;
        u64 f[16];
#pragma unroll
        for (int q = 0; q < 16; q++) { x[q] = 0; f[q] = 1; }
        auto load_all = [&](auto MODE, auto CANON) {
            if (!act) return;
#pragma unroll
            for (int q = 0; q < 16; q++) {
                if (ld_lim > (int)GM::t_uni(load_map, q)) {
                    u64 v = ntt_ld(p.in + uni_off(0, load_map, q), ld_thr);
                    if (decltype(CANON)::value) v = gl_canon(v);
                    x[q] = v;
                    if (decltype(MODE)::value == 1) f[q] = ntt_ld(p.load_tab + uni_off(2, load_map, q), ldtab_thr);
                }
            }
        };
        using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
        if (p.canon_in) { if (need_mul) load_all(I1{}, std::true_type{}); else load_all(I0{}, std::true_type{}); }
        else { if (need_mul) load_all(I1{}, std::false_type{}); else load_all(I0{}, std::false_type{}); }
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
                if constexpr (SH64 && !std::remove_reference_t<decltype(fx)>::exact) {
                    ntt_tw64<P3, INV>(fx, x, lo_uni);                       // w_{2^(4+P3)}^(rev(i) lo): shifts, lo uniform per wave
                } else if (P3 > 0) {
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
                // the factor that follows the L phase, w_{2^W}^(k_low rev(upper slot bits)), splits into w_{2^(4+P3)}^(k_low rev4(s_mid)) and
                // w_{2^W}^(k_low rev4(s_top)).  SH64: the first part (order 32 / 64: shifts) is applied in the M mapping, where s_mid is the
                // register and k_low the wave's `lo`, and the second joins the factor that follows the M phase
                if constexpr (!SH64) {
#pragma unroll
                    for (int q = 0; q < 16; q++) {
                        const int i = q & ((1 << P3) - 1);
                        if (i == 0) continue;
                        u32 t, slot; map_L(q, t, slot);
                        x[q] = fx.mulw(x[q], shRW[act ? (u32)i * bitrev_u32(slot >> P3, W - P3) : 0]);
                    }
                }
            }
            to_lds(2); __syncthreads(); from_lds(1);
        }
        if (P2 > 0) {
            if (wact) {
                if constexpr (SH64) {
                    if constexpr (!std::remove_reference_t<decltype(fx)>::exact) ntt_tw64<P3, INV>(fx, x, lo_uni);
                    else {
#pragma unroll
                        for (int q = 1; q < 16; q++) x[q] = fx.mulw(x[q], shRW[(BR<4>::rev(q) * lo_uni) << 4]);
                    }
                }
#pragma unroll
                for (int g = 0; g < G2; g++) radix_net<P2, true, INV>(fx, x, g << P2, tw16);
                if constexpr (SH64) {
                    // w_{2^W}^(rev4(s_top) (k_low + 2^P3 k_mid)): k_mid the register, k_low = lo, s_top = hi
                    u32 t, slot; map_M(0, t, slot);
                    const u32 rh = bitrev_u32(slot >> (P2 + P3), 4);
                    if (lo_uni != 0) x[0] = fx.mulw(x[0], shRW[rh * lo_uni]);
#pragma unroll
                    for (int q = 1; q < 16; q++) x[q] = fx.mulw(x[q], shRW[rh * (lo_uni + ((u32)q << P3))]);
                } else {
#pragma unroll
                    for (int q = 0; q < 16; q++) {
                        const int i = q & ((1 << P2) - 1);
                        if (i == 0) continue;
                        u32 t, slot; map_M(q, t, slot);
                        u32 hi = slot >> (P2 + P3);
                        x[q] = fx.mulw(x[q], shRW[act ? ((u32)i * bitrev_u32(hi, 4)) << P3 : 0]);
                    }
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
        // (own one-barrier vote instead of __syncthreads_or: the library's goes through three barriers and keeps the three thread
        // indices alive across the tile loop; the word was cleared behind this tile's first barrier, see the top of the loop)
        if ((gf.fl != 0) | p.force_redo) redo_vote = 1u;
        __syncthreads();
        if (redo_vote != 0) { glx<true> ge; ge.fl = 0; compute(ge); }
#endif
    }

    // ---------------------------------------------------------------- store
    // (the thread id once more from scratch: one register less alive across the networks)
    tid = (wave_id0 << 6) + __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    asm volatile("" : "+v"(tid));
    constexpr int LAST_DIF = FIRST_DIT;
    constexpr int SMAP = (DIT || ROW) ? 0 : LAST_DIF;                  // mapping of the plain store (DIF ROW tiles exchange into mapping 0 first)
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
        // output element lin = tid + 512 q of the tile: row t = lin >> W, position pe = lin & (2^W - 1) inside the row; the thread bits
        // (tid < 512) and the register bits (512 q) are disjoint, and so are their images under the bit reversals below, so the global
        // offset is again a per-thread part (sc_thr) plus a uniform part per register
        const u32 t_thr = tid >> W, pe_thr = tid & ((1u << W) - 1), inner0 = base32 & ((1u << S) - 1);
        u32 sc_thr; int sc_lim = 0x7fffffff;
        if (ROW) {
            if (modeB) { sc_thr = (u32)(col_off(t_thr, p.out_stride, p.out_sub) + pe_thr) << 3; sc_lim = (int)(total32 - row032) - (int)t_thr; }
            else sc_thr = ((t_thr << W) + pe_thr) << 3;
        } else sc_thr = ((bitrev_u32(t_thr, S) << W) + pe_thr) << 3;
        if (act) {                 // (a permuting pass with an output scale is not taken by this kernel: the launcher checks)
#pragma unroll
            for (int q = 0; q < 16; q++) {
                const u32 t_uni = (512u * (u32)q) >> W, pe_uni = (512u * (u32)q) & ((1u << W) - 1);
                if (ROW && !(sc_lim > (int)t_uni)) continue;
                const u32 t = t_thr + t_uni, pe = pe_thr + pe_uni, e = bitrev_u32(pe, W);
                const u64 v = pswz ? sh[pidx(t, pe)] : sh[lds_idx(t, e)];
                size_t ub;
                if (ROW) ub = modeB ? col_off((size_t)row032 + t_uni, p.out_stride, p.out_sub) + pe_uni
                                    : col_off(vcolA, p.out_stride, p.out_sub) + ((rr0 + t_uni) << W) + pe_uni;
                else ub = col_off(vcolA, p.out_stride, p.out_sub) + ((bitrev_u32(inner0, S) + bitrev_u32(t_uni, S)) << W) + pe_uni;
                ntt_st(p.out + ub, sc_thr, v, p.nt_store != 0);
            }
        }
        continue;
    }
    if (!DIT && ROW && LAST_DIF != 0) { __syncthreads(); to_lds(LAST_DIF); __syncthreads(); from_lds(0); }
    // same structure as the load side: guarded batched factor loads, products in wave-uniform control flow, guarded stores;
    // x[] is left intact, so a flagged wave runs the loop again with exact products and overwrites what it stored
    const u32 st_thr = thr_off(1, SMAP), sttab_thr = thr_off(3, SMAP);
    const int st_lim = row_lim(SMAP);
    auto store_plain = [&](auto &fx) {
        const bool from_tab = p.store_tab != nullptr;             // as on the load side: one table, one constant for all, or nothing
        const u64 sc = p.store_const;
        const bool need_mul = from_tab || sc != 0;
        u64 f[16];
#pragma unroll
        for (int q = 0; q < 16; q++) f[q] = from_tab ? 1 : sc;
        if (act && from_tab) {
#pragma unroll
            for (int q = 0; q < 16; q++) {
                if (!(st_lim > (int)GM::t_uni(SMAP, q))) continue;
                f[q] = ntt_ld(p.store_tab + uni_off(3, SMAP, q), sttab_thr);
            }
        }
        if (wact && need_mul) {
#pragma unroll
            for (int q = 0; q < 16; q++) f[q] = fx.mul(x[q], f[q]);            // f now holds the values to store
        }
        if (act) {
#pragma unroll
            for (int q = 0; q < 16; q++) {
                if (!(st_lim > (int)GM::t_uni(SMAP, q))) continue;
                ntt_st(p.out + uni_off(1, SMAP, q), st_thr, need_mul ? f[q] : x[q], ROW && p.nt_store != 0);
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
    }       // tiles of this workgroup
}


// ---- launchers (one translation unit per family: ntt_fast.hip DIF, ntt_fast_dit.hip DIT, ntt_fast_cs.hip coset-merged DIT -- the
// families compile side by side under `make -j`)
template <bool DIT, bool ROW, int W, bool INV, bool CS>
static void ntt_fast_launch_wi(zk_ctx *ctx, const PassParams &p, const char *name) {
    size_t n = (size_t)1 << p.k, blocks;
    const size_t vcols = (size_t)p.n_cols << (CS ? p.cs_bits : 0);
    if (ROW) { size_t rows = (n >> p.W) * vcols; blocks = (rows + ((size_t)1 << p.TL) - 1) >> p.TL; }
    else blocks = ((n >> p.W) >> p.TL) * vcols;
    size_t shmem = (NttGeom<ROW, W>::tile_elems + ((size_t)1 << W)) * sizeof(u64);              // the tile image + the 2^W digit roots
    // persistent form: ZK_NTT_PERSIST workgroups (a multiple of 8; two fit a CU) walk the tiles
    static const int persist = getenv("ZK_NTT_PERSIST") ? atoi(getenv("ZK_NTT_PERSIST")) & ~7 : ZK_NTT_PERSIST_DEFAULT;
    if (persist > 0 && blocks > (size_t)persist) {
        PassParams q = p; q.n_blocks = (unsigned)blocks;
        ZK_LAUNCH(ctx, name, (ntt_fast_kernel<DIT, ROW, W, INV, CS>), dim3((unsigned)persist), dim3(512), shmem, q);
        return;
    }
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
