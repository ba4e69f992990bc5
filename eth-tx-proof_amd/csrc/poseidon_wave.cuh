// Wave-wide Poseidon permutation: ONE permutation per 64-lane wave, built for latency.  Same function as psd::permute
// (poseidon.cuh; plonky2 0.2.2 hash/poseidon.rs `Poseidon::poseidon`).  It is what a serial sponge needs -- the
// Fiat-Shamir transcript absorbing thousands of opening values is a chain of permutations nothing can run beside -- and what the
// top levels of a Merkle tree need once a level has fewer nodes than the chip has SIMDs.
//
// A single wave issues one instruction at a time, so latency = instruction count.  Against the 16-lane version
// (poseidon_coop.cuh, ~6100 instructions per permutation) this one runs ~2700:
//   * full rounds: state word l in lane l; the MDS row of lane l sits in 12 registers and the twelve state words arrive as
//     wave-uniform SGPR pairs (v_readlane) -- 24 v_mad_u64_u32 per layer, no LDS-path shuffles, no index arithmetic;
//   * partial rounds: the 22 rounds are one serial chain  u_q = sbox(x0_q) + K_q  with every other word eliminated
//     (tools/gen_constants.py derive_wave): lane k accumulates "x0 before round k", lanes 32..43 the final state, and a round is
//     readlane(x0) -> s-box on the uniform value -> ONE multiply-add per lane against the lane's coefficient of u_q.
//     No cross-lane reduction at all (the 16-lane version spends 4 shuffle+add stages per round on its dot product).
// Tables: PSD_WV_INIT[12][64], PSD_WV_CHAIN[22][64] (lane 63 = the round's constant K), PSD_WV_MDS[12][64], PSD_WV_RC[9][64] (the
// per-lane constants of the full rounds and PR_FIRST) -- 28 KB, staged in LDS by the caller (fill()): every constant the dependent
// chain needs arrives by a prefetched ds_read, none by a global or scalar load in the middle of the chain.
//
// Fast path (round 3): a wave alone on its SIMD pays 5.5-9.5 cycles for EVERY instruction it issues, pads included
// (tools/ubench_lat.hip), so the two round bodies are hand-scheduled instruction sequences (poseidon_wave_asm.inc, generated, simulated
// and hazard-checked by tools/gen_wave_asm.py): x^3 and x^4 of an s-box interleaved instruction by instruction, v_readlane two MDS terms
// ahead of its multiply-adds, the next layer's constants as the starting values of the MDS column sums, the partial rounds' constants
// K[q] folded into the lanes' starting values (PSD_WV_KSUM), loops unrolled inside the blocks with their table reads as ds_read one round
// ahead: 117 / 82 instructions per full / partial round against 148 / 120 + the compiler's loop code.
// The exact twin (every repair in place, taken when the sticky word is up) stays the compiled code below.
#pragma once
#include "poseidon.cuh"
#include "poseidon_wave_asm.inc"

#ifndef PSDW_USE_ASM
#define PSDW_USE_ASM 1       // 0: the compiled round bodies on the fast path as well (A/B builds)
#endif

namespace psdw {

constexpr int TAB_WORDS = (12 + 22 + 12 + 9 + 1) * 64;      // 28.5 KB

// all threads of the block copy the tables into `lds` (TAB_WORDS u64); caller synchronises afterwards.  NT = threads of the block when the
// caller knows it (64 for the transcript kernel, 256 for the level kernels): the copy is then 16 bytes per thread and step with constant
// trip counts, i.e. all loads in flight at once -- with the run-time loop a 64-thread block paid 56 dependent load-store steps, 5-6 us
// of a transcript launch that carries no permutation at all.
typedef u64 psdw_v2 __attribute__((ext_vector_type(2)));
template <int NT, int WORDS> __device__ __forceinline__ void copy_tab(u64 *dst, const u64 *src) {
    constexpr int PAIRS = WORDS / 2;
#pragma unroll
    for (int i = 0; i < (PAIRS + NT - 1) / NT; i++) {
        const int k = i * NT + (int)threadIdx.x;
        if (k < PAIRS) ((psdw_v2 *)dst)[k] = ((const psdw_v2 *)src)[k];
    }
}
template <int NT = 0> __device__ __forceinline__ void fill(u64 *lds) {
    if (NT > 0) {
        copy_tab<NT ? NT : 64, 12 * 64>(lds, PSD_WV_INIT);
        copy_tab<NT ? NT : 64, 22 * 64>(lds + 12 * 64, PSD_WV_CHAIN);
        copy_tab<NT ? NT : 64, 12 * 64>(lds + 34 * 64, PSD_WV_MDS);
        copy_tab<NT ? NT : 64, 9 * 64>(lds + 46 * 64, PSD_WV_RC);
        copy_tab<NT ? NT : 64, 64>(lds + 55 * 64, PSD_WV_KSUM);
        return;
    }
    for (int i = threadIdx.x; i < 12 * 64; i += blockDim.x) lds[i] = PSD_WV_INIT[i];
    for (int i = threadIdx.x; i < 22 * 64; i += blockDim.x) lds[12 * 64 + i] = PSD_WV_CHAIN[i];
    for (int i = threadIdx.x; i < 12 * 64; i += blockDim.x) lds[34 * 64 + i] = PSD_WV_MDS[i];
    for (int i = threadIdx.x; i < 9 * 64; i += blockDim.x) lds[46 * 64 + i] = PSD_WV_RC[i];
    for (int i = threadIdx.x; i < 64; i += blockDim.x) lds[55 * 64 + i] = PSD_WV_KSUM[i];
}

__device__ __forceinline__ u64 rdlane64(u64 v, int l) {
    u32 lo = (u32)__builtin_amdgcn_readlane((int)(u32)v, l), hi = (u32)__builtin_amdgcn_readlane((int)(u32)(v >> 32), l);
    return ((u64)hi << 32) | lo;
}

// EXACT = false: the products skip their one-in-2^32 borrow repair and flag it in pc.fl instead (poseidon.cuh psd::core);
// permute() below redoes the permutation with every repair in place when the flag is up -- results identical either way.
template <bool EXACT> struct perm {
    u32 m[12];              // MDS coefficients of this lane's output row: out_l = sum_i m[i] * s_i
    int lane, lc;
    __attribute__((address_space(3))) const u64 *tab;         // LDS copy of the tables, typed as LDS: ds_read, not flat_load
    psd::core<EXACT> pc;

    __device__ __forceinline__ void init(int lane_, const u64 *lds_tab) {
        lane = lane_; lc = lane < 12 ? lane : 0; tab = (__attribute__((address_space(3))) const u64 *)lds_tab; pc.fl = 0;
        // coefficient of s_i in out_l: CIRC[(i - l) mod 12], + 8 on the diagonal of row 0; rows of idle lanes are zero
#pragma unroll
        for (int i = 0; i < 12; i++) m[i] = (u32)tab[34 * 64 + 64 * i + lane];
    }

    // four full rounds starting at `round`; weak in (lanes 0..11), weak out, idle lanes come out 0
    __device__ __forceinline__ u64 full_rounds(u64 s, int round) {
        const auto *rcrow = tab + 46 * 64 + (round ? 4 * 64 : 0) + lane;
        u64 rcn = rcrow[0];
#pragma unroll 1
        for (int r = 0; r < 4; r++) {
            const u64 rc = rcn;
            rcn = rcrow[64 * (r < 3 ? r + 1 : 3)];
            s = pc.sbox(psd::wadd_c(s, rc));
            const u32 lo = (u32)s, hi = (u32)(s >> 32);
            u64 al = 0, ah = 0;
#pragma unroll
            for (int i = 0; i < 12; i++) {
                const u32 xl = (u32)__builtin_amdgcn_readlane((int)lo, i), xh = (u32)__builtin_amdgcn_readlane((int)hi, i);
                al += (u64)xl * m[i]; ah += (u64)xh * m[i];
            }
            // al + ah*2^32, both < 2^41: low64 + high*2^64 with high < 2^9 -> low + high*eps
            u32 th, cy;
            th = __builtin_addc((u32)(al >> 32), (u32)ah, 0u, &cy);
            const u32 high = (u32)(ah >> 32) + cy;
            s = gl_mad_eps(high, ((u64)th << 32) | (u32)al);
        }
        return s;
    }

    // ---- fast path: the generated round bodies (three asm statements per permutation; their table reads are ds_read inside the blocks)
    __device__ __forceinline__ u64 run_fast(u64 s) {
        const u32 rc_addr = (u32)(unsigned long)(tab + 46 * 64 + lane);        // this lane's cell of row 0 of the per-lane round constants
        const u32 ch_addr = (u32)(unsigned long)(tab + 12 * 64 + lane);        // ... of row 0 of the chain coefficients
        u64 f = 0;
        // rounds 0..3: the constants of round 0 here, those of rounds 1..3 and PR_FIRST (row 8) by the MDS layers
        s = psd::wadd_c(s, tab[46 * 64 + lane]);
        u32 sl = (u32)s, sh = (u32)(s >> 32);
        asm volatile(PSDW_FULL_ROUNDS_FIRST_ASM
                     : [sl] "+v"(sl), [sh] "+v"(sh), [fl] "+s"(f)
                     : [addr] "v"(rc_addr), [m0] "v"(m[0]), [m1] "v"(m[1]), [m2] "v"(m[2]), [m3] "v"(m[3]), [m4] "v"(m[4]), [m5] "v"(m[5]),
                       [m6] "v"(m[6]), [m7] "v"(m[7]), [m8] "v"(m[8]), [m9] "v"(m[9]), [m10] "v"(m[10]), [m11] "v"(m[11])
                     : PSDW_ASM_CLOBBERS);
        // ---- 22 partial rounds: z = state + PR_FIRST is what came out
        const u64 z = ((u64)sh << 32) | sl;
        u64 acc;
        {
            psd::acc3 a; a.init();
            a.c0 = tab[55 * 64 + lane];                  // the K[q] of all rounds, folded (PSD_WV_KSUM)
#pragma unroll
            for (int i = 0; i < 12; i++) a.mac_s(rdlane64(z, i), tab[64 * i + lane]);
            acc = pc.reduce(a);
        }
        u32 al = (u32)acc, ah = (u32)(acc >> 32);
        asm volatile(PSDW_PART_ROUNDS_ASM
                     : [al] "+v"(al), [ah] "+v"(ah), [fl] "+s"(f)
                     : [addr] "v"(ch_addr)
                     : PSDW_ASM_CLOBBERS);
        // final state sits in lanes 32..43
        al = __shfl(al, (lane + 32) & 63, 64); ah = __shfl(ah, (lane + 32) & 63, 64);
        // rounds 26..29: rows 4..7, nothing after the last
        s = psd::wadd_c(((u64)ah << 32) | al, tab[46 * 64 + 4 * 64 + lane]);
        sl = (u32)s; sh = (u32)(s >> 32);
        asm volatile(PSDW_FULL_ROUNDS_LAST_ASM
                     : [sl] "+v"(sl), [sh] "+v"(sh), [fl] "+s"(f)
                     : [addr] "v"(rc_addr), [m0] "v"(m[0]), [m1] "v"(m[1]), [m2] "v"(m[2]), [m3] "v"(m[3]), [m4] "v"(m[4]), [m5] "v"(m[5]),
                       [m6] "v"(m[6]), [m7] "v"(m[7]), [m8] "v"(m[8]), [m9] "v"(m[9]), [m10] "v"(m[10]), [m11] "v"(m[11])
                     : PSDW_ASM_CLOBBERS);
        pc.fl |= f;
        s = ((u64)sh << 32) | sl;
        return lane < 12 ? gl_canon(s) : 0;
    }

    // s: state word `lane` for lane < 12 (any 64-bit representative), ignored elsewhere.  Returns the permuted word, canonical, in
    // lanes 0..11 and 0 in the others.  Must be called by all 64 lanes.
    __device__ __forceinline__ u64 run(u64 s) {
        if (!EXACT && PSDW_USE_ASM) return run_fast(s);
        s = full_rounds(s, 0);
        // ---- 22 partial rounds
        const u64 z = psd::wadd_c(s, tab[46 * 64 + 8 * 64 + lane]);
        u64 acc;
        {
            psd::acc3 a; a.init();
#pragma unroll
            for (int i = 0; i < 12; i++) a.mac_s(rdlane64(z, i), tab[64 * i + lane]);
            acc = pc.reduce(a);
        }
        const auto *ch = tab + 12 * 64 + lane;
        u64 cnext = ch[0];
#pragma unroll 2
        for (int q = 0; q < 22; q++) {
            const u64 c = cnext;
            cnext = ch[64 * (q < 21 ? q + 1 : 21)];
            const u64 x0 = rdlane64(acc, q);
            const u64 u = psd::wadd_c(pc.sbox(x0), rdlane64(c, 63));
            acc = pc.mul_add(c, u, acc);
        }
        // final state sits in lanes 32..43
        {
            u32 lo = (u32)acc, hi = (u32)(acc >> 32);
            lo = __shfl(lo, (lane + 32) & 63, 64); hi = __shfl(hi, (lane + 32) & 63, 64);
            s = ((u64)hi << 32) | lo;
        }
        s = full_rounds(s, 26);
        return lane < 12 ? gl_canon(s) : 0;
    }
};

// s: state word `lane` for lane < 12, any representative; all 64 lanes must call.  `tab` = LDS tables (fill()).
// force_redo (wave-uniform, ZK_FORCE_EXACT_REDO): always take the exact branch as well -- for the tests.
__device__ __forceinline__ u64 permute(u64 s, int lane, const u64 *tab, u64 force_redo = 0) {
    perm<false> pf; pf.init(lane, tab);
    u64 r = pf.run(s);
    if (pf.pc.fl | force_redo) { perm<true> pe; pe.init(lane, tab); r = pe.run(s); }
    return r;
}

}  // namespace psdw
