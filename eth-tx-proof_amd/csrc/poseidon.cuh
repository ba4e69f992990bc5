// Poseidon-Goldilocks permutation for gfx950 (width 12, rate 8, x^7, 4+22+4 rounds).
// Mirrors plonky2 0.2.2 hash/poseidon.rs `Poseidon::poseidon` + poseidon_goldilocks.rs constants
// (crate pinned at /root/reference/Cargo.lock:3441-3443; spec and known-answer vectors: SURVEY.md A2).
// One thread = one permutation, state in VGPRs.
//
// Arithmetic is "weak" inside the permutation: state words are any representative in [0, 2^64) of their residue
// (as plonky2's own reduce128 returns), which removes every compare-and-select canonicalisation from the inner loops;
// the 12 outputs are canonicalised once at the end.  On gfx950 every VALU op (v_mad_u64_u32 included) issues at
// about the same rate (profiles/r01_ubench_int_gfx950.txt), so the cost model is simply the instruction count:
//   * full rounds: small-constant circulant MDS on 32-bit halves (v_mad_u64_u32 chains, one fold per lane);
//   * partial rounds: sparse factorisation M = Sp_i * blockdiag(1, M_hat) derived and verified by tools/gen_constants.py,
//     dot product accumulated in five 32-bit limbs with hardware carry chains (v_addc_co_u32).
#pragma once
#include "gl64.cuh"

#define PSD_CONST static __device__ const
#include "poseidon_constants.inc"

namespace psd {

PSD_CONST u64 PSD_ZERO12[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

// 64x64 -> 128 with four v_mad_u64_u32 (the running sums never overflow 64 bits)
__device__ __forceinline__ void mul128(u64 a, u64 b, u64 &lo, u64 &hi) {
    u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    u64 A1 = (u64)a0 * b0;
    u64 A2 = (u64)a0 * b1 + (A1 >> 32);
    u64 A3 = (u64)a1 * b0 + (u32)A2;
    u64 A4 = (u64)a1 * b1 + (A2 >> 32) + (A3 >> 32);
    lo = (A3 << 32) | (u32)A1; hi = A4;
}
// weak reduction of hi:lo: lo + h0*eps - h1 with 2^64 = eps, 2^96 = -1; each wrap is repaired by +-eps and cannot wrap twice
__device__ __forceinline__ u64 wred128(u64 lo, u64 hi) {
    u32 h0 = (u32)hi, h1 = (u32)(hi >> 32);
    u64 r1 = gl_mad_eps(h0, lo);
    u64 r2; bool b1 = __builtin_sub_overflow(r1, (u64)h1, &r2);
    r2 -= b1 ? GL_EPS : 0ull;
    return r2;
}
__device__ __forceinline__ u64 wmul(u64 a, u64 b) { u64 lo, hi; mul128(a, b, lo, hi); return wred128(lo, hi); }
// weak a + canonical c: the sum is below 2^64 + p, so one eps repair suffices
__device__ __forceinline__ u64 wadd_c(u64 a, u64 c) {
    u64 s; bool cy = __builtin_add_overflow(a, c, &s);
    return s + (cy ? GL_EPS : 0ull);
}
__device__ __forceinline__ u64 wsbox7(u64 x) {
    u64 x2 = wmul(x, x), x3 = wmul(x2, x), x4 = wmul(x2, x2);
    return wmul(x3, x4);
}
__device__ __forceinline__ u64 sbox7(u64 x) { return gl_canon(wsbox7(x)); }   // canonical in, canonical out

// five-limb accumulator for sums of up to 16 products of 64-bit words; carries ride the hardware carry chain
struct acc160 {
    u32 l0, l1, l2, l3, l4;
    __device__ __forceinline__ void init() { l0 = l1 = l2 = l3 = l4 = 0; }
    __device__ __forceinline__ void mac(u64 a, u64 b) {
        u64 pl, ph; mul128(a, b, pl, ph);
        u32 c;
        l0 = __builtin_addc(l0, (u32)pl, 0u, &c);
        l1 = __builtin_addc(l1, (u32)(pl >> 32), c, &c);
        l2 = __builtin_addc(l2, (u32)ph, c, &c);
        l3 = __builtin_addc(l3, (u32)(ph >> 32), c, &c);
        l4 += c;
    }
    // value = lo64 + hi64*2^64 + l4*2^128,  2^128 = -2^32 (mod p); l4 < 16.  Weak result.
    __device__ __forceinline__ u64 reduce() const {
        u64 r = wred128(((u64)l1 << 32) | l0, ((u64)l3 << 32) | l2);
        u64 sub = (u64)l4 << 32;
        u64 d; bool b = __builtin_sub_overflow(r, sub, &d);
        return d - (b ? GL_EPS : 0ull);
    }
};
// kept under the old name for the lane-cooperative variant
typedef acc160 acc192;

// Column accumulator for dot products  sum_j const_j * s_j  with wave-uniform constants (SGPR operands).
// The three 32-bit-aligned product columns (weights 1, 2^32, 2^64) are accumulated separately with v_mad_u64_u32, whose
// carry-out (an SGPR lane mask the compiler cannot express from C++) is counted by v_addc_co_u32: 8 instructions per term,
// no carry propagation between columns.  One block = 4 mads then 4 addc, so every carry mask is consumed >= 3
// instructions after it is produced (gfx950 needs 2 wait states between a VALU SGPR write and its VALU use).
struct acc3 {
    u64 c0, c1, c2; u32 o0, o1, o2;
    __device__ __forceinline__ void init() { c0 = c1 = c2 = 0; o0 = o1 = o2 = 0; }
    __device__ __forceinline__ void mac_s(u64 a /* wave-uniform */, u64 b) {
        u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
        u64 k0, k1, k2, k3;
        // (leading s_nop 1: under SGPR pressure hipcc parks constants in VGPR lanes and restores them with v_readlane right
        // before this block -- a vector write of an SGPR needs two wait states before a vector read, and hipcc cannot see into it)
        asm volatile(
            "s_nop 1\n\t"
            "v_mad_u64_u32 %[c0], %[k0], %[a0], %[b0], %[c0]\n\t"
            "v_mad_u64_u32 %[c1], %[k1], %[a0], %[b1], %[c1]\n\t"
            "v_mad_u64_u32 %[c2], %[k2], %[a1], %[b1], %[c2]\n\t"
            "v_mad_u64_u32 %[c1], %[k3], %[a1], %[b0], %[c1]\n\t"
            "v_addc_co_u32 %[o0], %[k0], 0, %[o0], %[k0]\n\t"
            "v_addc_co_u32 %[o1], %[k1], 0, %[o1], %[k1]\n\t"
            "v_addc_co_u32 %[o2], %[k2], 0, %[o2], %[k2]\n\t"
            "v_addc_co_u32 %[o1], %[k3], 0, %[o1], %[k3]\n\t"
            : [c0] "+v"(c0), [c1] "+v"(c1), [c2] "+v"(c2), [o0] "+v"(o0), [o1] "+v"(o1), [o2] "+v"(o2),
              [k0] "=&s"(k0), [k1] "=&s"(k1), [k2] "=&s"(k2), [k3] "=&s"(k3)
            : [a0] "s"(a0), [a1] "s"(a1), [b0] "v"(b0), [b1] "v"(b1));
    }
};

// The permutation proper.  Every reduction ends in "r - h" with h a 32-bit value and r freshly reduced, so the subtraction
// borrows only when r < 2^32: about one lane in 2^32.  The straight-line compare + select repair of that borrow was 4 of the
// ~19 VALU instructions of every product.  core<false> skips it: the borrow mask that v_subbrev_co_u32 delivers anyway is
// OR-ed into a wave-uniform sticky word `fl` with one SALU instruction, and the kernel that ran the fast path re-runs the
// same input through core<true> (all repairs in place) when fl != 0 -- results are identical either way, bit for bit.
// x + acc as ONE v_mad_u64_u32 (x*1 + acc): adds a 32-bit word to a 64-bit pair.  The C++ spelling (u64)x + acc makes hipcc
// build the zero-extended register pair {x, 0} first -- one or two v_mov_b32 per addition, a third of a full round.
__device__ __forceinline__ u64 madd1(u32 x, u64 acc) {
    u64 r, k;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %3" : "=v"(r), "=s"(k) : "v"(x), "v"(acc));
    return r;
}

template <bool EXACT> struct core {
    u64 fl;
    // r - h - (cin ? 1 : 0), cin a lane mask (the carry-out of a v_mad_u64_u32, weight 2^96 = -1)
    __device__ __forceinline__ u64 sub32c(u64 r, u32 h, u64 cin) {
        u32 rl = (u32)r, rh = (u32)(r >> 32), dl, dh; u64 k;
        if (EXACT) {
            // borrow out of the 64-bit subtraction: the true value is negative, + p = - eps (mod 2^64)
            u32 m;
            asm("v_subb_co_u32 %[dl], %[k], %[rl], %[h], %[c]\n\t"
                "s_nop 1\n\t"
                "v_subbrev_co_u32 %[dh], %[k], 0, %[rh], %[k]\n\t"
                "s_nop 1\n\t"
                "v_cndmask_b32_e64 %[m], 0, -1, %[k]\n\t"
                : [dl] "=&v"(dl), [dh] "=&v"(dh), [k] "=&s"(k), [m] "=&v"(m)
                : [rl] "v"(rl), [rh] "v"(rh), [h] "v"(h), [c] "s"(cin));
            return (((u64)dh << 32) | dl) - (u64)m;
        } else {
            // (leading s_nop 0: cin may have been restored from a VGPR lane by v_readlane just before this block; with hipcc's own
            // boundary state that makes the two wait states a vector-written SGPR needs before a vector read)
            asm("s_nop 0\n\t"
                "v_subb_co_u32 %[dl], %[k], %[rl], %[h], %[c]\n\t"
                "s_nop 1\n\t"
                "v_subbrev_co_u32 %[dh], %[k], 0, %[rh], %[k]\n\t"
                : [dl] "=&v"(dl), [dh] "=&v"(dh), [k] "=&s"(k)
                : [rl] "v"(rl), [rh] "v"(rh), [h] "v"(h), [c] "s"(cin));
            fl |= k;
            return ((u64)dh << 32) | dl;
        }
    }
    __device__ __forceinline__ u64 sub32(u64 r, u32 h) {
        if (EXACT) {
            u64 d; bool b = __builtin_sub_overflow(r, (u64)h, &d);
            return d - (b ? GL_EPS : 0ull);
        } else {
            u32 rl = (u32)r, rh = (u32)(r >> 32), dl, dh; u64 k;
            asm("v_sub_co_u32 %[dl], %[k], %[rl], %[h]\n\t"
                "s_nop 1\n\t"
                "v_subbrev_co_u32 %[dh], %[k], 0, %[rh], %[k]\n\t"
                : [dl] "=&v"(dl), [dh] "=&v"(dh), [k] "=&s"(k)
                : [rl] "v"(rl), [rh] "v"(rh), [h] "v"(h));
            fl |= k;
            return ((u64)dh << 32) | dl;
        }
    }
    __device__ __forceinline__ u64 red128(u64 lo, u64 hi) { return sub32(gl_mad_eps((u32)hi, lo), (u32)(hi >> 32)); }
    // 12 VALU instructions: six v_mad_u64_u32 form  a*b + c = lo + hi*2^64 + cy*2^96  (the mid column a0*b1 + a1*b0 keeps its
    // carry-out cy as a lane mask instead of propagating it), one v_mov pairs the low words, then lo + hi0*eps - hi1 - cy
    // where cy rides the borrow-in of the final subtraction.
    __device__ __forceinline__ u64 mul(u64 a, u64 b) {
        u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
        u64 P0 = (u64)a0 * b0;
        u64 A2 = madd1((u32)(P0 >> 32), (u64)a0 * b1);                  // <= (2^32-1)^2 + 2^32-1: no overflow
        u64 X, cy;
        asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(X), "=s"(cy) : "v"(a1), "v"(b0), "v"(A2));
        u64 A4 = madd1((u32)(X >> 32), (u64)a1 * b1);
        u64 lo = (X << 32) | (u32)P0;
        return sub32c(gl_mad_eps((u32)A4, lo), (u32)(A4 >> 32), cy);
    }
    // a*b + c for any 64-bit a, b, c (cannot overflow 128 bits): c's words join the first two columns
    __device__ __forceinline__ u64 mul_add(u64 a, u64 b, u64 c) {
        u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
        u64 P0 = madd1((u32)c, (u64)a0 * b0);
        u64 A2 = madd1((u32)(c >> 32), madd1((u32)(P0 >> 32), (u64)a0 * b1));   // <= (2^32-1)^2 + 2(2^32-1) = 2^64-1
        u64 X, cy;
        asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(X), "=s"(cy) : "v"(a1), "v"(b0), "v"(A2));
        u64 A4 = madd1((u32)(X >> 32), (u64)a1 * b1);
        u64 lo = (X << 32) | (u32)P0;
        return sub32c(gl_mad_eps((u32)A4, lo), (u32)(A4 >> 32), cy);
    }
    __device__ __forceinline__ u64 sbox(u64 x) {
        u64 x2 = mul(x, x), x3 = mul(x2, x), x4 = mul(x2, x2);
        return mul(x3, x4);
    }
    // acc3 value = c0 + c1*2^32 + c2*2^64 + o0*2^64 + o1*2^96 + o2*2^128;  2^64 = eps, 2^96 = -1, 2^128 = -2^32.  Weak result.
    __device__ __forceinline__ u64 reduce(const acc3 &a) {
        u64 L; bool cL = __builtin_add_overflow(a.c0, a.c1 << 32, &L);
        u64 H1 = a.c2 + (a.c1 >> 32);
        bool w1 = H1 < a.c2;
        u64 H = H1 + (cL ? 1 : 0);
        bool w2 = H < H1;
        u32 top = a.o2 + (w1 ? 1 : 0) + (w2 ? 1 : 0);
        u64 r = red128(L, H);
        u64 r1 = gl_mad_eps(a.o0, r);
        u64 sub = (u64)a.o1 + ((u64)top << 32);      // can exceed 32 bits: keep the straight-line repair here (22 per permutation)
        u64 r2; bool b = __builtin_sub_overflow(r1, sub, &r2);
        return r2 - (b ? GL_EPS : 0ull);
    }
// out[r] = sum_i s[(i+r)%12]*CIRC[i] + (r==0)*8*s[0] + k[r],  CIRC = [17,15,41,16,2,28,13,13,39,18,34,20]; weak in, weak out.
// k = the constants the NEXT layer would add (wave-uniform, canonical; PSD_ZERO12 = none): their two words start the two
// column sums as SGPR-pair addends of the first v_mad_u64_u32, so the constant layer costs no vector instruction.
    __device__ __forceinline__ void mds_layer(u64 s[12], const u64 *__restrict__ k) {
        const u32 C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
        u32 lo[12], hi[12];
#pragma unroll
        for (int i = 0; i < 12; i++) { lo[i] = (u32)s[i]; hi[i] = (u32)(s[i] >> 32); }
#pragma unroll
        for (int r = 0; r < 12; r++) {
            u64 al = 0, ah = 0;
            { u64 kr = k[r]; al = (u32)kr; ah = kr >> 32; }
#pragma unroll
            for (int i = 0; i < 12; i++) {
                al += (u64)lo[(i + r) % 12] * C[i];
                ah += (u64)hi[(i + r) % 12] * C[i];
            }
            if (r == 0) { al += (u64)lo[0] * 8u; ah += (u64)hi[0] * 8u; }
            // value = al + ah*2^32 with al, ah < 2^43:  low64 + high*2^64, high < 2^12  ->  low + high*eps, one eps repair
            // (32-bit add into al's high word + carry into ah's high word: two instructions, no register-pair shuffling)
            u32 th, cy;
            th = __builtin_addc((u32)(al >> 32), (u32)ah, 0u, &cy);
            u32 high = (u32)(ah >> 32) + cy;
            s[r] = gl_mad_eps(high, ((u64)th << 32) | (u32)al);
        }
    }

    // four full rounds starting at round `first_round`; `after` = constants of the layer that follows the fourth
    __device__ __forceinline__ void full_rounds(u64 s[12], int first_round, const u64 *__restrict__ after) {
        {
            const u64 *rc = PSD_RC + 12 * first_round;
#pragma unroll
            for (int i = 0; i < 12; i++) s[i] = wadd_c(s[i], rc[i]);
        }
#pragma unroll 1
        for (int r = 0; r < 4; r++) {
#pragma unroll
            for (int i = 0; i < 12; i++) s[i] = sbox(s[i]);
            mds_layer(s, r < 3 ? PSD_RC + 12 * (first_round + r + 1) : after);
        }
    }

    __device__ __forceinline__ void partial_rounds(u64 s[12]) {
        {   // (PSD_PR_FIRST was added by the MDS layer of the last full round)  s[1..] = PR_INIT * s[1..]
            u64 t[11];
#pragma unroll
            for (int i = 0; i < 11; i++) {
                acc3 a; a.init();
#pragma unroll
                for (int j = 0; j < 11; j++) a.mac_s(PSD_PR_INIT[11 * i + j], s[1 + j]);
                t[i] = reduce(a);
            }
#pragma unroll
            for (int i = 0; i < 11; i++) s[1 + i] = t[i];
        }
#pragma unroll 1
        for (int r = 0; r < 22; r++) {
            u64 s0 = wadd_c(sbox(s[0]), PSD_PR_K[r]);
            acc3 a; a.init();
            a.c0 = (u64)(u32)s0 * (u32)PSD_M00; a.c1 = (s0 >> 32) * (u32)PSD_M00;      // 25 * s0 in the first two columns
            const u64 *vh = PSD_PR_VHAT + 11 * r, *w = PSD_PR_W + 11 * r;
#pragma unroll
            for (int j = 0; j < 11; j++) a.mac_s(vh[j], s[1 + j]);
#pragma unroll
            for (int j = 0; j < 11; j++) s[1 + j] = mul_add(w[j], s0, s[1 + j]);
            s[0] = reduce(a);
        }
    }

    // any 64-bit representatives in; canonical out
    __device__ __forceinline__ void run(u64 s[12]) {
        full_rounds(s, 0, PSD_PR_FIRST);
        partial_rounds(s);
        full_rounds(s, 26, PSD_ZERO12);
#pragma unroll
        for (int i = 0; i < 12; i++) s[i] = gl_canon(s[i]);
    }
};

// any 64-bit representatives in; canonical out.  All repairs in place: for callers that are not hot.
__device__ __forceinline__ void permute(u64 s[12]) { core<true> c; c.fl = 0; c.run(s); }

}  // namespace psd
