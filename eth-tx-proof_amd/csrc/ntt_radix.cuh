// Register radix-2^Q butterfly networks of the NTT digit pass (ntt_fast.hip), on 16 elements held in VGPRs; shared with
// tools/ntt_mfma_stage.hip, which times one such network against the same transform done on the i8 matrix cores.
#pragma once
#include "gl64.cuh"

template <int Q> struct BR { __device__ static constexpr u32 rev(u32 i) { u32 r = 0; for (int b = 0; b < Q; b++) if (i & (1u << b)) r |= 1u << (Q - 1 - b); return r; } };

// -x as a weak residue: p - x; a borrow (x in (p, 2^64), about one weak value in 2^32) goes to the sticky word.
// (Kept here and not in gl64.cuh: that header is also the hiprtc include of the constraint kernels and keys their on-disk cache.)
template <class G> __device__ __forceinline__ u64 ntt_neg_weak(G &g, u64 x) {
    if (G::exact) return gl_neg(gl_canon(x));
#if defined(__HIP_DEVICE_COMPILE__)
    u32 xl = (u32)x, xh = (u32)(x >> 32), dl, dh;
    asm("v_sub_co_u32 %[dl], vcc, 1, %[xl]\n\t"
        "s_nop 1\n\t"
        "v_subb_co_u32 %[dh], vcc, -1, %[xh], vcc\n\t"
        "s_or_b64 %[fl], %[fl], vcc\n\t"
        : [dl] "=&v"(dl), [dh] "=&v"(dh), [fl] "+s"(g.fl) : [xl] "v"(xl), [xh] "v"(xh) : "vcc", "scc");
    return ((u64)dh << 32) | dl;
#else
    return 0;
#endif
}
// x * 2^S for any compile-time S mod 192 (2^96 = -1, 2^192 = 1), fast arithmetic only: the order-32 / order-64 inter-phase twiddles of
// the digit pass (w_32 = 2^6, w_64 = 2^3)
template <int S, class G> __device__ __forceinline__ u64 ntt_mul_pow2_192(G &g, u64 x) {
    constexpr int s = ((S % 192) + 192) % 192;
    static_assert((s % 96) == 0 || (s % 32) != 0, "shift twiddle: multiples of 32 other than 0 and 96 are not covered");
    if constexpr (s == 0) return x;
    else if constexpr (s == 96) return ntt_neg_weak(g, x);
    else if constexpr (s < 96) return g.template mul_pow2<s>(x);
    else return ntt_neg_weak(g, g.template mul_pow2<s - 96>(x));
}
// x[i] *= w_{2^(4+P3)}^(+-rev4(i) * LO) for the sixteen registers of an M-phase thread whose low slot bits are LO (uniform per wave)
template <int P3, bool INV, int LO, int I, class G> __device__ __forceinline__ void ntt_tw64_regs(G &g, u64 (&x)[16]) {
    if constexpr (I < 16) {
        constexpr int e = (int)BR<4>::rev(I) * LO * (P3 == 2 ? 3 : 6);            // exponent of 2: w_64 = 2^3, w_32 = 2^6
        x[I] = ntt_mul_pow2_192<INV ? 192 - (e % 192) : e>(g, x[I]);
        ntt_tw64_regs<P3, INV, LO, I + 1>(g, x);
    }
}
template <int P3, bool INV, class G> __device__ __forceinline__ void ntt_tw64(G &g, u64 (&x)[16], u32 lo_uniform) {
    static_assert(P3 == 1 || P3 == 2, "order-32 / order-64 twiddles only");
    if (lo_uniform == 1) ntt_tw64_regs<P3, INV, 1, 1>(g, x);
    if constexpr (P3 == 2) {
        if (lo_uniform == 2) ntt_tw64_regs<P3, INV, 2, 1>(g, x);
        if (lo_uniform == 3) ntt_tw64_regs<P3, INV, 3, 1>(g, x);
    }
}

// radix-2^Q network on registers x[g0 .. g0 + 2^Q); tw16[j] = w_16^j (j < 8) in the pass direction.
// The twiddles inside a network are powers of w_16 = 2^12.  The exact arithmetic multiplies by the value tw16[e]; the fast arithmetic
// shifts: forward w_16^e = 2^(12e); inverse w_16^-e = 2^(192 - 12e) = -2^(96 - 12e), the sign being absorbed by the butterfly that
// follows (DIT) or precedes (DIF) the twiddle: u +- (-t v) swaps the two outputs, (u - v)(-t) = (v - u) t swaps the two inputs.
// So tw16[e] = 2^(12e) forward and 2^(96 - 12e) inverse (the POSITIVE power; round 4: the exact twin takes the sign the same way,
// which makes its twiddles the same small set of constants in both directions).
template <int E16, bool INV, class G> __device__ __forceinline__ u64 tw_mul(G &g, u64 v, const u64 (&tw16)[8]) {
    if (G::exact) return g.mulw(v, tw16[E16]);
    return g.template mul_pow2<INV ? 96 - 12 * E16 : 12 * E16>(v);
}
template <int Q, bool DIT, bool INV, int S, int B0, class G>
__device__ __forceinline__ void radix_layer_step(G &g, u64 (&x)[16], const int g0, const u64 (&tw16)[8]) {
    constexpr int h = DIT ? S : (Q - 1 - S);
    constexpr int i1 = ((B0 >> h) << (h + 1)) | (B0 & ((1 << h) - 1)), j1 = i1 + (1 << h);
    constexpr int e1 = ((i1 & ((1 << h) - 1)) << (Q - 1 - h)) << (4 - Q);            // twiddle as a power of w_16
    constexpr bool neg = INV;                // inverse twiddles are -2^s: both arithmetics multiply by 2^s and let the butterfly take the sign
    if (Q == 1) {
        u64 &u = x[g0 + i1], &v = x[g0 + j1];
        if (DIT) {
            if (e1) v = tw_mul<e1 ? e1 : 1, INV>(g, v, tw16);
            g.bfly1(u, v);
            if (e1 && neg) { u64 t = u; u = v; v = t; }
        } else {
            if (e1 && neg) { u64 t = u; u = v; v = t; g.bfly1(u, v); }
            else g.bfly1(u, v);
            if (e1) v = tw_mul<e1 ? e1 : 1, INV>(g, v, tw16);
        }
        return;
    }
    constexpr int B1 = B0 + 1;
    constexpr int i2 = ((B1 >> h) << (h + 1)) | (B1 & ((1 << h) - 1)), j2 = i2 + (1 << h);
    constexpr int e2 = ((i2 & ((1 << h) - 1)) << (Q - 1 - h)) << (4 - Q);
    u64 &u1 = x[g0 + i1], &v1 = x[g0 + j1], &u2 = x[g0 + i2], &v2 = x[g0 + j2];
    if (DIT) {
        if (e1) v1 = tw_mul<e1 ? e1 : 1, INV>(g, v1, tw16);
        if (e2) v2 = tw_mul<e2 ? e2 : 1, INV>(g, v2, tw16);
        g.bfly2(u1, v1, u2, v2);
        if (e1 && neg) { u64 t = u1; u1 = v1; v1 = t; }
        if (e2 && neg) { u64 t = u2; u2 = v2; v2 = t; }
    } else {
        // (u, v) -> (u + v, (u - v) t): with t = -2^s feed the butterfly (v, u) and put the sum back in u's place
        if (e1 && neg) { u64 t = u1; u1 = v1; v1 = t; }
        if (e2 && neg) { u64 t = u2; u2 = v2; v2 = t; }
        g.bfly2(u1, v1, u2, v2);
        if (e1) v1 = tw_mul<e1 ? e1 : 1, INV>(g, v1, tw16);
        if (e2) v2 = tw_mul<e2 ? e2 : 1, INV>(g, v2, tw16);
    }
}
template <int Q, bool DIT, bool INV, int S, int B0, class G>
__device__ __forceinline__ void radix_layer(G &g, u64 (&x)[16], const int g0, const u64 (&tw16)[8]) {
    if constexpr (B0 < (1 << (Q - 1))) {
        radix_layer_step<Q, DIT, INV, S, B0>(g, x, g0, tw16);
        radix_layer<Q, DIT, INV, S, B0 + 2>(g, x, g0, tw16);
    }
}
template <int Q, bool DIT, bool INV, int S, class G>
__device__ __forceinline__ void radix_layers(G &g, u64 (&x)[16], const int g0, const u64 (&tw16)[8]) {
    if constexpr (S < Q) {
        radix_layer<Q, DIT, INV, S, 0>(g, x, g0, tw16);
        radix_layers<Q, DIT, INV, S + 1>(g, x, g0, tw16);
    }
}
template <int Q, bool DIT, bool INV, class G>
__device__ __forceinline__ void radix_net(G &g, u64 (&x)[16], const int g0, const u64 (&tw16)[8]) {
    radix_layers<Q, DIT, INV, 0>(g, x, g0, tw16);
}

