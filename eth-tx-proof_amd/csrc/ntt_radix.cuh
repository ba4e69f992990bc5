// Register radix-2^Q butterfly networks of the NTT digit pass (ntt_fast.hip), on 16 elements held in VGPRs; shared with
// tools/ntt_mfma_stage.hip, which times one such network against the same transform done on the i8 matrix cores.
#pragma once
#include "gl64.cuh"

template <int Q> struct BR { __device__ static constexpr u32 rev(u32 i) { u32 r = 0; for (int b = 0; b < Q; b++) if (i & (1u << b)) r |= 1u << (Q - 1 - b); return r; } };

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

