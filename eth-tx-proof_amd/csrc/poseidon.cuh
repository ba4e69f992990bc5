// Poseidon-Goldilocks permutation for gfx950 (width 12, rate 8, x^7, 4+22+4 rounds).
// Mirrors plonky2 0.2.2 hash/poseidon.rs `Poseidon::poseidon` + poseidon_goldilocks.rs constants
// (crate pinned at /root/reference/Cargo.lock:3441-3443; spec and known-answer vectors: SURVEY.md A2).
// One thread = one permutation, state in VGPRs.  Full rounds use the small-constant circulant MDS on 32-bit
// halves (v_mad_u64_u32 chains, one reduction per lane); partial rounds use the sparse factorisation
// M = Sp_i * blockdiag(1, M_hat) derived and verified by tools/gen_constants.py.
#pragma once
#include "gl64.cuh"

#define PSD_CONST static __device__ const
#include "poseidon_constants.inc"

namespace psd {

__device__ __forceinline__ u64 sbox7(u64 x) {
    u64 x2 = gl_sqr(x), x3 = gl_mul(x2, x), x4 = gl_sqr(x2);
    return gl_mul(x3, x4);
}

// 3-word accumulator for sums of up to 16 64x64-bit products
struct acc192 {
    u64 lo, hi; u32 top;
    __device__ __forceinline__ void init() { lo = 0; hi = 0; top = 0; }
    __device__ __forceinline__ void mac(u64 a, u64 b) {
        u64 pl = a * b, ph = __umul64hi(a, b);
        lo += pl;
        u64 c = lo < pl;
        u64 h1 = hi + ph;
        u32 c2 = h1 < ph;
        u64 h2 = h1 + c;
        c2 += h2 < h1;
        hi = h2; top += c2;
    }
    // 2^128 = -2^32 (mod p)
    __device__ __forceinline__ u64 reduce() const { return gl_sub(gl_reduce128(lo, hi), (u64)top << 32); }
};

// out[r] = sum_i s[(i+r)%12]*CIRC[i] + (r==0)*8*s[0],  CIRC = [17,15,41,16,2,28,13,13,39,18,34,20]
__device__ __forceinline__ void mds_layer(u64 s[12]) {
    const u32 C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    u32 lo[12], hi[12];
#pragma unroll
    for (int i = 0; i < 12; i++) { lo[i] = (u32)s[i]; hi[i] = (u32)(s[i] >> 32); }
#pragma unroll
    for (int r = 0; r < 12; r++) {
        u64 al = 0, ah = 0;
#pragma unroll
        for (int i = 0; i < 12; i++) {
            al += (u64)lo[(i + r) % 12] * C[i];
            ah += (u64)hi[(i + r) % 12] * C[i];
        }
        if (r == 0) { al += (u64)lo[0] * 8u; ah += (u64)hi[0] * 8u; }
        // value = al + ah*2^32, al, ah < 2^42
        u64 low = al + (ah << 32);
        u64 high = (ah >> 32) + (low < al);
        // high < 2^11: reduce = low + high*eps with wrap fix
        u64 t1 = (high << 32) - high;
        u64 r0 = low + t1;
        if (r0 < t1) r0 += GL_EPS;
        s[r] = r0 >= GL_P ? r0 - GL_P : r0;
    }
}

__device__ __forceinline__ void full_rounds(u64 s[12], int first_round) {
#pragma unroll 1
    for (int r = 0; r < 4; r++) {
        const u64 *rc = PSD_RC + 12 * (first_round + r);
#pragma unroll
        for (int i = 0; i < 12; i++) s[i] = sbox7(gl_add(s[i], rc[i]));
        mds_layer(s);
    }
}

__device__ __forceinline__ void partial_rounds(u64 s[12]) {
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = gl_add(s[i], PSD_PR_FIRST[i]);
    {   // s[1..] = PR_INIT * s[1..]
        u64 t[11];
#pragma unroll 1
        for (int i = 0; i < 11; i++) {
            acc192 a; a.init();
#pragma unroll
            for (int j = 0; j < 11; j++) a.mac(PSD_PR_INIT[11 * i + j], s[1 + j]);
            t[i] = a.reduce();
        }
#pragma unroll
        for (int i = 0; i < 11; i++) s[1 + i] = t[i];
    }
#pragma unroll 1
    for (int r = 0; r < 22; r++) {
        u64 s0 = gl_add(sbox7(s[0]), PSD_PR_K[r]);
        acc192 a; a.init();
        a.mac(PSD_M00, s0);
        const u64 *vh = PSD_PR_VHAT + 11 * r, *w = PSD_PR_W + 11 * r;
#pragma unroll
        for (int j = 0; j < 11; j++) a.mac(vh[j], s[1 + j]);
#pragma unroll
        for (int j = 0; j < 11; j++) s[1 + j] = gl_mul_add(w[j], s0, s[1 + j]);
        s[0] = a.reduce();
    }
}

// state must be canonical on entry; canonical on exit
__device__ __forceinline__ void permute(u64 s[12]) {
    full_rounds(s, 0);
    partial_rounds(s);
    full_rounds(s, 26);
}

}  // namespace psd
