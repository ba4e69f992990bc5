// Lane-cooperative Poseidon permutation: one permutation per 16-lane group (4 per wave), state element l in lane l
// (l < 12; lanes 12..15 of a group idle along).  Same function as psd::permute (poseidon.cuh, plonky2 0.2.2
// hash/poseidon.rs `Poseidon::poseidon`) -- about 7x lower latency per permutation at about 1/3 of the
// throughput, so it is used where the work is a serial chain or too small to fill the chip: the Fiat-Shamir
// transcript and the top levels of every Merkle tree.  Cross-lane traffic uses wave shuffles (ds_bpermute), no LDS.
#pragma once
#include "poseidon.cuh"

namespace psdc {

__device__ __forceinline__ u64 shfl64(u64 v, int src) {
    u32 lo = (u32)v, hi = (u32)(v >> 32);
    lo = __shfl(lo, src, 64); hi = __shfl(hi, src, 64);
    return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ u64 shfl_xor64(u64 v, int mask) {
    u32 lo = (u32)v, hi = (u32)(v >> 32);
    lo = __shfl_xor(lo, mask, 64); hi = __shfl_xor(hi, mask, 64);
    return ((u64)hi << 32) | lo;
}

// rotate within each row of 16 lanes by N (DPP row_ror: VALU latency instead of the LDS-path latency of ds_bpermute)
template <int N> __device__ __forceinline__ u64 ror16(u64 v) {
    int lo = (int)(u32)v, hi = (int)(u32)(v >> 32);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x120 + N, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x120 + N, 0xF, 0xF, false);
    return ((u64)(u32)hi << 32) | (u32)lo;
}

// s: this lane's state element (canonical).  l = lane & 15, gb = lane & ~15.  Lanes with l >= 12 pass s = 0 and ignore the result.
__device__ __forceinline__ u64 permute16(u64 s, const int l, const int gb) {
    const int lc = l < 12 ? l : 0;                      // clamp table indices of idle lanes
    const u32 C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
#pragma unroll 1
    for (int half = 0; half < 2; half++) {
        if (half == 1) {
            // ---- 22 partial rounds, sparse form (tools/gen_constants.py)
            s = gl_add(s, PSD_PR_FIRST[lc]);
            {   // s[1..] = PR_INIT * s[1..]
                psd::acc192 a; a.init();
                const int row = lc >= 1 ? lc - 1 : 0;
#pragma unroll 1
                for (int j = 0; j < 11; j++) {
                    u64 sj = shfl64(s, gb + 1 + j);
                    a.mac(PSD_PR_INIT[11 * row + j], sj);
                }
                u64 t = gl_canon(a.reduce());
                if (l >= 1) s = t;
            }
#pragma unroll 1
            for (int r = 0; r < 22; r++) {
                u64 s0 = 0;
                if (l == 0) s0 = gl_add(psd::sbox7(s), PSD_PR_K[r]);
                // broadcast lane 0 of the group: s0 is zero in the other lanes, so an OR all-reduce over the row does it
                s0 |= ror16<8>(s0); s0 |= ror16<4>(s0); s0 |= ror16<2>(s0); s0 |= ror16<1>(s0);
                const int j = lc >= 1 ? lc - 1 : 0;
                u64 t = l == 0 ? gl_mul(PSD_M00, s0) : gl_mul(PSD_PR_VHAT[11 * r + j], s);
                if (l >= 12) t = 0;
                // all-reduce over the 16 lanes of the group
                t = gl_add(t, ror16<8>(t));
                t = gl_add(t, ror16<4>(t));
                t = gl_add(t, ror16<2>(t));
                t = gl_add(t, ror16<1>(t));
                u64 upd = gl_mul_add(PSD_PR_W[11 * r + j], s0, s);
                s = l == 0 ? t : upd;
            }
        }
        // ---- 4 full rounds
#pragma unroll 1
        for (int r = 0; r < 4; r++) {
            const int round = (half == 0 ? 0 : 26) + r;
            s = psd::sbox7(gl_add(s, PSD_RC[12 * round + lc]));
            if (l >= 12) s = 0;
            // MDS: out_l = sum_i s_{(l+i)%12} * C[i] + (l == 0) * 8 * s_0
            u32 lo = (u32)s, hi = (u32)(s >> 32);
            u64 al = 0, ah = 0;
#pragma unroll
            for (int i = 0; i < 12; i++) {
                int src = l + i; if (src >= 12) src -= 12;
                if (l >= 12) src = 0;
                u32 xl = __shfl(lo, gb + src, 64), xh = __shfl(hi, gb + src, 64);
                al += (u64)xl * C[i]; ah += (u64)xh * C[i];
            }
            if (l == 0) { al += (u64)lo * 8u; ah += (u64)hi * 8u; }
            u64 low = al + (ah << 32);
            u64 high = (ah >> 32) + (low < al);
            u64 t1 = (high << 32) - high;
            u64 r0 = low + t1;
            if (r0 < t1) r0 += GL_EPS;
            s = r0 >= GL_P ? r0 - GL_P : r0;
        }
    }
    return s;
}

}  // namespace psdc
