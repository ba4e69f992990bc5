// Poseidon-Goldilocks permutation with its linear layers on the gfx950 matrix cores (v_mfma_i32_32x32x32_i8).
// Same function as psd::core::run (poseidon.cuh; plonky2 0.2.2 hash/poseidon.rs `Poseidon::poseidon`, crate pinned at
// /root/reference/Cargo.lock:3441-3443), bit for bit; tables and layout: poseidon_mfma_tables.h.
//
// One wave = 64 permutations in "pair layout": lane (n = l&31, h = l>>5) holds words 6h..6h+5 of permutations n and 32+n.
//   * s-boxes: 12 per lane per full round, exactly as in the one-thread-per-permutation kernel (psd::core products);
//   * MDS (+ next round constants): 8 matrix-core instructions per wave and layer (two byte planes each: the MDS constants are
//     single digits, poseidon_mfma_tables.h), operands from the state registers (bytes XOR 0x80, one v_perm_b32 per two words and
//     plane pair), results recombined from eight 25-bit limb sums per word -- against 24 multiply-adds + 5 per word on the vector ALU;
//   * the 22 partial rounds: the s-box chain x_{q+1} = E[q] + sum_{j<=q} PR_D[q][j] y_j runs one permutation per lane in blocks
//     of four rounds; what a block needs from the state (E) and from all earlier blocks is one matrix product per block, the exit
//     map another, and only the in-block feedback (31 products per permutation) runs on the vector ALU (two v_permlane32_swap
//     exchanges per word in and out).
// The rare-repair protocol of psd::core<false> (sticky word, exact redo) carries over unchanged: the matrix products and the
// recombination are exact.
#pragma once
#include "poseidon.cuh"
#include "poseidon_mfma_tables.h"

typedef int psdm_v4i __attribute__((ext_vector_type(4)));
typedef int psdm_v16i __attribute__((ext_vector_type(16)));
// PSDM_EXPERIMENT_NO_MFMA (tools/psdm_test.hip only; the results are wrong on purpose): every matrix instruction replaced by one vector
// instruction that keeps its operands alive -- what the kernel would cost if its linear layers were free, i.e. the most any change to the
// matrix-core side (fewer k-steps, block-sparse operands, lower power) could ever give.
// PSDM_EXPERIMENT_A_HOT (tools/psdm_test.hip only, wrong results on purpose): every operand fragment of the partial rounds' products read
// from ONE address per lane, i.e. always a first-level cache hit -- what the 60 KiB of F / G2 operands cost by not fitting that cache
// (5-7 % of the kernel at saturation; moving them into LDS with one 768-thread workgroup per CU was measured and is slower)
#ifdef PSDM_EXPERIMENT_A_HOT
#define PSDM_AHOT(i) 0
#else
#define PSDM_AHOT(i) (i)
#endif
#ifdef PSDM_EXPERIMENT_NO_MFMA
__device__ __forceinline__ psdm_v16i psdm_no_mfma(psdm_v4i a, psdm_v4i b, psdm_v16i c) { c[0] += a[0] ^ b[0]; return c; }
#define PSDM_MFMA(a, b, c) psdm_no_mfma((a), (b), (c))
#else
#define PSDM_MFMA(a, b, c) __builtin_amdgcn_mfma_i32_32x32x32_i8((a), (b), (c), 0, 0, 0)
#endif

// LDS image per workgroup, in 16-byte fragments: the MDS operand (one fragment per lane), then the accumulator
// starts Cin_mds | Cin_F | Cin_G2 (four fragments per (.., lane half)) -- 6 KiB
#define PSDM_LDS_A 64
#define PSDM_LDS_CMDS (PSDM_LDS_A)
#define PSDM_LDS_CF (PSDM_LDS_CMDS + 8 * 4 * 2 * 4)
#define PSDM_LDS_CG2 (PSDM_LDS_CF + 6 * 2 * 4)
#define PSDM_LDS_V4_SMALL (PSDM_LDS_CG2 + 3 * 2 * 4)
// PSDM_G2_LDS (round 5 experiment, VERDICT r4 #6a): the 27 KiB of operand fragments of the partial rounds' EXIT product (A_G2) live in the
// workgroup's LDS image as well -- three 256-thread workgroups per CU hold 3 x 33 KiB -- so that the first-level cache only has to serve
// the 33 KiB of A_F (it holds 32): a ds_read_b128 per fragment instead of a global load that misses about every second time.
#ifdef PSDM_G2_LDS
#define PSDM_LDS_AG2 PSDM_LDS_V4_SMALL
#define PSDM_LDS_V4 (PSDM_LDS_V4_SMALL + 3 * 9 * 64)
#else
#define PSDM_LDS_V4 PSDM_LDS_V4_SMALL
#endif

namespace psdm {

__device__ __forceinline__ psdm_v4i ld4(const void *p);
// every workgroup: all threads call this on its `__shared__ psdm_v4i sh[PSDM_LDS_V4]`, then __syncthreads().  The MDS operand is
// used 8 x 18 times per permutation: reading it from LDS costs no vector registers (36 if held) and no L1 traffic; the accumulator
// starts (the same 64 bytes for all 32 lanes of a half: an LDS broadcast) were 132 of the ~200 16-byte global loads per permutation.
__device__ __forceinline__ void lds_fill(const PsdmTables *T, psdm_v4i *sh) {
    const psdm_v4i *a = (const psdm_v4i *)&T->A_mds[0][0], *c0 = (const psdm_v4i *)&T->Cin_mds[0][0][0][0],
                   *c1 = (const psdm_v4i *)&T->Cin_F[0][0][0], *c2 = (const psdm_v4i *)&T->Cin_G2[0][0][0];
    for (u32 i = threadIdx.x; i < PSDM_LDS_A; i += blockDim.x) sh[i] = a[i];
    for (u32 i = threadIdx.x; i < PSDM_LDS_CF - PSDM_LDS_CMDS; i += blockDim.x) sh[PSDM_LDS_CMDS + i] = c0[i];
    for (u32 i = threadIdx.x; i < PSDM_LDS_CG2 - PSDM_LDS_CF; i += blockDim.x) sh[PSDM_LDS_CF + i] = c1[i];
    for (u32 i = threadIdx.x; i < PSDM_LDS_V4_SMALL - PSDM_LDS_CG2; i += blockDim.x) sh[PSDM_LDS_CG2 + i] = c2[i];
#ifdef PSDM_G2_LDS
    const psdm_v4i *g2 = (const psdm_v4i *)&T->A_G2[0][0][0][0];
    for (u32 i = threadIdx.x; i < 3 * 9 * 64; i += blockDim.x) sh[PSDM_LDS_AG2 + i] = g2[i];
#endif
}
// pointer into the workgroup's LDS image, typed as such: a generic pointer makes hipcc read LDS with flat_load (aperture check,
// both memory counters) instead of ds_read_b128 with an immediate offset
typedef __attribute__((address_space(3))) const psdm_v4i *psdm_lds_ptr;
__device__ __forceinline__ psdm_v16i ld16l(psdm_lds_ptr p) {
    psdm_v16i v;
    const psdm_v4i a = p[0], b = p[1], c = p[2], d = p[3];
    v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
    v[8] = c[0]; v[9] = c[1]; v[10] = c[2]; v[11] = c[3]; v[12] = d[0]; v[13] = d[1]; v[14] = d[2]; v[15] = d[3];
    return v;
}

__device__ __forceinline__ psdm_v4i ld4(const void *p) { return *(const psdm_v4i *)p; }
__device__ __forceinline__ psdm_v4i ld4(const __attribute__((address_space(1))) void *p) { return *(const __attribute__((address_space(1))) psdm_v4i *)p; }
__device__ __forceinline__ psdm_v16i ld16(const int32_t *p) {
    psdm_v16i v;
    psdm_v4i a = ld4(p), b = ld4(p + 4), c = ld4(p + 8), d = ld4(p + 12);
    v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
    v[8] = c[0]; v[9] = c[1]; v[10] = c[2]; v[11] = c[3]; v[12] = d[0]; v[13] = d[1]; v[14] = d[2]; v[15] = d[3];
    return v;
}
// sixteen operand bytes of one k-step: two state words, every byte XOR 0x80 (u - 128 as a signed byte)
__device__ __forceinline__ psdm_v4i bfrag(u64 w0, u64 w1) {
    psdm_v4i b;
    b[0] = (int)((u32)w0 ^ 0x80808080u); b[1] = (int)((u32)(w0 >> 32) ^ 0x80808080u);
    b[2] = (int)((u32)w1 ^ 0x80808080u); b[3] = (int)((u32)(w1 >> 32) ^ 0x80808080u);
    return b;
}
// sum_a u_a 2^(8a) as a weak residue, every limb sum u_a below 2^24 - 2^16 (poseidon_mfma_tables.h, OFF_LOG).
// Limbs are paired with 32-bit shift-adds (u_a + 2^8 u_{a+1} < 2^32), pairs joined by one v_mad_u64_u32 (x 2^16): AL, AH < 2^49.
// AL + AH 2^32 = AL + AH.hi 2^64 + AH.lo 2^32  ->  (AL + AH.hi * eps) cannot overflow (both terms < 2^49); adding AH.lo into the high
// word may wrap once: + eps.  10 vector instructions per word (was 13).  The multipliers are handed over as opaque wave-uniform
// values (struct rk) so that hipcc emits one v_mad_u64_u32 per product instead of zero-extend + shift + add.  The first readers of
// a matrix-core result stay compiler-generated instructions on purpose: they must be visible to the hazard recognizer (an
// inline-asm reader is not).
struct rk {
    u32 k16, keps;
    __device__ __forceinline__ void init() {
        k16 = 1u << 16; keps = 0xFFFFFFFFu;
        asm("" : "+s"(k16), "+s"(keps));
    }
};
__device__ __forceinline__ u64 recombine(const rk &K, u32 u0, u32 u1, u32 u2, u32 u3, u32 u4, u32 u5, u32 u6, u32 u7) {
    const u32 t0 = (u1 << 8) + u0, t1 = (u3 << 8) + u2, t2 = (u5 << 8) + u4, t3 = (u7 << 8) + u6;     // v_lshl_add_u32
    const u64 AL = (u64)t1 * K.k16 + t0, AH = (u64)t3 * K.k16 + t2;
    const u64 r = (u64)(u32)(AH >> 32) * K.keps + AL;
    // r + AH.lo 2^32, one wrap repaired by + eps: add into the high word, carry -> mask -> x*1 + acc (the operands are results of
    // ordinary vector instructions, not of the matrix core, so this may be inline asm).  s_nop 1: the two wait states between a vector
    // write of VCC and its vector read (the unpadded pair was exact in every run, but unpadded VCC pairs elsewhere were not --
    // DESIGN.md section 10 -- and the pad costs nothing measurable in proofs/s)
    u32 rl = (u32)r, rh = (u32)(r >> 32), ahl = (u32)AH, m; u64 w;
    asm("v_add_co_u32 %[rh], vcc, %[rh], %[a]\n\t"
        "s_nop 1\n\t"
        "v_cndmask_b32_e64 %[m], 0, -1, vcc\n\t"
        : [rh] "+v"(rh), [m] "=&v"(m) : [a] "v"(ahl) : "vcc");
    w = ((u64)rh << 32) | rl;
    return gl_madd1(m, w);
}
// the same from the four limb PAIRS t_P = u_2P + 2^8 u_(2P+1) (each below 2^32): what an MDS layer keeps of a tile's accumulator
__device__ __forceinline__ u64 recombine_pairs(const rk &K, u32 t0, u32 t1, u32 t2, u32 t3) {
    const u64 AL = (u64)t1 * K.k16 + t0, AH = (u64)t3 * K.k16 + t2;
    const u64 r = (u64)(u32)(AH >> 32) * K.keps + AL;
    u32 rl = (u32)r, rh = (u32)(r >> 32), ahl = (u32)AH, m; u64 w;
    asm("v_add_co_u32 %[rh], vcc, %[rh], %[a]\n\t"
        "s_nop 1\n\t"
        "v_cndmask_b32_e64 %[m], 0, -1, vcc\n\t"
        : [rh] "+v"(rh), [m] "=&v"(m) : [a] "v"(ahl) : "vcc");
    w = ((u64)rh << 32) | rl;
    return gl_madd1(m, w);
}
__device__ __forceinline__ void recombine2(const rk &K, const psdm_v16i &acc, u64 &o0, u64 &o1) {
    o0 = recombine(K, acc[0], acc[1], acc[2], acc[3], acc[4], acc[5], acc[6], acc[7]);
    o1 = recombine(K, acc[8], acc[9], acc[10], acc[11], acc[12], acc[13], acc[14], acc[15]);
}
// vdst lanes 32..63 <-> src lanes 0..31.  v_permlane32_swap needs two wait states after a vector write of either operand; hipcc
// pads only writes it can see, and these operands come out of inline-asm blocks (gl_mad_eps), hence the explicit pad.
__device__ __forceinline__ void swap32(u64 &vdst, u64 &src) {
    asm volatile("s_nop 1" : "+v"(vdst), "+v"(src));
    auto lo = __builtin_amdgcn_permlane32_swap((u32)vdst, (u32)src, false, false);
    auto hi = __builtin_amdgcn_permlane32_swap((u32)(vdst >> 32), (u32)(src >> 32), false, false);
    vdst = ((u64)hi[0] << 32) | lo[0]; src = ((u64)hi[1] << 32) | lo[1];
}

// NC = permutations per lane pair: 2 = pair layout (64 permutations per wave), 1 = 32 permutations per wave (lane (n, h) holds words
// 6h..6h+5 of permutation n only) -- half the vector work per wave, for launches too small to give every SIMD a pair-layout wave.
template <bool EXACT, int NC = 2> struct perm {
    psd::core<EXACT> pc;
    const __attribute__((address_space(1))) PsdmTables *T;     // typed as global memory: global_load with a scalar base, not flat_load
    u32 lane;
    rk K;
    psdm_lds_ptr Alds;         // the workgroup's LDS image (A_mds, then the accumulator starts), see lds_fill

    __device__ __forceinline__ void init(const PsdmTables *tab, u32 lane_, const psdm_v4i *lds) { T = (const __attribute__((address_space(1))) PsdmTables *)tab; lane = lane_; pc.fl = 0; K.init(); Alds = (psdm_lds_ptr)lds; }

    // s <- MDS * s + constants of layer `layer`.  Two byte planes (= one halfword of every state word) per instruction: operand bytes
    // j < 12 of lane (n, h) are the bytes of halfword P of its six words (XOR 0x80), result registers 2i, 2i+1 the limbs 2P, 2P+1 of its
    // word i (poseidon_mfma_tables.h); of each accumulator only the six limb pairs are kept until the four tiles are done.
    __device__ __forceinline__ void mds(u64 (&s)[NC][6], int layer) {
        u32 x[NC][2][6];
#pragma unroll
        for (int c = 0; c < NC; c++)
#pragma unroll
            for (int i = 0; i < 6; i++) { x[c][0][i] = (u32)s[c][i] ^ 0x80808080u; x[c][1][i] = (u32)(s[c][i] >> 32) ^ 0x80808080u; }
        const psdm_v4i a = Alds[lane];
        u32 t[NC][4][6];
#pragma unroll
        for (int P = 0; P < 4; P++) {
            const psdm_v16i cin = ld16l(Alds + PSDM_LDS_CMDS + ((layer * 4 + P) * 2 + (lane >> 5)) * 4);
            const u32 sel = (P & 1) ? 0x07060302u : 0x05040100u;           // v_perm_b32: halfword (P & 1) of two dwords
#pragma unroll
            for (int c = 0; c < NC; c++) {
                psdm_v4i b;
                b[0] = (int)__builtin_amdgcn_perm(x[c][P >> 1][1], x[c][P >> 1][0], sel);
                b[1] = (int)__builtin_amdgcn_perm(x[c][P >> 1][3], x[c][P >> 1][2], sel);
                b[2] = (int)__builtin_amdgcn_perm(x[c][P >> 1][5], x[c][P >> 1][4], sel);
                b[3] = 0;
                psdm_v16i acc = PSDM_MFMA(a, b, cin);
#pragma unroll
                for (int i = 0; i < 6; i++) t[c][P][i] = ((u32)acc[2 * i + 1] << 8) + (u32)acc[2 * i];
                // Registers 12..15 of the result are padding rows and nothing reads them -- which makes them free registers for the compiler
                // from the instruction AFTER the matrix instruction on, while the hardware writes all sixteen when it finishes, 64 cycles
                // later.  hipcc pads its own instructions against that (write after write on an in-flight result), but the vector
                // instructions inside the inline-asm blocks of the s-boxes are invisible to its hazard recognizer: a value they had just
                // put into one of the four was overwritten (the second permutation of every pair came out wrong whenever register
                // pressure made the allocator take those registers).  The whole result stays live until its limbs have been read: the
                // empty statement READS AND WRITES all sixteen registers, so it cannot move above the limb reads before it (they would
                // see its "new" value) and nothing else can be given registers 12..15 before it; the first limb read is a compiler-made
                // instruction, which hipcc holds back until the matrix instruction has written its result (an input-only
                // operand ordered nothing).  tools/asm_clobber_audit.py --code checks the built code objects for the pattern.
#if defined(PSDM_RESULT_LIVENESS_INPUT_ONLY)   // (tools/psdm_test.hip only: round 4's form, for an A/B of what the stronger statement costs)
                asm volatile("" : : "v"(acc));
#elif !defined(PSDM_NO_RESULT_LIVENESS)       // (defined by the negative control of tests/test_abi_cpu.py only: the round-4 defect, for the code-object audit to find)
                asm volatile("" : "+v"(acc));
#endif
            }
        }
#pragma unroll
        for (int c = 0; c < NC; c++)
#pragma unroll
            for (int i = 0; i < 6; i++) s[c][i] = recombine_pairs(K, t[c][0][i], t[c][1][i], t[c][2][i], t[c][3][i]);
    }
    __device__ __forceinline__ void full_rounds(u64 (&s)[NC][6], int layer0) {
#pragma unroll 1
        for (int r = 0; r < 4; r++) {
#pragma unroll
            for (int c = 0; c < NC; c++)
#pragma unroll
                for (int i = 0; i < 6; i++) s[c][i] = pc.sbox(s[c][i]);
            mds(s, layer0 + r);
        }
    }

    // The 22 partial rounds.  Chain:  y_q = x_q^7 + K[q];  x_{q+1} = E[q] + 25 y_q + sum_{j<q} PR_D[q][j] y_j  (x_0 = state word 0).
    // Rounds run in blocks of four.  For the rows of block b everything that is known when the block starts -- E (linear in the
    // state) and the feedback of all earlier blocks -- is ONE matrix product F = A_F[b] * (state | y_0 .. y_{4b-1}) whose K grows by
    // one k-step per finished block; the vector ALU only adds the in-block feedback (0 + 1 + 2 + 3 products per block instead of
    // q products in round q: 31 instead of 210 per permutation).  The chain itself runs one permutation per lane (lane (n, h) runs
    // permutation (c = h, n)); matrix-core operands and results are in pair layout, two v_permlane32_swap exchanges per word.
    __device__ __forceinline__ void partial_rounds(u64 (&s)[NC][6]) {
        psdm_v4i bs[NC][3];                // operand bytes of the state; the state registers themselves are free from here on
#pragma unroll
        for (int c = 0; c < NC; c++)
#pragma unroll
            for (int k = 0; k < 3; k++) bs[c][k] = bfrag(s[c][2 * k], s[c][2 * k + 1]);
        psdm_v4i by[NC][6];                // operand bytes of the finished blocks' s-box outputs
        u64 x;
        // (NC = 1: both lane halves run the chain of permutation n -- a copy, not garbage, so that the sticky word stays as rare)
        { u64 d = s[0][0], w = s[NC - 1][0]; swap32(d, w); x = d; }
        const u64 *Dm = PSD_PR_D, *Kc = PSD_PR_K;
#pragma unroll
        for (int b = 0; b < 6; b++) {
            // ---- F rows 4b .. 4b+3 for both column tiles, then to one permutation per lane
            u64 F[4];
            {
                const psdm_v16i cin = ld16l(Alds + PSDM_LDS_CF + (b * 2 + (lane >> 5)) * 4);
                psdm_v16i acc0 = cin, acc1 = cin;
#pragma unroll
                for (int k = 0; k < 3 + b; k++) {
                    const psdm_v4i a = ld4(&T->A_F[PSDM_AHOT(b)][PSDM_AHOT(k)][lane][0]);
                    acc0 = PSDM_MFMA(a, k < 3 ? bs[0][k] : by[0][k - 3], acc0);
                    if (NC == 2) acc1 = PSDM_MFMA(a, k < 3 ? bs[NC - 1][k] : by[NC - 1][k - 3], acc1);
                }
                u64 f0[2], f1[2];
                recombine2(K, acc0, f0[0], f0[1]);
                if (NC == 2) recombine2(K, acc1, f1[0], f1[1]);
                else { f1[0] = f0[0]; f1[1] = f0[1]; }
#pragma unroll
                for (int i = 0; i < 2; i++) { u64 yy = f0[i], xx = f1[i]; swap32(yy, xx); F[i] = yy; F[2 + i] = xx; }
            }
            // ---- four rounds of the chain (two in the last block)
            u64 y[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int q = 4 * b + i;
                if (q > 21) { y[i] = 0; continue; }
                y[i] = psd::wadd_c(pc.sbox(x), Kc[q]);
                if (q == 21) continue;
                // F[q] + 25 y_q: two 38-bit column sums, folded like an MDS row; then one multiply-add per in-block term
                u64 al = (u64)(u32)F[i] + (u64)(u32)y[i] * (u32)PSD_M00, ah = (F[i] >> 32) + (y[i] >> 32) * (u32)PSD_M00;
                u32 th, cy;
                th = __builtin_addc((u32)(al >> 32), (u32)ah, 0u, &cy);
                u32 high = (u32)(ah >> 32) + cy;
                x = gl_mad_eps(high, ((u64)th << 32) | (u32)al);
#pragma unroll
                for (int j = 0; j < i; j++) x = pc.mul_add(Dm[22 * q + 4 * b + j], y[j], x);
                // the sticky word is brought up to date once per round: left alone, hipcc defers the ORs of the borrow masks across the
                // unrolled chain and parks ~150 pending masks in VGPR lanes (300 v_writelane / v_readlane per permutation)
                asm volatile("" : "+s"(pc.fl));
            }
            // ---- the block's s-box outputs to pair layout, as operand bytes
            {
                u64 v0 = y[0], s0 = y[2], v1 = y[1], s1 = y[3];
                swap32(v0, s0); swap32(v1, s1);
                by[0][b] = bfrag(v0, v1);
                if (NC == 2) by[NC - 1][b] = bfrag(s0, s1);
            }
        }
        // ---- exit map G2 (K = 12 state words + 6 blocks of y), + RC[26]
#pragma unroll
        for (int t = 0; t < 3; t++) {
            const psdm_v16i cin = ld16l(Alds + PSDM_LDS_CG2 + (t * 2 + (lane >> 5)) * 4);
            psdm_v16i acc0 = cin, acc1 = cin;
#pragma unroll
            for (int k = 0; k < 9; k++) {
#ifdef PSDM_G2_LDS
                const psdm_v4i a = Alds[PSDM_LDS_AG2 + (t * 9 + k) * 64 + lane];
#else
                const psdm_v4i a = ld4(&T->A_G2[PSDM_AHOT(t)][PSDM_AHOT(k)][lane][0]);
#endif
                acc0 = PSDM_MFMA(a, k < 3 ? bs[0][k] : by[0][k - 3], acc0);
                if (NC == 2) acc1 = PSDM_MFMA(a, k < 3 ? bs[NC - 1][k] : by[NC - 1][k - 3], acc1);
            }
            recombine2(K, acc0, s[0][2 * t], s[0][2 * t + 1]);
            if (NC == 2) recombine2(K, acc1, s[NC - 1][2 * t], s[NC - 1][2 * t + 1]);
        }
    }

    // any 64-bit representatives in, canonical out
    __device__ __forceinline__ void run(u64 (&s)[NC][6]) {
        // the table pointers are made opaque once per permutation: otherwise hipcc hoists the (loop-invariant) operand and
        // accumulator-start loads of every layer out of the caller's loop and keeps ~100 registers alive for them
        asm volatile("" : "+s"(T), "+v"(Alds));
        {
            const auto *rc0 = T->RC_first + 6 * (lane >> 5);
#pragma unroll
            for (int i = 0; i < 6; i++) { const u64 k = rc0[i]; s[0][i] = psd::wadd_c(s[0][i], k); if (NC == 2) s[NC - 1][i] = psd::wadd_c(s[NC - 1][i], k); }
        }
        full_rounds(s, 0);
        partial_rounds(s);
        full_rounds(s, 4);
#pragma unroll
        for (int c = 0; c < NC; c++)
#pragma unroll
            for (int i = 0; i < 6; i++) s[c][i] = gl_canon(s[c][i]);
    }
};

}  // namespace psdm
