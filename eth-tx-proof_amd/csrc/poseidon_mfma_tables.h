// Host-side builder of the constant tables of the matrix-core Poseidon permutation (poseidon_mfma.cuh).
//
// The linear layers of Poseidon-Goldilocks (plonky2 0.2.2 hash/poseidon.rs: mds_layer, partial_first_constant_layer,
// mds_partial_layer_init, mds_partial_layer_fast; crate pinned at /root/reference/Cargo.lock:3441-3443) are
// products of a CONSTANT matrix over F_p with the state vectors of many independent permutations: a dense contraction.  It is
// done exactly on the i8 matrix cores by splitting both sides into bytes:
//     out[W] = sum_wi M[W][wi] * x[wi],   x[wi] = sum_b u_b 2^(8b)   (u_b = bytes of the 64-bit word)
//            = sum_a 2^(8a) * sum_(wi,b) digit_a( M[W][wi] * 2^(8b) mod p ) * u_b
// where digit_a(.) are the eight signed base-256 digits (-128..127) of the representative of least absolute value.  The
// matrix-core operands are signed bytes, so u_b enters as s_b = u_b - 128 (one XOR with 0x80 per byte) and the accumulator
// starts at 128 * rowsum, plus an offset that keeps it non-negative, plus the matching byte of the additive constant of the
// layer (round constants), so that constant layers cost nothing.
//
// Lane maps of v_mfma_i32_32x32x32_i8 (checked with exact integer data by tools/mfma_probe.hip): A: lane l holds
// A[row l&31][(l>>5, j)], j = 0..15; B: lane l holds B[(l>>5, j)][col l&31]; D: register r of lane l is
// D[row (r&3) + 8(r>>2) + 4(l>>5)][col l&31].
//
// State layout ("pair layout"): a wave holds 64 permutations P = 32c + n; lane l = (n = l&31, h = l>>5) holds words
// 6h .. 6h+5 of permutations (c = 0, n) and (c = 1, n).  K index (k-step ks, lane half h, byte j) = byte j&7 of word
// 6h + 2ks + (j>>3); output row (tile T, r = 4t + i on lane half h') = limb 4(t&1) + i of output slot 2T + (t>>1) of that
// half: both operands of every layer come straight from, and land straight in, the state registers.
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "gl64.cuh"

// The MDS layers are a special case: their constants are below 2^7, so M * 2^(8b) has ONE non-zero digit, at position b -- output limb a
// depends on byte a of the twelve state words and on nothing else, the 96 x 96 operand of the general scheme is block-diagonal by byte
// plane and seven eighths of its products multiply zeros.  An MDS layer therefore runs TWO planes per instruction (round 4: the matrix
// instructions are NOT hidden behind the vector ALU in these kernels -- at saturation each cost about 68 cycles of the SIMD, twice its 32
// cycles of pipe; removing all of them makes the leaf hash 28 % faster, DESIGN.md section 4 -- so instruction count on this side matters):
//   tile P = 0..3 covers byte planes 2P, 2P+1 = halfword P of every word;  K index (lane half h, byte j < 12) = byte (j & 1) of halfword
//   P of word 6h + (j >> 1) (j = 12..15: padding);  output row (lane half h', register r < 12) = limb 2P + (r & 1) of word 6h' + (r >> 1).
// One operand fragment serves all four tiles and all eight layers; 8 instructions per layer and wave instead of 18.
struct PsdmTables {
    int8_t A_mds[64][16];
    int32_t Cin_mds[8][4][2][16];      // [layer][plane pair][lane half][register]; layer L: next-layer constants RC[L+1] (L<3), PR_FIRST (3), RC[27+L-4] (4..6), none (7)
    // s-box chain of the partial rounds in blocks of four rounds: tile b gives, for rows q = 4b .. 4b+3 (lane half h', slot i:
    // q = 4b + 2h' + i), F[q] = E[q] + sum_{j < 4b} PR_D[q][j] y_j with E[q] = <vhat_q, PR_INIT * s[1..]>.
    // K-steps: 0..2 = the 12 state words, 3 + b' = the four s-box outputs y_{4b'} .. y_{4b'+3} of block b' < b (lane half h, word
    // w: y_{4b' + 2h + w}) -- everything already known when block b starts is a matrix product; only the in-block feedback
    // (six products per block) stays on the vector ALU.
    int8_t A_F[6][8][64][16];
    int32_t Cin_F[6][2][16];
    int8_t A_G2[3][9][64][16];         // exit of the partial rounds: K = 12 state words + 22 (24 slots, blocks of four) s-box outputs
    int32_t Cin_G2[3][2][16];          // + RC[26]
    uint64_t RC_first[12];
};

namespace psdm_host {
#undef PSD_CONST
#define PSD_CONST static const
#include "poseidon_constants.inc"

// The accumulators of K <= 288 signed-byte products lie within +-288 * 2^14 = +-2^22.17; started at 2^23 (plus the constant's byte)
// every limb sum u stays in (2^21.8, 2^23.66): below 2^24 - 2^16, which is what lets the recombination pair limbs with 32-bit
// shift-adds (u_a + 2^8 u_{a+1} < 2^32, poseidon_mfma.cuh recombine).
static const uint32_t OFF_LOG = 23;

// eight signed base-256 digits (-128..127) of a representative of c (mod p).  Eight such digits cover
// [-128 (2^64-1)/255, 127 (2^64-1)/255]: c itself up to 0x7F7F7F7F7F7F7F7F, c - p above (c - p > -128 (2^64-1)/255 there).
static inline void signed_digits(u64 c, int8_t d[8]) {
    __int128 sv = (c <= 0x7F7F7F7F7F7F7F7FULL) ? (__int128)c : (__int128)c - (__int128)GL_P;
    for (int a = 0; a < 8; a++) {
        int low = (int)(sv & 0xFF);
        if (low >= 128) low -= 256;
        d[a] = (int8_t)low;
        sv = (sv - low) >> 8;                           // exact: sv - low is a multiple of 256
    }
    if (sv != 0) abort();
}

struct RowSpec { bool live; int word; int limb; };
// row (tile T, index rho = 8t + 4h' + i)
static inline void row_of(int T, int rho, int &hp, int &slot, int &limb) {
    int t = rho >> 3, i = rho & 3; hp = (rho >> 2) & 1;
    slot = 2 * T + (t >> 1); limb = 4 * (t & 1) + i;
}
// lane l, register r of the accumulator of tile T  ->  row index in the tile
static inline int rho_of(int l, int r) { return (r & 3) + 8 * (r >> 2) + 4 * (l >> 5); }

// Generic filler: `coef(out_id, kword)` gives the matrix entry (mod p) of output out_id (what `out_of(hp, slot)` returns, or -1
// for a dead row) against K-word kword (what `kword_of(ks, h, wsel)` returns, or -1 for padding); addc(out_id) the additive
// constant.
template <class OutOf, class KWordOf, class Coef, class AddC>
static void fill(int n_tiles, int n_ksteps, int8_t *A /*[T][ks][64][16]*/, int32_t *Cin /*[T][2][16]*/, OutOf out_of, KWordOf kword_of,
                 Coef coef, AddC addc) {
    const u64 ONES = 0x0101010101010101ULL % GL_P;
    const u64 offc = gl_mul(((u64)1 << OFF_LOG) % GL_P, ONES);
    for (int T = 0; T < n_tiles; T++) {
        long rowsum[32];
        for (int m = 0; m < 32; m++) rowsum[m] = 0;
        for (int ks = 0; ks < n_ksteps; ks++)
            for (int l = 0; l < 64; l++) {
                int m = l & 31, h = l >> 5, hp, slot, limb;
                row_of(T, m, hp, slot, limb);
                int out_id = out_of(hp, slot);
                int8_t *dst = A + (((size_t)T * n_ksteps + ks) * 64 + l) * 16;
                for (int j = 0; j < 16; j++) {
                    int kw = kword_of(ks, h, j >> 3), b = j & 7;
                    int8_t v = 0;
                    if (out_id >= 0 && kw >= 0) {
                        u64 c = gl_mul(coef(out_id, kw), gl_pow(2, 8 * b));
                        int8_t d[8]; signed_digits(c, d);
                        v = d[limb];
                    }
                    dst[j] = v;
                    rowsum[m] += v;
                }
            }
        for (int hh = 0; hh < 2; hh++)                      // the accumulator start depends on the row only: one copy per lane half
            for (int r = 0; r < 16; r++) {
                int rho = rho_of(32 * hh, r), hp, slot, limb;
                row_of(T, rho, hp, slot, limb);
                int out_id = out_of(hp, slot);
                int32_t v = (int32_t)1 << OFF_LOG;
                if (out_id >= 0) {
                    u64 kk = gl_sub(addc(out_id) % GL_P, offc);
                    v += (int32_t)(128 * rowsum[rho]) + (int32_t)((kk >> (8 * limb)) & 0xFF);
                }
                Cin[((size_t)T * 2 + hh) * 16 + r] = v;
            }
    }
}

static inline u64 mds_entry(int W, int wi) {
    static const u64 C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    return C[((wi - W) % 12 + 12) % 12] + ((W == 0 && wi == 0) ? 8 : 0);
}

// the constants that MDS layer L adds: those of the layer that follows it
static inline const u64 *mds_next_constants(int L) {
    if (L < 3) return PSD_RC + 12 * (L + 1);
    if (L == 3) return PSD_PR_FIRST;
    if (L < 7) return PSD_RC + 12 * (26 + (L - 4) + 1);
    return nullptr;
}
// MDS layers, two byte planes per instruction (see PsdmTables)
static void fill_mds(PsdmTables &t) {
    const u64 ONES = 0x0101010101010101ULL % GL_P;
    const u64 offc = gl_mul(((u64)1 << OFF_LOG) % GL_P, ONES);
    long rowsum[32];
    for (int m = 0; m < 32; m++) rowsum[m] = 0;
    for (int l = 0; l < 64; l++) {
        const int m = l & 31, h = l >> 5;
        const int hp = (m >> 2) & 1, u = 4 * (m >> 3) + (m & 3);           // row m = register u of lane half hp (rho_of)
        for (int j = 0; j < 16; j++) {
            int8_t v = 0;
            if (u < 12 && j < 12 && (u & 1) == (j & 1)) {
                const u64 c = mds_entry(6 * hp + (u >> 1), 6 * h + (j >> 1));
                if (c >= 128) abort();                                      // one digit, as a signed byte
                v = (int8_t)c;
            }
            t.A_mds[l][j] = v;
            rowsum[m] += v;
        }
    }
    for (int L = 0; L < 8; L++) {
        const u64 *nxt = mds_next_constants(L);
        for (int P = 0; P < 4; P++)
            for (int hh = 0; hh < 2; hh++)
                for (int r = 0; r < 16; r++) {
                    int32_t v = (int32_t)1 << OFF_LOG;
                    if (r < 12) {
                        const int W = 6 * hh + (r >> 1), limb = 2 * P + (r & 1);
                        const u64 kk = gl_sub((nxt ? nxt[W] : (u64)0) % GL_P, offc);
                        v += (int32_t)(128 * rowsum[rho_of(32 * hh, r)]) + (int32_t)((kk >> (8 * limb)) & 0xFF);
                    }
                    t.Cin_mds[L][P][hh][r] = v;
                }
    }
}

static void build(PsdmTables &t) {
    memset(&t, 0, sizeof t);
    auto word_out = [](int hp, int slot) { return slot < 6 ? 6 * hp + slot : -1; };
    // ---- MDS (+ the constants of the layer that follows)
    fill_mds(t);
    // ---- G1: E[e] = sum_j (sum_i vhat_e[i] * PR_INIT[11 i + j]) * s[1 + j]
    static u64 G1[22][12];
    for (int e = 0; e < 22; e++) {
        G1[e][0] = 0;
        for (int j = 0; j < 11; j++) {
            u64 acc = 0;
            for (int i = 0; i < 11; i++) acc = gl_add(acc, gl_mul(PSD_PR_VHAT[11 * e + i], PSD_PR_INIT[11 * i + j]));
            G1[e][1 + j] = acc;
        }
    }
    // K words: 0..11 = state at the entry of the partial rounds, 12 + q = y_q (k-step 3 + b', lane half h, word w: q = 4b' + 2h + w)
    auto chain_kword = [](int ks, int h, int wsel) {
        if (ks < 3) return 6 * h + 2 * ks + wsel;
        int q = 4 * (ks - 3) + 2 * h + wsel;
        return q < 22 ? 12 + q : -1;
    };
    for (int b = 0; b < 6; b++)
        fill(1, 3 + b, &t.A_F[b][0][0][0], &t.Cin_F[b][0][0],
             [b](int hp, int slot) { int q = 4 * b + 2 * hp + slot; return (slot < 2 && q < 21) ? q : -1; }, chain_kword,
             [&](int q, int kw) -> u64 { return kw < 12 ? G1[q][kw] : PSD_PR_D[22 * q + (kw - 12)]; }, [](int) { return (u64)0; });
    // ---- G2: new s[0] = E[21] + sum_q PR_D[21][q] y_q;  new s[1+i] = (PR_INIT s[1..])[i] + sum_q PR_W[q][i] y_q;  + RC[26]
    fill(3, 9, &t.A_G2[0][0][0][0], &t.Cin_G2[0][0][0], word_out, chain_kword,
         [&](int W, int kw) -> u64 {
             if (kw < 12) {
                 if (W == 0) return G1[21][kw];
                 return kw == 0 ? 0 : PSD_PR_INIT[11 * (W - 1) + (kw - 1)];
             }
             int q = kw - 12;
             return W == 0 ? PSD_PR_D[22 * 21 + q] : PSD_PR_W[11 * q + (W - 1)];
         },
         [](int W) { return PSD_RC[12 * 26 + W]; });
    for (int i = 0; i < 12; i++) t.RC_first[i] = PSD_RC[i];
}

}  // namespace psdm_host
