// Host-only check of the matrix-core Poseidon tables (poseidon_mfma_tables.h): emulates v_mfma_i32_32x32x32_i8 with plain integer
// loops (operand and result lane maps as documented there) and verifies, for random states, that
//   * the MDS tables reproduce  MDS * s + next-layer constants,
//   * F (per block of four rounds: E + the feedback of earlier blocks) and G2 reproduce the partial rounds' chain inputs and exit map
// against direct modular arithmetic on the constants of poseidon_constants.inc.  No GPU needed.
// build: hipcc -O2 -std=c++17 tools/psdm_tables_check.hip -o tools/bin/psdm_tables_check   (host code only)
#include <stdio.h>
#include <stdlib.h>
#include "../eth-tx-proof_amd/csrc/poseidon_mfma_tables.h"
using namespace psdm_host;

static u64 splitmix(u64 &z) { z += 0x9E3779B97F4A7C15ULL; u64 x = z; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL; x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL; return x ^ (x >> 31); }

// one column (permutation) through n_tiles x n_ksteps emulated instructions; kw[ks][h][wsel] = the 64-bit word lane half h feeds
// into k-step ks; returns out[hp][slot] recombined mod p
static void emulate(int n_tiles, int n_ksteps, const int8_t *A, const int32_t *Cin, const u64 (*kw)[2][2], u64 out[2][12]) {
    for (int T = 0; T < n_tiles; T++)
        for (int hp = 0; hp < 2; hp++) {
            long acc[16];
            for (int r = 0; r < 16; r++) acc[r] = Cin[((size_t)T * 2 + hp) * 16 + r];
            for (int ks = 0; ks < n_ksteps; ks++)
                for (int h = 0; h < 2; h++)
                    for (int j = 0; j < 16; j++) {
                        u64 w = kw[ks][h][j >> 3];
                        int sb = (int)(int8_t)(((w >> (8 * (j & 7))) & 0xFF) ^ 0x80);       // byte XOR 0x80 as a signed byte
                        for (int r = 0; r < 16; r++) {
                            int rho = rho_of(32 * hp, r);
                            acc[r] += (long)A[(((size_t)T * n_ksteps + ks) * 64 + (32 * h + rho)) * 16 + j] * sb;
                        }
                    }
            for (int half = 0; half < 2; half++) {
                u64 v = 0;
                for (int a = 7; a >= 0; a--) {
                    if (acc[8 * half + a] < 0 || acc[8 * half + a] >= (1L << 24) - (1L << 16)) { printf("accumulator out of range\n"); exit(1); }
                    v = gl_add(gl_mul(v, 256), (u64)acc[8 * half + a] % GL_P);
                }
                out[hp][2 * T + half] = v;
            }
        }
}

// one column through the four two-plane instructions of an MDS layer (PsdmTables: K index (h, j < 12) = byte j & 1 of halfword P of word
// 6h + (j >> 1); register r < 12 of lane half h' = limb 2P + (r & 1) of word 6h' + (r >> 1)); returns the twelve words mod p
static void emulate_mds(const PsdmTables &t, int L, const u64 s[12], u64 out[12]) {
    long limb[12][8];
    for (int P = 0; P < 4; P++)
        for (int hp = 0; hp < 2; hp++) {
            long acc[16];
            for (int r = 0; r < 16; r++) acc[r] = t.Cin_mds[L][P][hp][r];
            for (int h = 0; h < 2; h++)
                for (int j = 0; j < 16; j++) {
                    int sb = 0;
                    if (j < 12) sb = (int)(int8_t)(((s[6 * h + (j >> 1)] >> (16 * P + 8 * (j & 1))) & 0xFF) ^ 0x80);
                    for (int r = 0; r < 16; r++) acc[r] += (long)t.A_mds[32 * h + rho_of(32 * hp, r)][j] * sb;
                }
            for (int r = 0; r < 12; r++) {
                if (acc[r] < 0 || acc[r] >= (1L << 24) - (1L << 16)) { printf("MDS accumulator out of range\n"); exit(1); }
                limb[6 * hp + (r >> 1)][2 * P + (r & 1)] = acc[r];
            }
        }
    for (int W = 0; W < 12; W++) {
        u64 v = 0;
        for (int a = 7; a >= 0; a--) v = gl_add(gl_mul(v, 256), (u64)limb[W][a] % GL_P);
        out[W] = v;
    }
}
static int worst_case_mds(const PsdmTables &t) {
    long gmin = 1L << 40, gmax = -(1L << 40);
    for (int L = 0; L < 8; L++) for (int P = 0; P < 4; P++) for (int hp = 0; hp < 2; hp++) for (int r = 0; r < 16; r++) {
        long lo = 0, hi = 0;
        for (int h = 0; h < 2; h++) for (int j = 0; j < 16; j++) {
            const long d = t.A_mds[32 * h + rho_of(32 * hp, r)][j];
            if (d > 0) { hi += 127 * d; lo -= 128 * d; } else { hi -= 128 * d; lo += 127 * d; }
        }
        const long c = t.Cin_mds[L][P][hp][r];
        if (c + lo < gmin) gmin = c + lo;
        if (c + hi > gmax) gmax = c + hi;
    }
    const bool ok = gmin >= 0 && gmax < (1L << 24) - (1L << 16);
    if (!ok) printf("MDS: limb sums can reach [%ld, %ld], outside [0, 2^24 - 2^16)\n", gmin, gmax);
    return ok ? 0 : 1;
}

// worst case over EVERY operand (each signed byte at whichever end of [-128, 127] hurts most): all limb sums must stay inside
// [0, 2^24 - 2^16), the range the 32-bit limb pairing of psdm::recombine relies on
static int worst_case(const char *name, int n_tiles, int n_ksteps, const int8_t *A, const int32_t *Cin) {
    long gmin = 1L << 40, gmax = -(1L << 40);
    for (int T = 0; T < n_tiles; T++)
        for (int m = 0; m < 32; m++) {
            long lo = 0, hi = 0;
            for (int ks = 0; ks < n_ksteps; ks++)
                for (int h = 0; h < 2; h++)
                    for (int j = 0; j < 16; j++) {
                        const long d = A[(((size_t)T * n_ksteps + ks) * 64 + (32 * h + m)) * 16 + j];
                        if (d > 0) { hi += 127 * d; lo -= 128 * d; } else { hi -= 128 * d; lo += 127 * d; }
                    }
            for (int hh = 0; hh < 2; hh++) for (int r = 0; r < 16; r++) if (rho_of(32 * hh, r) == m) {
                const long c = Cin[((size_t)T * 2 + hh) * 16 + r];
                if (c + lo < gmin) gmin = c + lo;
                if (c + hi > gmax) gmax = c + hi;
            }
        }
    const bool ok = gmin >= 0 && gmax < (1L << 24) - (1L << 16);
    if (!ok) printf("%s: limb sums can reach [%ld, %ld], outside [0, 2^24 - 2^16)\n", name, gmin, gmax);
    return ok ? 0 : 1;
}

int main() {
    static PsdmTables t; build(t);
    {
        int wc = 0;
        wc += worst_case_mds(t);
        for (int b = 0; b < 6; b++) wc += worst_case("F", 1, 3 + b, &t.A_F[b][0][0][0], &t.Cin_F[b][0][0]);
        wc += worst_case("G2", 3, 9, &t.A_G2[0][0][0][0], &t.Cin_G2[0][0][0]);
        if (wc) { printf("matrix-core Poseidon tables: limb range FAIL\n"); return 1; }
    }
    u64 z = 7; int bad = 0;
    for (int trial = 0; trial < 40; trial++) {
        u64 s[12];
        for (int i = 0; i < 12; i++) s[i] = trial < 4 ? (trial & 1 ? ~0ULL : 0) + (u64)(trial >> 1) * i : splitmix(z);      // any 64-bit representatives
        u64 kw[9][2][2];
        for (int ks = 0; ks < 3; ks++) for (int h = 0; h < 2; h++) for (int w = 0; w < 2; w++) kw[ks][h][w] = s[6 * h + 2 * ks + w];
        // ---- MDS layers
        for (int L = 0; L < 8; L++) {
            u64 out[12];
            emulate_mds(t, L, s, out);
            const u64 *nxt = L < 3 ? PSD_RC + 12 * (L + 1) : L == 3 ? PSD_PR_FIRST : L < 7 ? PSD_RC + 12 * (27 + L - 4) : nullptr;
            for (int W = 0; W < 12; W++) {
                u64 want = nxt ? nxt[W] : 0;
                for (int wi = 0; wi < 12; wi++) want = gl_add(want, gl_mul(mds_entry(W, wi), s[wi] % GL_P));
                if (gl_canon(out[W]) != want) { if (bad++ < 5) printf("MDS layer %d word %d mismatch\n", L, W); }
            }
        }
        // ---- partial rounds, straight from the sparse form (s = state with PR_FIRST already added)
        u64 st[12];
        for (int i = 0; i < 12; i++) st[i] = s[i] % GL_P;
        u64 t0[11];
        for (int i = 0; i < 11; i++) { u64 a = 0; for (int j = 0; j < 11; j++) a = gl_add(a, gl_mul(PSD_PR_INIT[11 * i + j], st[1 + j])); t0[i] = a; }
        u64 E_want[22], y[22], x = st[0], tj[11];
        for (int i = 0; i < 11; i++) tj[i] = t0[i];
        for (int q = 0; q < 22; q++) {
            u64 e = 0; for (int i = 0; i < 11; i++) e = gl_add(e, gl_mul(PSD_PR_VHAT[11 * q + i], t0[i]));
            E_want[q] = e;
            y[q] = gl_add(gl_pow(x, 7), PSD_PR_K[q]);
            u64 d = gl_mul(PSD_M00, y[q]);
            for (int i = 0; i < 11; i++) d = gl_add(d, gl_mul(PSD_PR_VHAT[11 * q + i], tj[i]));
            for (int i = 0; i < 11; i++) tj[i] = gl_add(tj[i], gl_mul(PSD_PR_W[11 * q + i], y[q]));
            // the same next s-box input through the triangular form
            u64 d2 = E_want[q];
            for (int j = 0; j <= q; j++) d2 = gl_add(d2, gl_mul(PSD_PR_D[22 * q + j], y[j]));
            if (d != d2) { if (bad++ < 5) printf("PR_D row %d mismatch\n", q); }
            x = d;
        }
        // k-step 3 + b' carries y_{4b'} .. y_{4b'+3}: lane half h, word w -> y_{4b' + 2h + w}
        for (int ks = 3; ks < 9; ks++) for (int h = 0; h < 2; h++) for (int w = 0; w < 2; w++) {
            int q = 4 * (ks - 3) + 2 * h + w;
            kw[ks][h][w] = q < 22 ? y[q] : splitmix(z);         // the two padding slots of the last block may hold anything
        }
        for (int b = 0; b < 6; b++) {
            // F[q] = E[q] + sum_{j < 4b} PR_D[q][j] y_j for the rows q = 4b + 2h' + slot of block b
            u64 out[2][12];
            emulate(1, 3 + b, &t.A_F[b][0][0][0], &t.Cin_F[b][0][0], kw, out);
            for (int hp = 0; hp < 2; hp++) for (int sl = 0; sl < 2; sl++) {
                int q = 4 * b + 2 * hp + sl;
                if (q > 20) continue;
                u64 want = E_want[q];
                for (int j = 0; j < 4 * b; j++) want = gl_add(want, gl_mul(PSD_PR_D[22 * q + j], y[j]));
                if (gl_canon(out[hp][sl]) != want) { if (bad++ < 5) printf("F[%d] mismatch\n", q); }
            }
        }
        {
            u64 out[2][12];
            emulate(3, 9, &t.A_G2[0][0][0][0], &t.Cin_G2[0][0][0], kw, out);
            for (int W = 0; W < 12; W++) {
                u64 want = gl_add(W == 0 ? x : tj[W - 1], PSD_RC[12 * 26 + W]);
                if (gl_canon(out[W / 6][W % 6]) != want) { if (bad++ < 5) printf("G2 word %d mismatch\n", W); }
            }
        }
    }
    printf("matrix-core Poseidon tables vs direct arithmetic: %s (%d mismatches)\n", bad ? "FAIL" : "OK", bad);
    return bad != 0;
}
