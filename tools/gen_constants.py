#!/usr/bin/env python3
"""Regenerates the Poseidon-Goldilocks constant tables and writes them as C headers.

  oracle/poseidon_constants.h                      (oracle: naive rounds only)
  eth-tx-proof_amd/csrc/poseidon_constants.inc     (product: naive + sparse partial-round form)
  tests/golden/poseidon_rc.json                    (the 360 round constants + sha256)

The 360 round constants are ChaCha8Rng::seed_from_u64(0) draws;
the sparse partial-round tables (INIT / K / V_HAT / W) are derived here from the
MDS matrix by the factorisation  M = Sp * blockdiag(1, M_hat)  and are checked
against the naive permutation before anything is written.
"""
import hashlib
import json
import os
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyspec as S  # noqa: E402

P = S.P
R_P = S.N_PARTIAL


def mat_mul(A, B):
    n, m, k = len(A), len(B[0]), len(B)
    return [[sum(A[i][t] * B[t][j] for t in range(k)) % P for j in range(m)] for i in range(n)]


def mat_vec(A, v):
    return [sum(A[i][t] * v[t] for t in range(len(v))) % P for i in range(len(A))]


def mat_inv(A):
    n = len(A)
    M = [list(r) + [1 if i == j else 0 for j in range(n)] for i, r in enumerate(A)]
    for c in range(n):
        piv = next(r for r in range(c, n) if M[r][c] % P)
        M[c], M[piv] = M[piv], M[c]
        iv = pow(M[c][c], P - 2, P)
        M[c] = [x * iv % P for x in M[c]]
        for r in range(n):
            if r != c and M[r][c]:
                f = M[r][c]
                M[r] = [(x - f * y) % P for x, y in zip(M[r], M[c])]
    return [r[n:] for r in M]


def transpose(A):
    return [list(r) for r in zip(*A)]


def derive_sparse():
    RC = S.rc()
    M = S.mds_matrix()
    Minv = mat_inv(M)
    # --- constants: push every partial-round constant vector back through M^-1
    c = [[RC[12 * (4 + i) + j] for j in range(12)] for i in range(R_P)]
    k = [0] * R_P
    for i in range(R_P - 2, -1, -1):
        cp = mat_vec(Minv, c[i + 1])
        for j in range(1, 12):
            c[i][j] = (c[i][j] + cp[j]) % P
        k[i] = cp[0]
    first = c[0]
    # --- matrices: M_cur = Sp_i * blockdiag(1, M_hat);  M_cur <- blockdiag(1, M_hat) * M
    vhat = [None] * R_P
    wcol = [None] * R_P
    Mcur = M
    for i in range(R_P - 1, -1, -1):
        Mhat = [row[1:] for row in Mcur[1:]]
        v = Mcur[0][1:]
        w = [Mcur[r][0] for r in range(1, 12)]
        MhatinvT = transpose(mat_inv(Mhat))
        vhat[i] = mat_vec(MhatinvT, v)
        wcol[i] = w
        D = [[1] + [0] * 11] + [[0] + Mhat[r] for r in range(11)]
        Mcur = mat_mul(D, M)
        init = Mhat
    m00 = M[0][0]
    return first, init, k, vhat, wcol, m00


def poseidon_sparse(state, tabs):
    """Permutation using the sparse partial-round form; must equal S.poseidon."""
    first, init, k, vhat, wcol, m00 = tabs
    RC = S.rc()
    s = [x % P for x in state]
    r = 0
    for _ in range(4):
        s = [(s[i] + RC[12 * r + i]) % P for i in range(12)]
        s = [pow(x, 7, P) for x in s]
        s = S._mds(s)
        r += 1
    s = [(s[i] + first[i]) % P for i in range(12)]
    s = [s[0]] + mat_vec(init, s[1:])
    for i in range(R_P):
        s0 = (pow(s[0], 7, P) + k[i]) % P
        d = (m00 * s0 + sum(vhat[i][j] * s[1 + j] for j in range(11))) % P
        s = [d] + [(s[1 + j] + wcol[i][j] * s0) % P for j in range(11)]
    r += R_P
    for _ in range(4):
        s = [(s[i] + RC[12 * r + i]) % P for i in range(12)]
        s = [pow(x, 7, P) for x in s]
        s = S._mds(s)
        r += 1
    return s


def derive_wave(tabs):
    """Tables of the wave-wide permutation (poseidon_wave.cuh): the 22 partial rounds as ONE serial chain of s-boxes with every
    other word of the state eliminated.  With x = state after PR_FIRST/PR_INIT and u_q = sbox(x0 before round q) + K[q]:
        x_j before round r  = x_j + sum_{q<r} W[q][j-1] u_q                                   (j >= 1)
        x_0 before round k  = M00 u_{k-1} + <VHAT[k-1], x_1..11> + sum_{q<k-1} <VHAT[k-1], W[q]> u_q   (k >= 1)
    Lane k (k = 0..21) accumulates x_0 before round k, lane 32 the final x_0, lane 32+j the final x_j.
      init[i][lane]  : coefficient of z_i (z = state after the first four full rounds + PR_FIRST) in the lane's starting value
      chain[q][lane] : coefficient of u_q in the lane's accumulator"""
    first, init, k, vhat, wcol, m00 = tabs

    def lane_of_x0(kk):
        return kk if kk < R_P else 32
    T_init = [[0] * 64 for _ in range(12)]
    T_chain = [[0] * 64 for _ in range(R_P)]
    T_init[0][0] = 1
    for kk in range(1, R_P + 1):
        ln = lane_of_x0(kk)
        for i in range(1, 12):
            T_init[i][ln] = sum(vhat[kk - 1][j] * init[j][i - 1] for j in range(11)) % P
        for q in range(kk):
            T_chain[q][ln] = m00 if q == kk - 1 else sum(vhat[kk - 1][j] * wcol[q][j] for j in range(11)) % P
    for j in range(1, 12):
        for i in range(1, 12):
            T_init[i][32 + j] = init[j - 1][i - 1]
        for q in range(R_P):
            T_chain[q][32 + j] = wcol[q][j - 1]
    # lane 63 of a chain row carries K[q] (an idle lane: its accumulator is never read), so the round's constant arrives with the
    # round's coefficients instead of through a scalar load in the middle of the dependent chain
    for q in range(R_P):
        T_chain[q][63] = k[q]
    return T_init, T_chain


def poseidon_wave(state, tabs, wave):
    """The permutation as poseidon_wave.cuh computes it (64 "lanes"); must equal S.poseidon."""
    first, init, k, vhat, wcol, m00 = tabs
    T_init, T_chain = wave
    RC = S.rc()
    s = [x % P for x in state]
    r = 0
    for _ in range(4):
        s = [(s[i] + RC[12 * r + i]) % P for i in range(12)]
        s = [pow(x, 7, P) for x in s]
        s = S._mds(s)
        r += 1
    z = [(s[i] + first[i]) % P for i in range(12)]
    acc = [sum(T_init[i][ln] * z[i] for i in range(12)) % P for ln in range(64)]
    for q in range(R_P):
        assert T_chain[q][63] == k[q]
        u = (pow(acc[q], 7, P) + T_chain[q][63]) % P
        acc = [(acc[ln] + T_chain[q][ln] * u) % P for ln in range(64)]
    s = [acc[32 + j] for j in range(12)]
    r += R_P
    for _ in range(4):
        s = [(s[i] + RC[12 * r + i]) % P for i in range(12)]
        s = [pow(x, 7, P) for x in s]
        s = S._mds(s)
        r += 1
    return s


def wave_ksum(wave):
    """the round constants K[q] folded into the lanes' starting values: sum_q chain[q][lane] * K[q] over the lanes that are read
    (0..21: x0 before round k, 32..43: the final state); lane k < 22 is read before rounds >= k have added anything (chain[q][k] = 0
    for q >= k), so the fold is exact for every value that is ever used"""
    T_init, T_chain = wave
    ks = [0] * 64
    for ln in list(range(R_P)) + list(range(32, 44)):
        assert all(T_chain[q][ln] == 0 for q in range(ln, R_P)) or ln >= 32
        ks[ln] = sum(T_chain[q][ln] * T_chain[q][63] for q in range(R_P)) % P
    return ks


def poseidon_wave_folded(state, tabs, wave):
    """poseidon_wave with the K[q] folded into the starting values (the fast path of poseidon_wave.cuh): u_q = x0^7 only"""
    first, init, k, vhat, wcol, m00 = tabs
    T_init, T_chain = wave
    ks = wave_ksum(wave)
    RC = S.rc()
    s = [x % P for x in state]
    r = 0
    for _ in range(4):
        s = [(s[i] + RC[12 * r + i]) % P for i in range(12)]
        s = [pow(x, 7, P) for x in s]
        s = S._mds(s)
        r += 1
    z = [(s[i] + first[i]) % P for i in range(12)]
    acc = [(sum(T_init[i][ln] * z[i] for i in range(12)) + ks[ln]) % P for ln in range(64)]
    for q in range(R_P):
        u = pow(acc[q], 7, P)
        acc = [(acc[ln] + T_chain[q][ln] * u) % P if ln != 63 else 0 for ln in range(64)]
    s = [acc[32 + j] for j in range(12)]
    r += R_P
    for _ in range(4):
        s = [(s[i] + RC[12 * r + i]) % P for i in range(12)]
        s = [pow(x, 7, P) for x in s]
        s = S._mds(s)
        r += 1
    return s


def c_array(name, vals, per_line=4, ctype="uint64_t", qual="static const"):
    lines = []
    for i in range(0, len(vals), per_line):
        lines.append("    " + ", ".join("0x%016xULL" % v for v in vals[i:i + per_line]) + ",")
    return "%s %s %s[%d] = {\n%s\n};\n" % (qual, ctype, name, len(vals), "\n".join(lines))


def main():
    RC = S.rc()
    tabs = derive_sparse()
    import random
    rnd = random.Random(1)
    tests = [k[0] for k in S.KATS] + [[rnd.randrange(P) for _ in range(12)] for _ in range(8)]
    wave = derive_wave(tabs)
    for t in tests:
        assert poseidon_sparse(t, tabs) == S.poseidon(t)
        assert poseidon_wave(t, tabs, wave) == S.poseidon(t)
        assert poseidon_wave_folded(t, tabs, wave) == S.poseidon(t)
    first, init, k, vhat, wcol, m00 = tabs
    hdr = ("/* GENERATED by tools/gen_constants.py -- do not edit.\n"
           " * Poseidon-Goldilocks round constants = 360 x ChaCha8Rng::seed_from_u64(0).gen_range(0..p)\n"
           " * (plonky2 0.2.2 hash/poseidon_goldilocks.rs ALL_ROUND_CONSTANTS; SURVEY.md A2).\n"
           " * sha256(LE bytes) = %s */\n" % S.RC_SHA256)
    with open(os.path.join(ROOT, "oracle", "poseidon_constants.h"), "w") as f:
        f.write(hdr + "#include <stdint.h>\n")
        f.write(c_array("POSEIDON_RC", RC))
        f.write("static const uint64_t POSEIDON_MDS_CIRC[12] = {%s};\n" % ", ".join(map(str, S.MDS_CIRC)))
        f.write("static const uint64_t POSEIDON_MDS_DIAG[12] = {%s};\n" % ", ".join(map(str, S.MDS_DIAG)))
    os.makedirs(os.path.join(ROOT, "eth-tx-proof_amd", "csrc"), exist_ok=True)
    with open(os.path.join(ROOT, "eth-tx-proof_amd", "csrc", "poseidon_constants.inc"), "w") as f:
        f.write(hdr)
        f.write("/* Sparse partial-round form (derived, checked == naive):\n"
                " *   s += PR_FIRST;  s[1..] = PR_INIT * s[1..];\n"
                " *   for i<22: s0 = s0^7 + PR_K[i]; d = 25*s0 + <PR_VHAT[i], s[1..]>;\n"
                " *             s[j] += PR_W[i][j-1]*s0 (j>=1); s[0] = d\n */\n")
        f.write(c_array("PSD_RC", RC, qual="PSD_CONST"))
        f.write(c_array("PSD_PR_FIRST", first, qual="PSD_CONST"))
        f.write(c_array("PSD_PR_INIT", [x for row in init for x in row], per_line=11, qual="PSD_CONST"))
        f.write(c_array("PSD_PR_K", k, qual="PSD_CONST"))
        f.write(c_array("PSD_PR_VHAT", [x for row in vhat for x in row], per_line=11, qual="PSD_CONST"))
        f.write(c_array("PSD_PR_W", [x for row in wcol for x in row], per_line=11, qual="PSD_CONST"))
        # s-box chain of the partial rounds with the untouched words eliminated (poseidon_mfma.cuh):
        #   x_{q+1} = E[q] + sum_{j<=q} PR_D[q][j] * y_j,  PR_D[q][j] = <PR_VHAT[q], PR_W[j]> (j < q), PR_D[q][q] = M00
        dmat = [[(m00 if j == q else sum(vhat[q][i] * wcol[j][i] for i in range(11)) % P) if j <= q else 0 for j in range(R_P)]
                for q in range(R_P)]
        f.write(c_array("PSD_PR_D", [x for row in dmat for x in row], per_line=11, qual="PSD_CONST"))
        f.write("#define PSD_M00 %dULL\n" % m00)
        # wave-wide permutation (poseidon_wave.cuh): see derive_wave()
        # (16-byte aligned: poseidon_wave.cuh copies these five into LDS two words at a time)
        f.write(c_array("PSD_WV_INIT", [x for row in wave[0] for x in row], per_line=8, qual="PSD_CONST __attribute__((aligned(16)))"))
        f.write(c_array("PSD_WV_CHAIN", [x for row in wave[1] for x in row], per_line=8, qual="PSD_CONST __attribute__((aligned(16)))"))
        # MDS rows per lane: out_l = sum_i PSD_WV_MDS[i][l] * s_i (zero rows for the idle lanes l >= 12)
        mds = [[(S.MDS_CIRC[(i - l) % 12] + (S.MDS_DIAG[0] if i == 0 and l == 0 else 0)) if l < 12 else 0 for l in range(64)] for i in range(12)]
        Mfull = S.mds_matrix()
        assert all(mds[i][l] == Mfull[l][i] for i in range(12) for l in range(12)) and all(S.MDS_DIAG[i] == 0 for i in range(1, 12))
        f.write(c_array("PSD_WV_MDS", [x for row in mds for x in row], per_line=8, qual="PSD_CONST __attribute__((aligned(16)))"))
        # per-lane constants of the eight full rounds (rows 0..3: rounds 0..3, rows 4..7: rounds 26..29) and PR_FIRST as row 8
        wrc = [[RC[12 * rnd + l] if l < 12 else 0 for l in range(64)] for rnd in (0, 1, 2, 3, 26, 27, 28, 29)] + [[first[l] if l < 12 else 0 for l in range(64)]]
        f.write(c_array("PSD_WV_RC", [x for row in wrc for x in row], per_line=8, qual="PSD_CONST __attribute__((aligned(16)))"))
        # the partial rounds' constants folded into the lanes' starting values (fast path of poseidon_wave.cuh): see wave_ksum()
        f.write(c_array("PSD_WV_KSUM", wave_ksum(wave), per_line=8, qual="PSD_CONST __attribute__((aligned(16)))"))
    os.makedirs(os.path.join(ROOT, "tests", "golden"), exist_ok=True)
    with open(os.path.join(ROOT, "tests", "golden", "poseidon_rc.json"), "w") as f:
        json.dump({"sha256_le_u64": hashlib.sha256(b"".join(struct.pack("<Q", c) for c in RC)).hexdigest(),
                   "round_constants_hex": ["%016x" % c for c in RC]}, f, indent=0)
    print("ok: m00=%d max(wcol)=%x" % (m00, max(max(r) for r in wcol)))


if __name__ == "__main__":
    main()
