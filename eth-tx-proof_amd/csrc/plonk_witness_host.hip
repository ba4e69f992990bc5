// CPU stand-in for recursion WITNESS GENERATION (VERDICT r4 #5b).  Upstream produces the witness of every recursion proof on the CPU
// (plonky2 0.2.2 iop::generator::generate_partial_witness: one pass of the gates' SimpleGenerators, single-threaded per proof) before
// proving it -- ops::TxProof::execute -> generate_txn_proof -> prove_root, /root/reference/ops/src/lib.rs:52.  Round 4's transaction
// record contained none of that work: SyntheticRecursion keeps one precomputed witness per circuit.  The real circuits are absent
// (SURVEY A10), so the generators cannot be reproduced; what can be is their COST on the stand-in circuits: the cells that upstream's
// PoseidonGenerator and ArithmeticBaseGenerator compute are recomputed here, row by row, from the rows' input cells -- the permutation
// with every s-box input the gate exposes (135 wires a row), the twenty multiply-adds of an ArithmeticGate row -- and written back into
// the witness.  The witness satisfies its circuit already, so the cells get the values they had: proofs stay the same words, which is
// also the check (recompute returns the number of cells whose value CHANGED; the caller requires 0).  Host code only; called from the
// thread that proves the circuit (one host thread per chain in flight).
#include "zk_internal.h"
#include "transcript_host_core.h"       // psdh: the scalar field arithmetic and the sparse-form Poseidon constants

namespace {
using psdh::u128; using psdh::mul; using psdh::add; using psdh::red; using psdh::sbox;
inline u64 sub(u64 a, u64 b) { return a >= b ? a - b : a + (psdh::P - b); }
enum { PG_IN = 0, PG_OUT = 12, PG_SWAP = 24, PG_DELTA = 25, PG_FULL0 = 29, PG_PARTIAL = 65, PG_FULL1 = 87 };      // gates.py (plonky2 PoseidonGate wire layout)

// one PoseidonGate row: column c of the row is w[c * stride]
inline u64 poseidon_gate_row(u64 *w, size_t stride) {
    u64 changed = 0;
    auto put = [&](int col, u64 v) { u64 &cell = w[(size_t)col * stride]; changed += cell != v; cell = v; };
    u64 in[12], s[12];
    for (int i = 0; i < 12; i++) in[i] = w[(size_t)(PG_IN + i) * stride];
    const u64 swap = w[(size_t)PG_SWAP * stride];
    for (int i = 0; i < 4; i++) {
        const u64 d = mul(swap, sub(in[4 + i], in[i]));
        put(PG_DELTA + i, d);
        s[i] = add(in[i], d); s[4 + i] = sub(in[4 + i], d); s[8 + i] = in[8 + i];
    }
    for (int r = 0; r < 4; r++) {
        for (int i = 0; i < 12; i++) s[i] = add(s[i], psdh::PSD_RC[12 * r + i]);
        if (r) for (int i = 0; i < 12; i++) put(PG_FULL0 + 12 * (r - 1) + i, s[i]);
        for (int i = 0; i < 12; i++) s[i] = sbox(s[i]);
        psdh::mds(s, nullptr);
    }
    for (int i = 0; i < 12; i++) s[i] = add(s[i], psdh::PSD_PR_FIRST[i]);
    u64 rest[11];
    for (int i = 0; i < 11; i++) { u128 a = 0; for (int j = 0; j < 11; j++) a += (u128)mul(psdh::PSD_PR_INIT[11 * i + j], s[1 + j]); rest[i] = red(a); }
    u64 s0 = s[0];
    for (int r = 0; r < 22; r++) {
        put(PG_PARTIAL + r, s0);
        const u64 x = add(sbox(s0), psdh::PSD_PR_K[r]);
        u128 d = (u128)x * PSD_M00;
        const u64 *vh = psdh::PSD_PR_VHAT + 11 * r, *wv = psdh::PSD_PR_W + 11 * r;
        for (int j = 0; j < 11; j++) d += (u128)mul(vh[j], rest[j]);
        for (int j = 0; j < 11; j++) rest[j] = add(rest[j], mul(wv[j], x));
        s0 = red(d);
    }
    s[0] = s0;
    for (int j = 0; j < 11; j++) s[1 + j] = rest[j];
    for (int r = 0; r < 4; r++) {
        for (int i = 0; i < 12; i++) s[i] = add(s[i], psdh::PSD_RC[12 * (26 + r) + i]);
        for (int i = 0; i < 12; i++) put(PG_FULL1 + 12 * r + i, s[i]);
        for (int i = 0; i < 12; i++) s[i] = sbox(s[i]);
        psdh::mds(s, nullptr);
    }
    for (int i = 0; i < 12; i++) put(PG_OUT + i, s[i]);
    return changed;
}
// one ArithmeticGate{20} row: output_j = c0 m0 m1 + c1 addend
inline u64 arithmetic_gate_row(u64 *w, size_t stride, u64 c0, u64 c1) {
    u64 changed = 0;
    for (int j = 0; j < 20; j++) {
        const u64 v = add(mul(mul(w[(size_t)(4 * j) * stride], w[(size_t)(4 * j + 1) * stride]), c0), mul(w[(size_t)(4 * j + 2) * stride], c1));
        u64 &cell = w[(size_t)(4 * j + 3) * stride];
        changed += cell != v; cell = v;
    }
    return changed;
}
}  // namespace

// wires: num_wires x n column-major (column c at wires + c * n), host memory.  poseidon_rows / arith_rows: the rows of those gates;
// arith_c0 / arith_c1: the two gate constants of every ArithmeticGate row.  Returns the number of cells whose value changed (0 for a
// witness that satisfied its circuit), or a negative ZK_ERR_*.
extern "C" int64_t zk_host_plonk_witness_recompute(uint64_t *wires, uint64_t n, uint32_t num_wires, const uint32_t *poseidon_rows, uint64_t n_poseidon,
                                                   const uint32_t *arith_rows, const uint64_t *arith_c0, const uint64_t *arith_c1, uint64_t n_arith) {
    if (!wires || num_wires < 135 || (!poseidon_rows && n_poseidon) || ((!arith_rows || !arith_c0 || !arith_c1) && n_arith)) return ZK_ERR_INVALID_ARG;
    u64 changed = 0;
    for (u64 k = 0; k < n_poseidon; k++) { if (poseidon_rows[k] >= n) return ZK_ERR_INVALID_ARG; changed += poseidon_gate_row(wires + poseidon_rows[k], (size_t)n); }
    for (u64 k = 0; k < n_arith; k++) { if (arith_rows[k] >= n) return ZK_ERR_INVALID_ARG; changed += arithmetic_gate_row(wires + arith_rows[k], (size_t)n, arith_c0[k], arith_c1[k]); }
    return (int64_t)changed;
}
