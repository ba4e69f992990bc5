/* ORACLE -- test infrastructure, not product code.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may link or call this.
 *
 * CPU restatement (plain C, gcc) of the proving hot path the reference enters at
 *   /root/reference/ops/src/lib.rs:52   generate_txn_proof(p_state(), input, None)
 * with the parameters fixed at
 *   /root/reference/common/src/prover_state/circuit.rs:200-206   StarkConfig::standard_fast_config()
 * The algorithms live in crates that are NOT in /root/reference (Cargo.lock:3441-3443 plonky2 0.2.2,
 * :3466-3468 plonky2_field 0.2.2, :4529-4531 starky 0.4.0); each function cites the upstream item it
 * restates and the SURVEY.md appendix paragraph that holds the restated spec.
 *
 * PARITY STATUS: the field constants, the 360 Poseidon round constants and the permutation are pinned by
 * upstream known-answer vectors (tests/golden).  Everything above the permutation -- sponge mode, Merkle
 * layout, challenger, STARK transcript, FRI -- is PARITY UNPINNED (no golden vector exists in the reference,
 * SURVEY.md 8c); its evidence is internal: prover output is accepted by the independent verifier below.
 */
#ifndef ORACLE_H
#define ORACLE_H
#include <stdint.h>
#include <stddef.h>
#include "../include/zkgl.h" /* shared plain-data structs only: zk_air, zk_instr, zk_stark_config */

/* Hasher::hash_pad pads to a multiple of this many words: RATE (8), restated from recall; see ZK_HASH_PAD_MULTIPLE in include/zkgl.h */
#define ORC_HASH_PAD_MULTIPLE 8u

#ifdef __cplusplus
extern "C" {
#endif

int32_t orc_selftest(void);
void orc_set_threads(int32_t n);
int32_t orc_get_threads(void);

/* field helpers for tests */
uint64_t orc_gl_mul(uint64_t a, uint64_t b);
uint64_t orc_gl_inv(uint64_t a);
uint64_t orc_gl_root(uint32_t k);

/* plonky2::hash::poseidon */
void orc_poseidon(uint64_t state[12]);
void orc_poseidon_batch(uint64_t *states, size_t n);
void orc_hash_no_pad(const uint64_t *in, size_t len, uint64_t out[4]);
void orc_hash_or_noop(const uint64_t *in, size_t len, uint64_t out[4]);
void orc_two_to_one(const uint64_t l[4], const uint64_t r[4], uint64_t out[4]);
void orc_hash_rows(const uint64_t *rows, size_t n_rows, uint32_t row_len, uint64_t *out);

/* plonky2_field::fft -- natural order in and out, in place, n_cols columns (column-major) */
void orc_fft(uint64_t *data, uint32_t log_n, uint32_t n_cols, int32_t inverse, uint64_t shift);

/* plonky2::hash::merkle_tree */
int32_t orc_merkle_build(const uint64_t *leaves, size_t n_leaves, uint32_t leaf_len, uint32_t cap_height,
                         uint64_t *cap_out, uint64_t *digests_out);
int32_t orc_merkle_prove(const uint64_t *leaves, size_t n_leaves, uint32_t leaf_len, uint32_t cap_height,
                         uint32_t leaf_idx, uint64_t *siblings_out);
int32_t orc_merkle_verify(const uint64_t *leaf, uint32_t leaf_len, uint32_t leaf_idx, const uint64_t *siblings,
                          uint32_t n_siblings, const uint64_t *cap);

typedef struct { uint64_t **levels; uint32_t n_levels, cap_height; size_t n_leaves; } orc_tree;
orc_tree *orc_tree_new(const uint64_t *leaves, size_t n_leaves, uint32_t leaf_len, uint32_t cap_height);
const uint64_t *orc_tree_cap(const orc_tree *t);
void orc_tree_prove(const orc_tree *t, uint32_t leaf_idx, uint64_t *siblings_out);
void orc_tree_free(orc_tree *t);

/* plonky2::iop::challenger */
typedef struct {
    uint64_t state[12];
    uint64_t inbuf[8];
    uint64_t outbuf[8];
    uint32_t n_in, n_out;
} orc_challenger;
void orc_challenger_init(orc_challenger *c);
void orc_challenger_observe(orc_challenger *c, const uint64_t *x, uint32_t n);
uint64_t orc_challenger_get(orc_challenger *c);
void orc_challenger_get_n(orc_challenger *c, uint32_t n, uint64_t *out);
void orc_challenger_compact(orc_challenger *c, uint64_t out[12]);

/* plonky2::fri::oracle::PolynomialBatch */
typedef struct {
    uint32_t log_n, n_cols, rate_bits, cap_height;
    uint64_t *coeffs;  /* n_cols x n, column-major, natural order */
    uint64_t *leaves;  /* (n << rate_bits) x n_cols row-major; row r = values at g*w^bitrev(r) */
    uint64_t *digests; /* upstream recursive layout */
    uint64_t *cap;     /* 2^cap_height x 4 */
} orc_batch;
orc_batch *orc_batch_from_values(const uint64_t *cols, uint32_t log_n, uint32_t n_cols, uint32_t rate_bits, uint32_t cap_height);
orc_batch *orc_batch_from_coeffs(const uint64_t *cols, uint32_t log_n, uint32_t n_cols, uint32_t rate_bits, uint32_t cap_height);
void orc_batch_free(orc_batch *b);
const uint64_t *orc_batch_cap(const orc_batch *b);
const uint64_t *orc_batch_leaves(const orc_batch *b);
const uint64_t *orc_batch_coeffs(const orc_batch *b);
const uint64_t *orc_batch_digests(const orc_batch *b);
void orc_batch_prove(const orc_batch *b, uint32_t leaf_idx, uint64_t *siblings_out);
void orc_batch_eval_at(const orc_batch *b, const uint64_t point[2], uint64_t *out);

/* synthetic witness: fills n_trace_cols x 2^log_n (column-major) for the synthetic AIR family of
 * SURVEY.md 8d cfg3 and writes its public inputs (first-row a0,b0 and last-row a0). */
void orc_synth_trace(uint32_t n_cols, uint32_t log_n, uint64_t seed, uint64_t *trace_out, uint64_t public_inputs_out[3]);
/* auxiliary (logUp/CTL-shaped) columns for given challenges [beta0,gamma0,beta1,gamma1..]; returns 0 or ZK_ERR_ZERO_INVERSE */
int32_t orc_aux_columns(const zk_air *air, uint32_t num_challenges, const uint64_t *challenges, const uint64_t *trace,
                        uint32_t log_n, uint64_t *aux_out);

/* starky::prover + evm_arithmetization::prover transcript over n_tables tables sharing one challenger */
typedef struct {
    const zk_air *air;
    uint32_t log_n;
    uint32_t reserved;
    const uint64_t *trace;          /* n_trace_cols x n column-major */
    const uint64_t *public_inputs;  /* n_public_inputs */
} orc_table;
int32_t orc_prove(const zk_stark_config *cfg, uint32_t n_tables, const orc_table *tables, uint64_t *proof_out,
                  size_t proof_cap_words, size_t *proof_len_words);
/* starky::verifier + plonky2::fri::verifier; trace pointers in `tables` are ignored */
int32_t orc_verify(const zk_stark_config *cfg, uint32_t n_tables, const orc_table *tables, const uint64_t *proof,
                   size_t proof_len_words);
/* finer-grained pieces, one per C-ABI entry point of the same name (zk_quotient, zk_pow_grind, zk_fri_prove_openings) */
int32_t orc_quotient(const zk_stark_config *cfg, const zk_air *air, uint32_t log_n, const orc_batch *trace_b, const orc_batch *aux_b,
                     const uint64_t *public_inputs, const uint64_t *challenges, const uint64_t *alphas, uint64_t *out);
int32_t orc_pow_grind(const uint64_t state[12], uint32_t pos, uint32_t min_zeros, uint64_t *witness);
int32_t orc_fri_prove(const zk_stark_config *cfg, const orc_batch *trace_b, const orc_batch *aux_b, const orc_batch *quot_b,
                      orc_challenger *ch, const uint64_t zeta[2], uint64_t *out, size_t cap_words, size_t *len_words);
int32_t orc_fri_prove_instance(const zk_stark_config *cfg, uint32_t n_oracles, const orc_batch *const *oracles, uint32_t n_batches,
                               const zk_fri_batch *batches, orc_challenger *ch, uint64_t *out, size_t cap_words, size_t *len_words);
int32_t orc_fri_combine(const zk_stark_config *cfg, uint32_t n_oracles, const orc_batch *const *oracles, uint32_t n_batches,
                        const zk_fri_batch *batches, const uint64_t alpha[2], uint64_t *out);
size_t orc_proof_size_words(const zk_stark_config *cfg, uint32_t n_tables, const orc_table *tables);

/* plonky2::plonk::{prover, verifier} over a circuit given as plain data (zk_plonk_circuit, include/zkgl.h); plonk.c */
typedef struct orc_plonk_state orc_plonk_state;
orc_plonk_state *orc_plonk_setup(const zk_stark_config *cfg, const zk_plonk_circuit *circuit);
void orc_plonk_state_free(orc_plonk_state *st);
void orc_plonk_digest(const orc_plonk_state *st, uint64_t out[4]);
size_t orc_plonk_proof_size_words(const zk_stark_config *cfg, const zk_plonk_circuit *circuit);
int32_t orc_prove_plonk(const zk_stark_config *cfg, orc_plonk_state *st, const uint64_t *wires, const uint64_t *public_inputs, uint32_t n_public_inputs,
                        uint64_t *proof_out, size_t cap_words, size_t *len_words);
int32_t orc_verify_plonk(const zk_stark_config *cfg, const orc_plonk_state *st, const uint64_t *public_inputs, uint32_t n_public_inputs,
                         const uint64_t *proof, size_t len_words);

#ifdef __cplusplus
}
#endif
#endif
