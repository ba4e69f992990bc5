/*
 * zkgl.h -- C ABI of the MI355X-native Goldilocks/Poseidon/FRI STARK prover backend.
 *
 * Drop-in boundary.  The reference reaches this path through one call,
 *   generate_txn_proof(p_state(), input, None)            /root/reference/ops/src/lib.rs:52
 * configured by
 *   AllRecursiveCircuits::new(&AllStark::default(), .., &StarkConfig::standard_fast_config())
 *                                                         /root/reference/common/src/prover_state/circuit.rs:200-206
 * and the arithmetic itself lives in the crates pinned at /root/reference/Cargo.lock:3441-3443
 * (plonky2 0.2.2), :3466-3468 (plonky2_field 0.2.2), :4529-4531 (starky 0.4.0).  Every entry point
 * below names the upstream item a Rust `-sys` crate would forward to it.
 *
 * Conventions
 *   - every function returns an int32_t status (ZK_OK == 0, negative == error); nothing throws or aborts;
 *   - a zk_ctx is bound to ONE device and ONE HIP stream; use one ctx per worker thread;
 *   - field elements are uint64_t; inputs may be non-canonical (in [p, 2^64)), outputs are canonical;
 *   - extension elements (F_p[x]/(x^2-7)) are two consecutive uint64_t [a0, a1];
 *   - pointers named d_* are device pointers (HBM), everything else is host memory;
 *   - polynomials are column-major: column c of an n-row table starts at base + c*n.
 *   - there is NO CPU fallback: without a usable HIP device zk_ctx_create fails with ZK_ERR_NO_DEVICE.
 */
#ifndef ZKGL_H
#define ZKGL_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ZK_OK 0
#define ZK_ERR_INVALID_ARG (-1)
#define ZK_ERR_OOM (-2)
#define ZK_ERR_HIP (-3)
#define ZK_ERR_QUOTIENT_DEGREE (-4)   /* upstream: trim_to_len panics "quotient not divisible" */
#define ZK_ERR_ZERO_INVERSE (-5)      /* upstream: batch_multiplicative_inverse asserts no zeros */
#define ZK_ERR_POW (-6)               /* no proof-of-work witness in the search window */
#define ZK_ERR_NO_DEVICE (-7)
#define ZK_ERR_ZETA_IN_SUBGROUP (-8)  /* upstream: ensure!(zeta^n != 1) */
#define ZK_ERR_BUFFER_TOO_SMALL (-9)

#define ZK_GOLDILOCKS_P 0xFFFFFFFF00000001ULL
#define ZK_COSET_SHIFT 14293326489335486720ULL /* F::coset_shift() = MULTIPLICATIVE_GROUP_GENERATOR */

typedef struct zk_ctx zk_ctx;
typedef struct zk_batch zk_batch;           /* device-resident PolynomialBatch */
typedef struct zk_challenger zk_challenger; /* device-resident Challenger<F, PoseidonHash> */

/* ------------------------------------------------------------------ context */
/* stream: a hipStream_t to enqueue on (e.g. torch's current stream), or NULL to create a private one. */
int32_t zk_ctx_create(int32_t device, void *stream, zk_ctx **out);
int32_t zk_ctx_destroy(zk_ctx *ctx);
/* Latency switch of zk_prove_tables for proofs of several tables (default off): the trace commitment of the table with the most cells --
 * for the zkEVM's shapes a hash of hundreds of permutations per leaf over few leaves, i.e. one wave per SIMD -- runs on a second stream
 * beside the other tables' commitments (they are independent until their caps are observed, in table order, afterwards).  Same proof.
 * One transaction alone gains a few milliseconds; with the chip saturated by other proofs there is nothing to gain. */
int32_t zk_ctx_set_side_commit(zk_ctx *ctx, int32_t on);
/* Latency switch: the Fiat-Shamir sponge of this context's proofs on a host thread instead of the device-resident one-wave kernel (a CPU
 * core permutes in about a microsecond, the wave in 7.5; SURVEY.md 8a row a8 placed the challenger on the host).  No host synchronisation is
 * added: observed words reach the host by asynchronous copies, challenges come back behind a stream-side wait.  Same proof words either way.
 * Call it between proofs.  Environment: ZK_TRANSCRIPT_HOST=1 sets it for every context. */
int32_t zk_ctx_set_host_transcript(zk_ctx *ctx, int32_t on);
/* The host challenger's permutation and sponge by themselves (pure host code, no context, no GPU): for the CPU test suite, which pins them
 * to upstream's known-answer vectors.  zk_host_challenger_script runs one pass over a script: ops[i] > 0 observes that many of the next
 * `elems`, ops[i] < 0 squeezes -ops[i] challenges into `out`, ops[i] == 0 is Challenger::compact; returns the number of challenges written. */
int32_t zk_host_poseidon_permute(uint64_t state[12]);
int64_t zk_host_challenger_script(const int32_t *ops, uint32_t n_ops, const uint64_t *elems, size_t n_elems, uint64_t *out, size_t cap_out,
                                  uint64_t state_out[12]);
/* Flight recorder (csrc/fault_recorder.hip): the library registers a handler for the runtime's GPU-memory-fault event that writes to stderr
 * what the runtime and the library know about the faulting address (and the same address 4 GiB higher / lower), the nearest live allocations
 * on both sides, and every context's last launches, and then ends the process with exit code 134 (the runtime's own handler would abort
 * it anyway; on the measured stack a faulted process otherwise lingers until it is killed).  ZK_FAULT_EXIT=0: report, then the runtime's
 * handler as before.  ZK_FAULT_RECORDER=0 in the environment leaves the handler unregistered.  zk_debug_fault_report writes that report for any address without
 * a fault (returns 0 when the handler is registered). */
int32_t zk_debug_fault_report(uint64_t addr);
int32_t zk_ctx_synchronize(zk_ctx *ctx);
/* Gives the context's CACHED device memory back to the driver: the free list of its block allocator (blocks are kept for reuse after a
 * proof -- after a 2^26-row table that is tens of GB no other context of the process can get at) and the per-position NTT factor tables
 * of transforms above 2^24 points (up to 2 GiB each; rebuilt on demand).  Waits for the stream first.  Live objects (batches, Plonk
 * states, challengers) are untouched.  bytes_freed (optional): what went back. */
int32_t zk_ctx_trim(zk_ctx *ctx, uint64_t *bytes_freed);
const char *zk_last_error(zk_ctx *ctx);
/* returns the hipStream_t the ctx enqueues on (for callers that record their own HIP events) */
void *zk_ctx_stream(zk_ctx *ctx);

/* Per-kernel hipEvent timing on the ctx stream (mirrors upstream's `timed!` scopes that the reference
 * creates and drops at ops/src/lib.rs:52 via TimingTree::default()).  Off by default. */
int32_t zk_profile_enable(zk_ctx *ctx, int32_t on);
int32_t zk_profile_reset(zk_ctx *ctx);
/* number of distinct kernel names seen; zk_profile_get(i) -> name, launches, total milliseconds */
int32_t zk_profile_count(zk_ctx *ctx);
int32_t zk_profile_get(zk_ctx *ctx, int32_t i, const char **name, uint64_t *launches, double *total_ms);

/* Device memory for callers without HIP bindings (Rust side keeps only opaque pointers).  zk_dev_upload / zk_dev_download are synchronous
 * and accept any host memory: the library never hands pageable memory to the DMA engines -- every such copy bounces through pinned memory
 * owned by the context. */
int32_t zk_dev_alloc(zk_ctx *ctx, size_t bytes, void **d_ptr);
int32_t zk_dev_free(zk_ctx *ctx, void *d_ptr);
int32_t zk_dev_upload(zk_ctx *ctx, void *d_dst, const void *src, size_t bytes);
int32_t zk_dev_download(zk_ctx *ctx, void *dst, const void *d_src, size_t bytes);

/* Host buffers the GPU can read at full PCIe rate.  A witness held in memory from zk_host_alloc (pinned) is uploaded by
 * zk_prove_tables with one asynchronous copy on the context's stream -- no staging, no host wait; traces of other in-flight
 * contexts keep the chip busy meanwhile.  Ordinary (pageable) host memory is accepted everywhere too: it is then staged through
 * two pinned chunks of the context (the CPU fills one while the DMA engine drains the other). */
int32_t zk_host_alloc(zk_ctx *ctx, size_t bytes, void **h_ptr);
int32_t zk_host_free(zk_ctx *ctx, void *h_ptr);
/* Prefetch for a single worker: start uploading the NEXT unit's trace (n_words u64) on the context's copy stream; *d_ptr is device
 * memory owned by the context (zk_dev_free it after the proof).  Returns at once for a PINNED source (zk_host_alloc); a pageable source
 * is staged through the context's pinned chunks by the calling thread, so the call returns when the CPU has copied the last chunk.  zk_prove_tables /
 * zk_prove_plonk given this pointer (trace_on_device = 1) first wait, on the device, for the upload that fills it -- and only for
 * that one, so a prefetch of the unit after next never holds a proof back.  Call it BEFORE the zk_prove_tables of the current
 * unit so that the copy overlaps that proof. */
int32_t zk_trace_upload_async(zk_ctx *ctx, const uint64_t *trace, size_t n_words, uint64_t **d_ptr);
/* A trace with NARROW columns over PCIe: column j of `packed` holds its n = 2^log_n rows in col_bytes[j] bytes each (1, 2, 4 or 8,
 * little-endian, unsigned; every column starts at the next multiple of its element size after the previous one); the device widens to
 * the usual n_cols x n uint64 layout in memory owned by the context (zk_dev_free it after the proof; pass it with trace_on_device = 1).
 * Most columns of the zkEVM's tables are flags, bytes and 16 / 32-bit limbs: their traces cross the link 2-6 x smaller this way (on hosts
 * with 21-24 GB/s links the unit proof is link-bound from host-resident 64-bit traces).  Enqueued on the context's
 * stream; returns at once for a pinned source. */
#define ZK_PACKED_MAX_COLS 65535u
int32_t zk_trace_upload_packed(zk_ctx *ctx, const void *packed, uint32_t log_n, uint32_t n_cols, const uint8_t *col_bytes, uint64_t **d_ptr);

/* ------------------------------------------------------------------ field / NTT
 * plonky2_field::fft::{fft_with_options, ifft_with_options}, PolynomialCoeffs::coset_fft_with_options,
 * PolynomialValues::coset_ifft.  Natural order in, natural order out, in place, n_cols independent columns.
 *   inverse == 0:  v[i] = sum_j c[j] (shift^j) w_n^{ij}          (shift == 0 or 1: plain FFT)
 *   inverse == 1:  c[j] = shift^{-j} n^{-1} sum_i v[i] w_n^{-ij}
 */
int32_t zk_ntt_batch(zk_ctx *ctx, uint64_t *data, uint32_t log_n, uint32_t n_cols, int32_t inverse, uint64_t shift);
int32_t zk_ntt_batch_dev(zk_ctx *ctx, uint64_t *d_data, uint32_t log_n, uint32_t n_cols, int32_t inverse, uint64_t shift);
/* the same with upstream's `zero_factor: Option<usize>` (fft_with_options / ifft_with_options / coset_fft_with_options): zero_factor = r
 * promises that only the first 2^log_n >> r inputs of every column are non-zero; the rest of the input is TAKEN AS zero (upstream skips
 * the butterfly layers that would only copy).  r = 0: no promise (None).  The forward transform then runs as 2^r transforms of size
 * 2^(log_n - r) on cosets instead of one of size 2^log_n.  r > log_n: ZK_ERR_INVALID_ARG. */
int32_t zk_ntt_batch_opts(zk_ctx *ctx, uint64_t *data, uint32_t log_n, uint32_t n_cols, int32_t inverse, uint64_t shift, uint32_t zero_factor);
int32_t zk_ntt_batch_opts_dev(zk_ctx *ctx, uint64_t *d_data, uint32_t log_n, uint32_t n_cols, int32_t inverse, uint64_t shift, uint32_t zero_factor);

/* ------------------------------------------------------------------ Poseidon
 * plonky2::hash::poseidon::{Poseidon::poseidon, PoseidonHash::{hash_no_pad, two_to_one}}, hashing::hash_or_noop.
 */
int32_t zk_poseidon_permute_batch(zk_ctx *ctx, uint64_t *states /* n x 12 */, size_t n);
/* rows: n_rows x row_len row-major;  out: n_rows x 4.  hash_or_noop semantics when row_len <= 4. */
int32_t zk_hash_no_pad_batch(zk_ctx *ctx, const uint64_t *rows, size_t n_rows, uint32_t row_len, uint64_t *out);
int32_t zk_two_to_one_batch(zk_ctx *ctx, const uint64_t *left, const uint64_t *right, size_t n, uint64_t *out);

/* ------------------------------------------------------------------ Merkle tree over explicit leaves
 * plonky2::hash::merkle_tree::MerkleTree::new(leaves, cap_height).  leaves: n_leaves x leaf_len row-major (host).
 * cap_out: 2^cap_height x 4.  digests_out (may be NULL): 2*(n_leaves - 2^cap_height) x 4 in upstream's recursive
 * layout  left_subtree | left_digest | right_digest | right_subtree  per cap sub-tree.
 */
int32_t zk_merkle_build(zk_ctx *ctx, const uint64_t *leaves, size_t n_leaves, uint32_t leaf_len, uint32_t cap_height,
                        uint64_t *cap_out, uint64_t *digests_out);

/* ------------------------------------------------------------------ PolynomialBatch (commit)
 * plonky2::fri::oracle::PolynomialBatch::{from_values, from_coeffs}(polys, rate_bits, blinding=false, cap_height, ..).
 * Input: n_cols columns of 2^log_n elements, column-major.  The batch keeps, in HBM: coefficients, the LDE on the
 * coset g*<w_{2^(log_n+rate_bits)}> in bit-reversed row order (column-major), every Merkle level, and the cap.
 * cap_out (may be NULL): 2^cap_height x 4, host.
 */
int32_t zk_batch_from_values(zk_ctx *ctx, const uint64_t *cols, uint32_t log_n, uint32_t n_cols, uint32_t rate_bits,
                             uint32_t cap_height, zk_batch **out, uint64_t *cap_out);
int32_t zk_batch_from_values_dev(zk_ctx *ctx, const uint64_t *d_cols, uint32_t log_n, uint32_t n_cols,
                                 uint32_t rate_bits, uint32_t cap_height, zk_batch **out, uint64_t *cap_out);
int32_t zk_batch_from_coeffs(zk_ctx *ctx, const uint64_t *cols, uint32_t log_n, uint32_t n_cols, uint32_t rate_bits,
                             uint32_t cap_height, zk_batch **out, uint64_t *cap_out);
int32_t zk_batch_from_coeffs_dev(zk_ctx *ctx, const uint64_t *d_cols, uint32_t log_n, uint32_t n_cols,
                                 uint32_t rate_bits, uint32_t cap_height, zk_batch **out, uint64_t *cap_out);
int32_t zk_batch_free(zk_batch *b);
int32_t zk_batch_info(const zk_batch *b, uint32_t *log_n, uint32_t *n_cols, uint32_t *rate_bits, uint32_t *cap_height);
int32_t zk_batch_get_cap(zk_batch *b, uint64_t *cap_out);
/* PolynomialBatch::get_lde_values(index, step) for each idx[i]: out[i*n_cols .. ] = leaves[bitrev(idx[i]*step)] */
int32_t zk_batch_get_lde_rows(zk_batch *b, const uint32_t *idx, uint32_t n_idx, uint32_t step, uint64_t *out);
/* raw leaf `leaf_idx` (already bit-reversed numbering, as FRI queries use it): n_cols elements */
int32_t zk_batch_get_leaf(zk_batch *b, uint32_t leaf_idx, uint64_t *out);
/* MerkleTree::prove(leaf_idx): (log_n + rate_bits - cap_height) x 4 siblings, leaf level first */
int32_t zk_batch_merkle_prove(zk_batch *b, uint32_t leaf_idx, uint64_t *siblings_out);
/* MerkleTree::digests in upstream's recursive layout: 2*(2^(log_n+rate_bits) - 2^cap_height) x 4 */
int32_t zk_batch_get_digests(zk_batch *b, uint64_t *digests_out);
/* polynomials[col].coeffs, natural order, 2^log_n elements */
int32_t zk_batch_get_coeffs(zk_batch *b, uint32_t col, uint64_t *out);
/* PolynomialCoeffs::to_extension().eval(point) for every column: out = n_cols x 2 */
int32_t zk_batch_eval_at(zk_batch *b, const uint64_t point[2], uint64_t *out);

/* ------------------------------------------------------------------ Challenger (device-resident transcript)
 * plonky2::iop::challenger::Challenger<F, PoseidonHash>.  The sponge state lives in HBM and is advanced by a
 * one-wave kernel, so a whole proof can be enqueued without host round trips; these entry points are the
 * (synchronous) host view of the same object.
 */
int32_t zk_challenger_new(zk_ctx *ctx, zk_challenger **out);
int32_t zk_challenger_free(zk_challenger *ch);
int32_t zk_challenger_observe(zk_challenger *ch, const uint64_t *elems, uint32_t n);
int32_t zk_challenger_get(zk_challenger *ch, uint32_t n, uint64_t *out);
int32_t zk_challenger_compact(zk_challenger *ch, uint64_t out_state[12]);

/* ------------------------------------------------------------------ AIR description (plain data)
 * Stand-in for a starky `Stark::eval_packed_generic` body: a straight-line program over
 * the base field evaluated at every point of the LDE coset.  Operand = kind<<24 | index.
 */
enum {
    ZK_OPND_REG = 0,       /* temporary register */
    ZK_OPND_LOCAL = 1,     /* vars.local_values[i]; i >= n_trace_cols addresses auxiliary column i - n_trace_cols */
    ZK_OPND_NEXT = 2,      /* vars.next_values[i], same indexing */
    ZK_OPND_CONST = 3,     /* air->constants[i] */
    ZK_OPND_PUBLIC = 4,    /* public_inputs[i] */
    ZK_OPND_CHALLENGE = 5  /* lookup/CTL challenges: [beta_0, gamma_0, beta_1, gamma_1, ...] */
};
enum {
    ZK_OP_ADD = 0, ZK_OP_SUB = 1, ZK_OP_MUL = 2,
    /* ConstraintConsumer::{constraint, constraint_transition, constraint_first_row, constraint_last_row}(a) */
    ZK_OP_EMIT = 3, ZK_OP_EMIT_TRANSITION = 4, ZK_OP_EMIT_FIRST = 5, ZK_OP_EMIT_LAST = 6,
    /* gate programs of zk_plonk_circuit only: gate constraint term `dst` += a (see zk_plonk_circuit below) */
    ZK_OP_EMIT_SLOT = 7
};
#define ZK_OPND(kind, idx) (((uint32_t)(kind) << 24) | (uint32_t)(idx))
typedef struct { uint32_t op, dst, a, b; } zk_instr;

/* One CTL-shaped helper column (starky::cross_table_lookup::partial_sums), plain form (see zk_helper_ext for filters and longer column sets):
 *   h[row] = 1/(t[a0] + beta*t[b0] + gamma) + 1/(t[a1] + beta*t[b1] + gamma)        (t = trace columns)
 * For every challenge set j the auxiliary columns are  [h_0 .. h_{H-1}, Z]  with the suffix sum
 *   Z[row] = sum_{s >= row} sum_m h_m[s].                                                            */
typedef struct { uint32_t a0, b0, a1, b1; } zk_helper_desc;
/* Optional refinement of helper m (zk_air.helpers_ext, n_helpers entries, or NULL for plain pairs) -- starky's CTL helper column in full
 * (cross_table_lookup::partial_sums): a sum over at most two (column set, filter) pairs of
 *     filter[row] / combine(cols)[row],   combine(cols) = cols[0] + beta cols[1] + beta^2 cols[2] + .. + gamma.
 * Pair k of helper m has the columns (a_k, b_k) of zk_helper_desc followed by n_extra[k] more (extra[k][..]: at most ZK_CTL_MAX_EXTRA, so
 * column sets of 2 .. 8 columns) and the 0/1 filter column filter[k] (ZK_NO_FILTER: constant 1):
 *     h_m[row] = f_0 / (t[a0] + beta t[b0] + beta^2 t[e0_0] + .. + gamma) + f_1 / (t[a1] + beta t[b1] + beta^2 t[e1_0] + .. + gamma). */
#define ZK_CTL_MAX_EXTRA 6
typedef struct { uint32_t filter[2]; uint32_t n_extra[2]; uint32_t extra[2][ZK_CTL_MAX_EXTRA]; } zk_helper_ext;

/* A logUp lookup argument inside one table (starky::lookup::{Lookup, lookup_helper_columns}): looked columns f_i, each
 * with an optional 0/1 filter column, are checked against a table column t with a multiplicity column m.  For every challenge set j
 * (challenge beta_j) the auxiliary columns are, in upstream's order and BEFORE the CTL-shaped ones:
 *   helper_b[row] = sum over the batch b of (constraint_degree - 1 = 2) looked columns of  filter_i[row] / (f_i[row] + beta_j)
 *   Z[0] = 0,  Z[r+1] = Z[r] + sum_b helper_b[r] - m[r] / (t[r] + beta_j)          (forward prefix sum; it closes, Z[n] = Z[0], iff the
 *                                                                                   multiset of looked values is the table's with m)
 * i.e. ceil(n_cols / 2) + 1 columns per challenge: [helpers .., Z] of challenge 0, then of challenge 1, ...                      */
#define ZK_NO_FILTER 0xFFFFFFFFu
typedef struct { uint32_t col, filter; } zk_lookup_col;     /* trace column looked up; filter trace column or ZK_NO_FILTER (constant 1) */
typedef struct {
    uint32_t n_cols;                 /* looked columns */
    uint32_t table_col, freq_col;    /* t and m: trace columns */
    uint32_t reserved;
    const zk_lookup_col *cols;
} zk_lookup;

typedef struct {
    uint32_t n_trace_cols;
    uint32_t n_public_inputs;
    uint32_t n_helpers;        /* H; CTL-shaped auxiliary columns = num_challenges * (H + 1), or 0 when H == 0 */
    uint32_t n_constants;
    uint32_t n_regs;
    uint32_t n_instrs;
    uint32_t constraint_degree; /* 3 for every zkEVM table -> quotient_degree_factor 2 */
    uint32_t reserved;
    const zk_helper_desc *helpers;
    const uint64_t *constants;
    const zk_instr *instrs;
    const zk_lookup *lookup;   /* optional (NULL: none): adds num_challenges * (ceil(lookup->n_cols / 2) + 1) auxiliary columns in front */
    const zk_helper_ext *helpers_ext;   /* optional (NULL: every helper is two plain, unfiltered pairs) */
} zk_air;
/* number of auxiliary columns of an AIR: lookup columns first, CTL-shaped columns last */
static inline uint32_t zk_air_n_aux(const zk_air *air, uint32_t num_challenges) {
    uint32_t a = air->n_helpers ? num_challenges * (air->n_helpers + 1) : 0;
    if (air->lookup && air->lookup->n_cols) a += num_challenges * ((air->lookup->n_cols + 1) / 2 + 1);
    return a;
}

/* Structural check of a constraint program: every opcode known, destinations are registers below n_regs, LOCAL/NEXT indices
 * below n_trace_cols + auxiliary columns, CONST / PUBLIC / CHALLENGE indices inside their arrays (2 * num_challenges challenge
 * words), helper descriptors name trace columns.  ZK_OK or ZK_ERR_INVALID_ARG.  Every entry point that takes a zk_air runs it
 * first, so a malformed program is refused instead of indexing out of bounds on the host or the GPU.  (Upstream's constraints
 * are Rust code checked by the compiler: /root/reference/ops/src/lib.rs:52 never sees a malformed AIR.) */
int32_t zk_air_validate(const zk_air *air, uint32_t num_challenges);

/* starky::config::StarkConfig; zk_stark_config_standard_fast() == StarkConfig::standard_fast_config() */
typedef struct {
    uint32_t security_bits;        /* 100 */
    uint32_t num_challenges;       /* 2 */
    uint32_t rate_bits;            /* 1 */
    uint32_t cap_height;           /* 4 */
    uint32_t proof_of_work_bits;   /* 16 */
    uint32_t fri_arity_bits;       /* ConstantArityBits(4, 5) */
    uint32_t fri_final_poly_bits;  /* 5 */
    uint32_t num_query_rounds;     /* 84 */
} zk_stark_config;
void zk_stark_config_standard_fast(zk_stark_config *out);
/* the FRI / commitment parameters of plonky2's CircuitConfig::standard_recursion_config() -- rate_bits 3, cap_height 4, PoW 16,
 * ConstantArityBits(4, 5), 28 query rounds, 2 challenges: what the shrink / root / aggregation / block circuits behind
 * /root/reference/ops/src/lib.rs:52,72,95 are proved with; for zk_plonk_setup / zk_prove_plonk / zk_fri_prove_instance */
void zk_stark_config_standard_recursion(zk_stark_config *out);

/* ------------------------------------------------------------------ unit proof
 * One unit of work of the reference's map step (`IndexedStream::from(..).map(&TxProof)`, /root/reference/leader/src/prover.rs:26-27):
 * the STARK proofs of all tables of one transaction on one shared challenger.  Mirrors
 * evm_arithmetization::prover::prove_with_traces -> starky::prover::prove_with_commitment -> plonky2 fri_proof
 * (what generate_txn_proof at /root/reference/ops/src/lib.rs:52 spends its time in before the recursion wrappers).
 * Traces are column-major n_trace_cols x 2^log_n; `trace` is a device pointer when trace_on_device != 0.
 *
 * Proof = canonical little-endian u64 words, fields in upstream order (StarkProof / FriProof):
 *   "ZKGLPRF1", n_tables, then per table:
 *   log_n, n_trace_cols, n_aux_cols, trace_cap[2^h x4], (aux_cap)[2^h x4], quotient_cap[2^h x4],
 *   openings: local[C x2] next[C x2] aux[A x2] aux_next[A x2] (ctl_zs_first[num_challenges]) quotient[Q x2],
 *   fri: n_layers, commit_phase_merkle_caps[n_layers][2^h x4], final_poly_len, final_poly[len x2], pow_witness,
 *        query_round_proofs[num_query_rounds]: per initial oracle (trace,(aux),quotient): leaf[n_cols], siblings[(lgL-h) x4];
 *                                              per layer: evals[2^arity_bits x2], siblings[.. x4]
 * The proof-of-work witness is the SMALLEST valid one (upstream's rayon find_any is non-deterministic).
 */
typedef struct {
    const zk_air *air;
    uint32_t log_n;
    uint32_t trace_on_device;
    const uint64_t *trace;
    const uint64_t *public_inputs;   /* host, air->n_public_inputs elements */
} zk_table;
size_t zk_proof_size_words(const zk_stark_config *cfg, uint32_t n_tables, const zk_table *tables);
/* Parameter window (ZK_ERR_INVALID_ARG outside it; upstream's own limits are wider): 1..4 challenges, rate_bits 1 (constraint degree 3:
 * quotient_degree_factor 2), cap_height <= 8 and <= log_n + rate_bits, FRI arity bits 2..6, log_n <= 26.  A (config, log_n) pair on
 * which upstream's FriReductionStrategy::ConstantArityBits panics (usize underflow of degree_bits + rate_bits - arity_bits, or its
 * assert!(degree_bits >= arity_bits)) is refused the same way; zk_proof_size_words returns 0 for every refused combination. */
int32_t zk_prove_tables(zk_ctx *ctx, const zk_stark_config *cfg, uint32_t n_tables, const zk_table *tables,
                        uint64_t *proof_out, size_t proof_cap_words, size_t *proof_len_words);
/* The same proof, with a notification per table: tables are proved one after another on one challenger, and table t's StarkProof is
 * final long before the last table's -- `on_table(user, t, words, n_words)` is called (on the calling thread, tables in order) as soon as
 * table t's words have reached the host, while the later tables are still being proved on the GPU.  prove_root uses it to start a
 * table's shrink chain (by_table[t].by_stark_size[..].shrink, independent of the other tables) under the STARK proofs of the tables
 * that follow; upstream's loop (prove(), then the chains) computes the same values in sequence.  `words` points into the context's
 * pinned download buffer: copy before returning.  The device-side error word is checked after the LAST table as always: a callback may
 * see the words of a proof that the call then refuses (a consumer must drop what it derived when the call fails). */
typedef void (*zk_table_ready_fn)(void *user, uint32_t table, const uint64_t *words, size_t n_words);
int32_t zk_prove_tables_notify(zk_ctx *ctx, const zk_stark_config *cfg, uint32_t n_tables, const zk_table *tables,
                               uint64_t *proof_out, size_t proof_cap_words, size_t *proof_len_words, zk_table_ready_fn on_table, void *user);
/* Prover state: what the reference builds once at start-up (common/src/prover_state/mod.rs:59-95, used through
 * ops::p_state, ops/src/lib.rs:16-18).  Here: the constraint programs of the given tables compiled to gfx950 code (hiprtc) and
 * cached for the process (and in the prover-state directory, zk_prover_state_cache_dir); without this call proofs evaluate
 * constraints with the bytecode interpreter unless the AIR's kernel was built with the library.  Every table given here is compiled,
 * whatever its size (programs of thousands of instructions become a sequence of call-free kernels: a minute or two at most), and a compiled kernel is used
 * at any table size afterwards.  Call from the thread that created the context, before handing units to worker threads.  `trace` of
 * the tables is not read. */
int32_t zk_prover_state_prepare(zk_ctx *ctx, const zk_stark_config *cfg, uint32_t n_tables, const zk_table *tables);
/* Persistence of the compiled part of the prover state -- the counterpart of the reference's `./prover_state_<digest>` file
 * (/root/reference/common/src/prover_state/persistence.rs:40-88, name from get_configuration_digest,
 * /root/reference/common/src/prover_state/circuit.rs:177-197).  With a cache directory set (this call, or the environment variable
 * ZK_PROVER_STATE_DIR read at first use; NULL or "" switches persistence off), every constraint kernel that hiprtc compiles is
 * written to  <dir>/prover_state_<digest>.gfx950.co  where <digest> names the AIR (program, column counts, challenges,
 * rate_bits), and a later process start loads the code object from there instead of compiling.  A file that is missing, truncated,
 * for another program or fails its checksum is ignored and rebuilt (the reference warns and regenerates likewise); the file's key holds
 * the GPU architecture and the hiprtc version, and an image that does not load on this device is deleted and rebuilt.  The directory is
 * TRUSTED: its files are GPU code that this process will run (the checksum guards against damage, not against an attacker who can write
 * there) -- give it the permissions of the binary itself.  Process-wide.
 * zk_prover_state_stats: kernels compiled by hiprtc / loaded from disk so far in this process. */
int32_t zk_prover_state_cache_dir(const char *dir);
int32_t zk_prover_state_stats(uint64_t *compiled, uint64_t *loaded_from_disk);
/* The same files written WITHOUT a GPU or a context (packaging / build step): the constraint kernels of the given AIRs (only `air` of every
 * table is read) are compiled by hiprtc for gfx950 and stored in the cache directory under the key a process on the target would look up.
 * gcn_arch_name: what hipGetDeviceProperties reports there ("gfx950:sramecc+:xnack-" on MI355X).  n_built / n_present (optional): files
 * written / found already there.  Kernels that were built with the library are skipped. */
/* CPU stand-in for the witness generation of a recursion proof (host only, no context): recomputes, from their input cells, the cells that
 * plonky2's PoseidonGenerator and ArithmeticBaseGenerator produce, for the given rows of a column-major num_wires x n witness in host
 * memory -- the cost upstream pays per proof inside generate_txn_proof (/root/reference/ops/src/lib.rs:52).  Returns the number of cells
 * whose value changed (0 for a witness that satisfied its circuit) or a negative ZK_ERR_*. */
int64_t zk_host_plonk_witness_recompute(uint64_t *wires, uint64_t n, uint32_t num_wires, const uint32_t *poseidon_rows, uint64_t n_poseidon,
                                        const uint32_t *arith_rows, const uint64_t *arith_c0, const uint64_t *arith_c1, uint64_t n_arith);
int32_t zk_prover_state_build_offline(const zk_stark_config *cfg, uint32_t n_tables, const zk_table *tables, const char *gcn_arch_name,
                                      uint64_t *n_built, uint64_t *n_present);
/* Build-time half of the same idea (upstream's constraints are Rust code monomorphised when the prover is built): the HIP source of
 * the constraint kernel of one AIR and the hash that names it.  Host only (no context, no GPU); tools/gen_aot_quotient.py writes
 * the kernels of the AIRs known at build time into csrc/prover_quotient_aot.hip, which is compiled with the library.
 * buf may be NULL to query the length. */
int32_t zk_quotient_kernel_source(const zk_air *air, uint32_t num_challenges, uint32_t rate_bits, char *buf, size_t cap,
                                  size_t *len, uint64_t *key_hash);
/* what the last zk_quotient_kernel_source call of this thread generated: the number of kernels (a program of 640 instructions or more becomes
 * a SEQUENCE of call-free kernels <name>_0, <name>_1, ..) and the state words per point that one half of the sequence's hand-over buffer holds */
int32_t zk_quotient_kernel_source_plan(uint32_t *n_kernels, uint32_t *n_state);
/* ------------------------------------------------------------------ the pieces of a unit proof, one upstream item each
 * starky::lookup::lookup_helper_columns + cross_table_lookup::partial_sums: auxiliary columns of one table for the
 * challenges [beta_0, gamma_0, beta_1, gamma_1, ..] (host in, host out; aux_out = n_aux_cols x 2^log_n column-major,
 * helper columns of every challenge first, then one Z column per challenge).  ZK_ERR_ZERO_INVERSE if a denominator is 0. */
int32_t zk_aux_columns(zk_ctx *ctx, const zk_air *air, uint32_t num_challenges, const uint64_t *challenges,
                       const uint64_t *trace, uint32_t log_n, uint64_t *aux_out);
/* starky::prover::compute_quotient_polys: (num_challenges << rate_bits) quotient polynomials of 2^log_n coefficients each,
 * natural coefficient order, column-major (host out).  aux may be NULL.  alphas: num_challenges base-field elements. */
int32_t zk_quotient(zk_ctx *ctx, const zk_stark_config *cfg, const zk_air *air, zk_batch *trace, zk_batch *aux,
                    const uint64_t *public_inputs, const uint64_t *challenges, const uint64_t *alphas,
                    uint64_t *quotient_coeffs_out);
/* plonky2::fri::prover::fri_proof_of_work: the SMALLEST w such that permute(state with state[pos] = w)[7] has at least
 * min_zeros leading zero bits (state = the duplex-intermediate sponge state, pos = number of pending inputs). */
int32_t zk_pow_grind(zk_ctx *ctx, const uint64_t state[12], uint32_t pos, uint32_t min_zeros, uint64_t *witness_out);
/* plonky2::fri::oracle::PolynomialBatch::prove_openings -> fri_proof on the oracles (trace, aux or NULL, quotient) at zeta:
 * advances `ch` exactly as upstream does (alpha, per-layer caps and betas, final polynomial, PoW, query indices) and writes
 * the FriProof words (layout: the "fri:" part of the proof description above). */
size_t zk_fri_proof_size_words(const zk_stark_config *cfg, uint32_t log_n, uint32_t n_trace_cols, uint32_t n_aux_cols);
int32_t zk_fri_prove_openings(zk_ctx *ctx, const zk_stark_config *cfg, zk_batch *trace, zk_batch *aux, zk_batch *quot,
                              zk_challenger *ch, const uint64_t zeta[2], uint64_t *fri_proof_out, size_t cap_words,
                              size_t *len_words);

/* plonky2::fri::structure::{FriPolynomialInfo, FriBatchInfo, FriInstanceInfo} and
 * plonky2::fri::oracle::PolynomialBatch::prove_openings(instance, oracles, challenger, fri_params) for ANY instance (the
 * entry point the Plonk recursion proofs of SURVEY.md 8f-2 need: 4 oracles, rate_bits 3, batches at zeta and g*zeta):
 * every batch opens the listed polynomials (oracle_index into `oracles`, polynomial_index into that oracle's columns) at
 * `point` (an extension element).  All oracles share log_n and were committed with cfg->rate_bits / cfg->cap_height.
 * n_oracles <= 8, n_batches <= 4, every batch non-empty.  Advances `ch` as upstream does (alpha, per-layer caps and betas,
 * final polynomial, PoW witness, query indices) and writes the FriProof words: n_layers, caps, final_poly_len, final_poly,
 * pow_witness, then per query round {per oracle: leaf row, Merkle path} {per layer: 2^arity_bits ext evals, path}.
 * A base-field point that lies on the LDE coset g*H (where the pointwise quotient is undefined; upstream's synthetic
 * division would still succeed there) is refused with ZK_ERR_ZETA_IN_SUBGROUP. */
typedef struct zk_fri_poly { uint32_t oracle_index, polynomial_index; } zk_fri_poly;
typedef struct zk_fri_batch { uint64_t point[2]; uint32_t n_polys; uint32_t reserved; const zk_fri_poly *polys; } zk_fri_batch;
size_t zk_fri_instance_proof_size_words(const zk_stark_config *cfg, uint32_t log_n, uint32_t n_oracles, const uint32_t *oracle_n_cols);
int32_t zk_fri_prove_instance(zk_ctx *ctx, const zk_stark_config *cfg, uint32_t n_oracles, zk_batch *const *oracles,
                              uint32_t n_batches, const zk_fri_batch *batches, zk_challenger *ch, uint64_t *fri_proof_out,
                              size_t cap_words, size_t *len_words);

/* The same FRI prover piece by piece (SURVEY.md 8b "finer-grained"), one upstream item per call, synchronous:
 *   zk_fri_combine            the batch combination of PolynomialBatch::prove_openings for a given alpha: final_values_out = the
 *                             2^(log_n + rate_bits) extension values of sum_b alpha^(..) (P_b - P_b(z_b)) / (X - z_b) on the LDE coset,
 *                             in leaf (bit-reversed) order, [re, im] interleaved  (upstream: lde_final_values after reverse_index_bits)
 *   zk_fri_commit_begin       takes such values (host) and holds the commit-phase state on the device
 *   zk_fri_fold_commit_layer  one iteration of fri_committed_trees: Merkle tree over the current values (leaves of 2^arity_bits
 *                             extension values), cap observed by `ch`, beta drawn from `ch`, coefficients folded, values of the
 *                             next layer = coset_fft(shift^arity).  cap_out: 2^cap_height x 4 words
 *   zk_fri_commit_finish      final polynomial (len >> rate_bits coefficients), observed by `ch`; out: 2 * len_ext words
 *   zk_fri_query              fri_prover_query_round(x_index): per oracle (leaf row, Merkle path), per layer (2^arity_bits
 *                             extension evaluations, Merkle path) -- the words of one query round of the FriProof layout above
 * The proof of work between finish and the queries is zk_pow_grind on the challenger's state. */
typedef struct zk_fri_commit zk_fri_commit;
int32_t zk_fri_combine(zk_ctx *ctx, const zk_stark_config *cfg, uint32_t n_oracles, zk_batch *const *oracles, uint32_t n_batches,
                       const zk_fri_batch *batches, const uint64_t alpha[2], uint64_t *final_values_out);
int32_t zk_fri_commit_begin(zk_ctx *ctx, const zk_stark_config *cfg, uint32_t log_n, const uint64_t *final_values, zk_fri_commit **out);
int32_t zk_fri_commit_layers(const zk_fri_commit *s, uint32_t *n_layers, uint32_t *layers_done);
int32_t zk_fri_fold_commit_layer(zk_fri_commit *s, zk_challenger *ch, uint64_t *cap_out, uint64_t beta_out[2]);
int32_t zk_fri_commit_finish(zk_fri_commit *s, zk_challenger *ch, uint64_t *final_poly_out, uint32_t *len_ext);
size_t zk_fri_query_size_words(const zk_fri_commit *s, uint32_t n_oracles, zk_batch *const *oracles);
int32_t zk_fri_query(zk_fri_commit *s, uint32_t n_oracles, zk_batch *const *oracles, uint32_t x_index, uint64_t *out, size_t cap_words,
                     size_t *len_words);
int32_t zk_fri_commit_free(zk_fri_commit *s);

/* ------------------------------------------------------------------ Plonk-shaped proof
 * What generate_txn_proof / generate_agg_proof / generate_block_proof (/root/reference/ops/src/lib.rs:52,72,95) spend their time in
 * after the STARK proofs: plonky2::plonk::prover::prove over a circuit built with CircuitConfig::standard_recursion_config()
 * (135 wires, 80 routed, quotient degree factor 8, FRI rate_bits 3 / cap 4 / 28 queries).  The real recursion circuits (gate set,
 * recursive verifier gadgets, witness generators) live in crates that are absent from the reference tree, so -- as for the tables'
 * AIRs -- a circuit is plain data here: wire count, routed-wire permutation (sigma values), constant / selector columns and a
 * straight-line gate-constraint program.  The prover work is upstream's, step by step (plonk/prover.rs prove_with_partition_witness):
 *   wires commit -> observe circuit digest, public-input hash, wires cap -> betas, gammas -> partial products and Z of the
 *   routed-wire permutation (chunks of quotient_degree_factor) -> commit [Z.., partial products..] -> alphas -> vanishing polynomial
 *   on the 8n coset = [L_0 (Z - 1) .., partial-product checks .., gate constraints ..] reduced with powers of alpha, / Z_H ->
 *   quotient chunks commit -> zeta -> OpeningSet -> PolynomialBatch::prove_openings over the four oracles
 *   (constants_sigmas, wires, zs_partial_products, quotient; batches at zeta and at g*zeta for the Zs).
 * Gate program: zk_air with n_trace_cols = num_wires + num_constants (LOCAL index i < num_wires = wire i, else constant column
 * i - num_wires), ops ADD / SUB / MUL / EMIT / EMIT_SLOT only, no NEXT / CHALLENGE operands, n_helpers = 0; PUBLIC index i < 4 = word i
 * of public_inputs_hash = hash_no_pad(public_inputs) -- what upstream's PublicInputGate constrains its wires against.
 * Constraint terms: upstream evaluates EVERY gate of the circuit at every point, each multiplied by its selector filter, and ADDS
 * the gates' constraints position by position (plonk::vanishing_poly::evaluate_gate_constraints_base_batch: constraints_batch[j] +=
 * gate_constraints[j]), so the vanishing polynomial has max-over-gates(num_constraints) gate terms.  ZK_OP_EMIT_SLOT (dst = j) adds
 * its operand to gate term j; ZK_OP_EMIT adds to term (number of ZK_OP_EMIT instructions before it).  Number of gate terms =
 * 1 + the highest term index written (at most 4096).
 *
 * Proof words: "ZKGLPLK1", degree_bits, num_wires, num_routed, num_constants, wires_cap, zs_partial_products_cap, quotient_cap,
 *   openings: constants[NC x2] sigmas[R x2] wires[W x2] zs[nch x2] zs_next[nch x2] partial_products[nch*npp x2]
 *             quotient[nch*qdf x2],  then the FriProof words in the layout of zk_fri_prove_instance. */
typedef struct {
    uint32_t degree_bits;             /* n = 2^degree_bits rows */
    uint32_t num_wires;               /* 135 */
    uint32_t num_routed_wires;        /* 80; a multiple of quotient_degree_factor (any number of chunks) */
    uint32_t num_constants;           /* constant + selector columns */
    uint32_t quotient_degree_factor;  /* 8: chunk size of the partial products, number of quotient chunks per challenge */
    uint32_t n_domain_separator;      /* words of CircuitBuilder::set_domain_separator (0: none, as in every circuit of the reference) */
    const zk_air *gates;              /* the gate-constraint program */
    const uint64_t *constants;        /* num_constants x n, column-major: values on the subgroup H */
    const uint64_t *sigmas;           /* num_routed_wires x n: sigma_j(w^r) = k_j' * w^r' for the cell (j', r') that (j, r) maps to, k_j = g^j */
    const uint64_t *domain_separator; /* n_domain_separator words (may be NULL when 0) */
} zk_plonk_circuit;
typedef struct zk_plonk_state zk_plonk_state;   /* prover data of one circuit: constants_sigmas commitment, digest; device resident */
/* CircuitBuilder::build's prover-side part: commit constants | sigmas (rate_bits, cap_height of cfg), derive the circuit digest
 * = hash_no_pad(constants_sigmas_cap | hash_pad(domain_separator) | degree_bits); hash_pad = pad10*1 to a multiple of the rate,
 * so the empty separator hashes [1,0,0,0,0,0,0,1].  cfg: zk_stark_config with the recursion parameters (rate_bits 3, 28 query rounds ..). */
/* Hasher::hash_pad (the domain separator inside the circuit digest) pads  x | 1 | 0.. | 1  to a multiple of this many words.  plonky2 0.2.2 is
 * not in the reference tree; RATE (8) is what both this library and the oracle restate, from recall -- WIDTH (12) would be the other reading.
 * One constant here, one in oracle/oracle.h (ORC_HASH_PAD_MULTIPLE), one in oracle/pyspec.py (HASH_PAD_MULTIPLE): a maintainer with the crate
 * at hand settles it in three lines. */
#define ZK_HASH_PAD_MULTIPLE 8u
int32_t zk_plonk_setup(zk_ctx *ctx, const zk_stark_config *cfg, const zk_plonk_circuit *circuit, zk_plonk_state **out);
int32_t zk_plonk_state_free(zk_plonk_state *st);
int32_t zk_plonk_circuit_digest(zk_plonk_state *st, uint64_t digest_out[4]);
size_t zk_plonk_proof_size_words(const zk_stark_config *cfg, const zk_plonk_circuit *circuit);
/* wires: num_wires x n column-major witness (host, or device when wires_on_device != 0); public_inputs: hashed into the transcript */
int32_t zk_prove_plonk(zk_ctx *ctx, const zk_stark_config *cfg, zk_plonk_state *st, const uint64_t *wires, uint32_t wires_on_device,
                       const uint64_t *public_inputs, uint32_t n_public_inputs, uint64_t *proof_out, size_t proof_cap_words,
                       size_t *proof_len_words);

/* Build-time half of the circuits' prover state, as zk_quotient_kernel_source for the tables: the HIP source of the vanishing-polynomial
 * kernel with this gate program as straight-line code, and the hash that names it.  Host only; tools/gen_aot_quotient.py writes the
 * kernel of the recursion gate set into csrc/prover_plonk_aot.hip, which is compiled with the library; zk_plonk_setup finds it there,
 * compiles other programs of >= 256 instructions with hiprtc (kept in the prover-state directory like the tables' kernels) and leaves
 * small ones to the interpreter kernel (ZK_PLONK_JIT=0: always the interpreter).  A long program comes out as a SEQUENCE of *n_kernels kernels
 * (<name>_0 ..), launched back to back, that hand *n_state u64 per coset point to each other through a state buffer.  buf may be NULL. */
int32_t zk_plonk_gates_kernel_source(const zk_air *gates, uint32_t num_wires, uint32_t num_constants, uint32_t num_challenges, uint32_t rate_bits,
                                     char *buf, size_t cap, size_t *len, uint64_t *key_hash, uint32_t *n_kernels, uint32_t *n_state);

/* Synthetic satisfying witness for eth_tx_proof_amd.air.cpu_like_air, generated in HBM;
 * public_inputs_out (host, 3 words, may be NULL) = first-row a0, b0 and last-row a0. */
int32_t zk_synth_trace_dev(zk_ctx *ctx, uint32_t n_cols, uint32_t log_n, uint64_t seed, uint64_t *d_trace,
                           uint64_t *public_inputs_out);

#ifdef __cplusplus
}
#endif
#endif /* ZKGL_H */
