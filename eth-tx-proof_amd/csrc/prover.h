// Internal interfaces between the prover translation units (prover_*.hip).
#pragma once
#include <vector>
#include "zk_internal.h"
#include "transcript.h"

// device copy of an AIR (program, constants, helper descriptors live in HBM)
struct DevAir {
    const zk_air *host;
    zk_instr *d_instrs;
    u64 *d_constants;
    zk_helper_desc *d_helpers;
    zk_helper_ext *d_helpers_ext;      // nullptr: plain pairs
    u32 n_aux(u32 nch) const { return zk_air_n_aux(host, nch); }
};

// prover_aux.hip
int32_t zk_aux_columns_dev(zk_ctx *ctx, const zk_air *air, const zk_helper_desc *d_helpers, const zk_helper_ext *d_helpers_ext, u32 nch, const u64 *d_chal,
                           const u64 *d_trace, u32 log_n, u64 *d_aux);

// prover_quotient.hip: compute_quotient_polys -> d_qcoeffs: (nch << rate_bits) columns of n coefficients, natural order
int32_t zk_quotient_dev(zk_ctx *ctx, const DevAir &air, const zk_stark_config &cfg, const zk_batch *trace, const zk_batch *aux,
                        const u64 *d_public, const u64 *d_chal, const u64 *d_alphas, u64 *d_qcoeffs);

// prover_quotient_jit.hip: the constraint program compiled to gfx950 code (hiprtc); same results as the interpreter
struct QJitParams { const u64 *trace_lde, *aux_lde, *pub, *chal, *alphas, *lag, *xA, *xB; u64 *out; u64 L; u64 last_inv; u64 zh_inv[8]; u32 lgL; u32 force; u64 *state; u32 redo_cap; u32 pad; };
#define ZK_JIT_UNAVAILABLE 1        // internal: the generated kernel cannot be used (it would need scratch); not an error
bool zk_quotient_jit_enabled(const zk_ctx *ctx, size_t L);
int32_t zk_quotient_jit_launch(zk_ctx *ctx, const zk_air *air, u32 C, u32 A, u32 nch, u32 rate_bits, QJitParams &p);

// prover_fri.hip: PolynomialBatch::prove_openings + fri_proof, everything written into the device proof buffer
struct FriLayout {           // offsets (in words) into the proof buffer, per table
    size_t o_local, o_next, o_aux, o_auxn, o_zs, o_q;     // openings
    size_t o_nl, o_layer_caps, o_flen, o_final, o_witness, o_queries;
    size_t per_query;
};
int32_t zk_fri_prove_dev(zk_ctx *ctx, const zk_stark_config &cfg, DevChallenger *ch, const zk_batch *trace, const zk_batch *aux,
                         const zk_batch *quot, const u64 *d_zeta /* zeta, g*zeta */, u64 *d_proof, const FriLayout &lay);

// fri_proof on final values F (2 columns of L, leaf order); only the o_nl..per_query fields of `lay` are used
int32_t zk_fri_proof_dev(zk_ctx *ctx, const zk_stark_config &cfg, DevChallenger *ch, u32 log_n, u64 *F, const zk_batch *const *oracles,
                         u32 n_oracles, u64 *d_proof, const FriLayout &lay);
// prove_openings for any FriInstanceInfo: polynomial lists on the host; a batch's point either on the host (canonical, d_point ==
// nullptr) or already in HBM (d_point: 2 words, e.g. a challenge the transcript kernel just wrote -- no download in the middle of a proof)
struct FriBatchDesc { u64 point[2]; u32 n_polys; const zk_fri_poly *polys; const u64 *d_point = nullptr; };
int32_t zk_fri_prove_instance_dev(zk_ctx *ctx, const zk_stark_config &cfg, DevChallenger *ch, const zk_batch *const *oracles, u32 n_oracles,
                                  const FriBatchDesc *batches, u32 n_batches, u64 *d_proof, const FriLayout &lay);

// prover_plonk.hip / prover_quotient_jit.hip: parameters of the Plonk vanishing-polynomial kernel (interpreter and compiled gate programs)
struct PlkQuotParams {
    const u64 *w_lde, *z_lde, *cs_lde;     // column-major over L, leaf order
    const u64 *l0;                         // L_0 on the coset, leaf order
    const u64 *xA, *xB;                    // x_i = g w_L^i
    const u64 *kis, *bg;
    const u64 *apow;                       // [challenge][term]: alpha_j^t for every term t of the vanishing polynomial
    const zk_instr *instrs; const u64 *gconst;
    const u64 *pih;                        // public_inputs_hash (4 words): the PUBLIC operand of the gate program (PublicInputGate)
    u64 zh_inv[8];
    size_t L; u32 lgL, W, NC, R, NZ, qdf, nchunks, nch, n_instrs, n_regs, NT;
    u64 *out;                              // [challenge][leaf position]
    u64 *state;                            // compiled gate programs cut into several kernels: [slot][leaf position] between them
};
// the gate program of a circuit compiled to gfx950 code (build-time kernel, prover-state file, or hiprtc now); *fn_out == nullptr:
// none available (small program, ZK_PLONK_JIT=0) -> the interpreter.  is_module: launch with hipModuleLaunchKernel, else hipLaunchKernel
// (a long program is a SEQUENCE of kernels launched back to back; n_state = u64 slots per coset point of the state buffer between them)
int32_t zk_plonk_gates_kernel(zk_ctx *ctx, const zk_air *gates, u32 W, u32 NC, u32 nch, u32 rate_bits, std::vector<const void *> &fns, int *is_module, u32 *n_state);

// FriReductionStrategy::ConstantArityBits(arity_bits, final_poly_bits) (plonky2 0.2.2 fri/reduction_strategies.rs):
//   while degree_bits > final_poly_bits && degree_bits + rate_bits - arity_bits >= cap_height { push; assert!(degree_bits >= arity_bits); .. }
// upstream computes in usize: when degree_bits + rate_bits < arity_bits the subtraction underflows and, like a failing assert, PANICS.
// Returns false for exactly those (config, size) pairs -- the entry points turn that into ZK_ERR_INVALID_ARG -- and never loops.
static inline bool zk_fri_arities_checked(const zk_stark_config &cfg, u32 degree_bits, u32 *out, u32 *n_out) {
    u32 d = degree_bits, k = 0;
    bool ok = cfg.fri_arity_bits != 0;
    while (ok && d > cfg.fri_final_poly_bits) {
        if (d + cfg.rate_bits < cfg.fri_arity_bits) { ok = false; break; }
        if (d + cfg.rate_bits - cfg.fri_arity_bits < cfg.cap_height) break;
        if (d < cfg.fri_arity_bits || k >= 64) { ok = false; break; }
        out[k++] = cfg.fri_arity_bits; d -= cfg.fri_arity_bits;
    }
    *n_out = k;
    return ok;
}
static inline void zk_fri_arities(const zk_stark_config &cfg, u32 degree_bits, u32 *out, u32 *n_out) { (void)zk_fri_arities_checked(cfg, degree_bits, out, n_out); }
static inline bool zk_fri_reduction_ok(const zk_stark_config &cfg, u32 degree_bits) { u32 ar[64], nl; return zk_fri_arities_checked(cfg, degree_bits, ar, &nl); }

// prover_quotient.hip: L_first | L_last (two columns of L = n << rate_bits) on the coset g*H_L, leaf order; cached per (log_n, rate_bits)
int32_t zk_lagrange_table(zk_ctx *ctx, u32 log_n, u32 rate_bits, u64 **out);
// prover_fri.hip: the combination half of prove_openings (alpha on the device) -> final values F (re | im columns of L, leaf order)
int32_t zk_fri_final_values_dev(zk_ctx *ctx, const zk_stark_config &cfg, const u64 *d_alpha, const zk_batch *const *oracles, u32 n_oracles,
                                const FriBatchDesc *batches, u32 nb, u64 *F);
