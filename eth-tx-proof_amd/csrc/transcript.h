// Device-resident challenger: data structures shared by transcript.hip and the prover.
#pragma once
#include "zk_internal.h"

#include "transcript_ops.h"

int32_t zk_transcript_run(zk_ctx *ctx, DevChallenger *c, const TOpList &ops);
// fri_proof_of_work with the smallest accepted witness; d_pow_state: 14 words of scratch; writes d_witness_out[0]
int32_t zk_pow_grind_dev(zk_ctx *ctx, DevChallenger *c, u32 min_zeros, u64 *d_pow_state, u64 *d_witness_out);
// the search alone on a prepared 14-word state (12 state words, witness position, best-so-far = ~0)
int32_t zk_pow_search_dev(zk_ctx *ctx, u32 min_zeros, u64 *d_pow_state);
struct zk_challenger { zk_ctx *ctx; DevChallenger *d; u64 *scratch; size_t scratch_words; };
