// Device-resident challenger: data structures shared by transcript.hip and the prover.
#pragma once
#include "zk_internal.h"

struct DevChallenger {
    u64 state[12];
    u64 inbuf[8];
    u64 outbuf[8];
    u32 n_in, n_out;
};

enum {
    TOP_INIT = 0,
    TOP_OBSERVE,              // observe src[0..n)
    TOP_OBSERVE_BASE_AS_EXT,  // observe (src[i], 0) pairs: base-field values lifted to the extension
    TOP_GET,                  // dst[0..n) = get_challenge() x n
    TOP_COMPACT,              // Challenger::compact(); dst (optional) receives the 12-word state
    TOP_POW_SETUP,            // dst[0..12) = duplex-intermediate state, dst[12] = witness position, dst[13] = ~0
    TOP_POW_FINISH,           // observe witness src[13], get response, check >= n leading zeros, dst[0] = witness
    TOP_GET_INDICES           // dst[i] = get_challenge() mod 2^aux
};
struct TOp { int kind; u32 n; u32 aux; const u64 *src; u64 *dst; };
struct TOpList { int n; TOp op[8]; };

int32_t zk_transcript_run(zk_ctx *ctx, DevChallenger *c, const TOpList &ops);
// fri_proof_of_work with the smallest accepted witness; d_pow_state: 14 words of scratch; writes d_witness_out[0]
int32_t zk_pow_grind_dev(zk_ctx *ctx, DevChallenger *c, u32 min_zeros, u64 *d_pow_state, u64 *d_witness_out);
// the search alone on a prepared 14-word state (12 state words, witness position, best-so-far = ~0)
int32_t zk_pow_search_dev(zk_ctx *ctx, u32 min_zeros, u64 *d_pow_state);
struct zk_challenger { zk_ctx *ctx; DevChallenger *d; u64 *scratch; size_t scratch_words; };
