// The op list of the Fiat-Shamir transcript (shared by the device kernel, transcript.hip, and the host-side challenger,
// transcript_host_core.h).  Plain C++: no HIP types, so the host side's protocol can be built and tested without a GPU.
#pragma once
#include <stdint.h>
typedef uint64_t u64;
typedef uint32_t u32;

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
