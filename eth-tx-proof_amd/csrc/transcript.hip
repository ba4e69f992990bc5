// Device-resident Fiat-Shamir transcript.
// Mirrors plonky2 0.2.2 iop::challenger::Challenger<F, PoseidonHash> (crate pinned at
// /root/reference/Cargo.lock:3441-3443; semantics: SURVEY.md A6): overwrite-mode duplex sponge, rate 8, challenges
// popped from the back of the output buffer.
// The sponge state lives in HBM and is advanced by a tiny kernel on the ctx stream, so caps, openings and challenges
// never cross PCIe while a proof is being produced; every consumer kernel reads its challenges from device memory.
#include "zk_internal.h"
#include "poseidon_coop.cuh"
#include "poseidon_wave.cuh"
#include "transcript.h"

// The kernel runs as ONE wave and keeps the whole challenger in registers (round 4): state word l in lane l (l < 12), pending input l
// in lane l (l < 8), the two counters wave-uniform.  Upstream's output buffer is the first eight state words as of the last permutation
// -- between two permutations nothing writes the state (inputs wait in the input buffer: overwrite-mode duplex) -- so a challenge is a
// v_readlane of the state, an observed value a v_writelane-like select into the input register, and nothing goes through LDS or a
// barrier on the serial chain.  The permutation is the wave-wide one (poseidon_wave.cuh): the whole wave works on ONE permutation, its
// latency is all that counts.  (bit 2 of ZK_POSEIDON_MFMA selects the older 16-lane version, poseidon_coop.cuh, for comparison and
// for the tests.)
// The op list runs as a state machine whose loop body has exactly ONE permutation site: every step is "things to do before a possible
// duplexing -> duplexing if needed -> things to do after it".  Before, the permutation was a __noinline__ function with several call
// sites, and a call costs a stack frame: 32 bytes of scratch per lane in a kernel that every stream of every proof launches hundreds
// of times (VERDICT r3 / DESIGN.md section 9: no shipped kernel uses scratch).
__device__ __forceinline__ u64 ch_rdlane(u64 v, int l) {           // l wave-uniform
    const u32 lo = (u32)__builtin_amdgcn_readlane((int)(u32)v, l), hi = (u32)__builtin_amdgcn_readlane((int)(u32)(v >> 32), l);
    return ((u64)hi << 32) | lo;
}

__global__ __launch_bounds__(64) void transcript_kernel(DevChallenger *gc, TOpList ops, u64 *err, int wave, u64 force_redo) {
    __shared__ u64 wtab[psdw::TAB_WORDS];
    const int lane = threadIdx.x;
    if (wave) psdw::fill<64>(wtab);
    u64 st = lane < 12 ? gc->state[lane] : 0;
    u64 ib = lane < 8 ? gc->inbuf[lane] : 0;
    u32 n_in = gc->n_in, n_out = gc->n_out;
    __syncthreads();
    int k = 0;
    u32 i = 0;                      // progress inside op k
    u64 pref = 0; u32 pref_at = ~0u;    // OBSERVE: src[pref_at + lane] for lane < 8, loaded one permutation ahead
    while (k < ops.n) {
        const TOp o = ops.op[k];
        bool need = false;          // duplexing in this step
        int post = 0;               // 1: pop a challenge -> dst[i - 1]   2: ... masked to aux bits   3: compact   4: PoW response
        switch (o.kind) {
        case TOP_INIT:
            st = 0; n_in = 0; n_out = 0; k++; i = 0;
            continue;
        case TOP_OBSERVE:
        case TOP_OBSERVE_BASE_AS_EXT: {
            const u32 total = o.kind == TOP_OBSERVE ? o.n : 2 * o.n;
            if (i >= total) { k++; i = 0; pref_at = ~0u; continue; }
            if (o.kind == TOP_OBSERVE && n_in == 0 && i + 8 <= total) {
                // a whole rate-sized chunk: one coalesced load (issued before the previous permutation when the chunks follow each
                // other), one permutation per 8 elements -- wide opening sets are a serial chain of permutations
                u64 v = pref;
                if (pref_at != i) v = lane < 8 ? o.src[i + lane] : 0;
                ib = gl_canon(v);
                n_in = 8; i += 8;
                if (i + 8 <= total) { pref = lane < 8 ? o.src[i + lane] : 0; pref_at = i; }
            } else {
                const u64 x = o.kind == TOP_OBSERVE ? gl_canon(o.src[i]) : ((i & 1) ? 0 : gl_canon(o.src[i >> 1]));
                if ((u32)lane == n_in) ib = x;
                n_in++; i++;
            }
            n_out = 0;
            need = n_in == 8;
            break;
        }
        case TOP_GET:
        case TOP_GET_INDICES:
            if (i >= o.n) { k++; i = 0; continue; }
            need = n_in != 0 || n_out == 0;
            post = o.kind == TOP_GET ? 1 : 2; i++;
            break;
        case TOP_COMPACT:
            need = n_in != 0; post = 3; k++; i = 0;
            break;
        case TOP_POW_SETUP:
            // fri_proof_of_work: duplex_intermediate_state = sponge state with pending inputs written in; dst[12] = witness position
            if (lane < 12) o.dst[lane] = (u32)lane < n_in ? ib : st;
            if (lane == 0) { o.dst[12] = n_in; o.dst[13] = ~0ULL; }    // dst[13] = best witness so far
            k++; i = 0;
            continue;
        case TOP_POW_FINISH:
            if (i == 0) {           // observe the witness
                u64 w = __hip_atomic_load((u64 *)&o.src[13], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (w == ~0ULL) { if (lane == 0) atomicCAS((unsigned long long *)err, 0ULL, (unsigned long long)(uint32_t)ZK_ERR_POW); w = 0; }
                if (lane == 0) o.dst[0] = w;
                if ((u32)lane == n_in) ib = gl_canon(w);
                n_in++; n_out = 0; need = n_in == 8; i = 1;
            } else {                // the response
                need = n_in != 0 || n_out == 0; post = 4; k++; i = 0;
            }
            break;
        default:
            k++; i = 0;
            continue;
        }
        if (need) {                 // Challenger::duplexing -- the ONE permutation site of the kernel
            u64 s = (u32)lane < n_in ? ib : st;
            if (wave) s = psdw::permute(s, lane, wtab, force_redo);
            else {
                const u32 lo = __shfl((u32)s, lane & 15, 64), hi = __shfl((u32)(s >> 32), lane & 15, 64);
                s = psdc::permute16(((u64)hi << 32) | lo, lane & 15, lane & ~15);
            }
            st = lane < 12 ? s : 0;
            n_in = 0; n_out = 8;
        }
        if (post == 1 || post == 2 || post == 4) {
            const u64 v = ch_rdlane(st, (int)n_out - 1);
            n_out--;
            if (post == 4) { if (lane == 0 && (u32)__clzll((long long)v) < o.n) atomicCAS((unsigned long long *)err, 0ULL, (unsigned long long)(uint32_t)ZK_ERR_POW); }
            else if (lane == 0) o.dst[i - 1] = post == 1 ? v : (v & (((u64)1 << o.aux) - 1));     // fri_prover_query_rounds: x_index = challenge % lde_size
        } else if (post == 3) {
            n_out = 0;
            if (o.dst && lane < 12) o.dst[lane] = st;
        }
    }
    if (lane < 12) gc->state[lane] = st;
    if (lane < 8) { gc->inbuf[lane] = ib; gc->outbuf[lane] = st; }
    if (lane == 0) { gc->n_in = n_in; gc->n_out = n_out; }
}

// PoW search: candidate w at state[pos], response = permute(state)[7]; keep the smallest accepted w.
__global__ __launch_bounds__(256) void pow_grind_kernel(u64 *pow_state /* 14 words */, u64 base, u32 min_zeros) {
    u64 cand = base + (u64)blockIdx.x * blockDim.x + threadIdx.x;
    // a smaller witness is already known: agent-scope load (the per-XCD L2s are not coherent for plain loads)
    if (__hip_atomic_load(&pow_state[13], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < cand) return;
    u64 s[12];
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = pow_state[i];
    u32 pos = (u32)pow_state[12];
#pragma unroll
    for (int i = 0; i < 8; i++) if ((u32)i == pos) s[i] = cand;
    psd::permute(s);
    if ((u32)__clzll((long long)s[7]) >= min_zeros) atomicMin((unsigned long long *)&pow_state[13], (unsigned long long)cand);
}

int32_t zk_transcript_run(zk_ctx *ctx, DevChallenger *c, const TOpList &ops) {
    if (ctx->host_transcript) return zk_host_transcript_run(ctx, c, ops);
    const int wave = (ctx->poseidon_mfma & 4) ? 0 : 1;
    ZK_LAUNCH(ctx, "transcript", transcript_kernel, dim3(1), dim3(64), 0, c, ops, ctx->d_err, wave, ctx->force_redo);
    return ZK_OK;
}

int32_t zk_pow_search_dev(zk_ctx *ctx, u32 min_zeros, u64 *d_pow_state) {
    // expected 2^min_zeros candidates; 2^(min_zeros+6) of them fail with probability e^-64.  Later blocks exit early
    // once a smaller witness is recorded.
    u64 total = (u64)1 << (min_zeros + 6 > 40 ? 40 : min_zeros + 6);
    // Launches double: the first covers the expected search (2^min_zeros candidates: one wave per SIMD at 16 bits, i.e. ONE permutation's
    // latency; misses with probability 1/e), every further one as many candidates as all before it (cumulative 2, 4, 8 x: missed with
    // probability e^-2, e^-4, e^-8), the last the rest.  A later launch costs its blocks one load each once a witness is known.  Expected
    // work 1.7 x the expected search; round 3's 1 x / 4 x / rest was 3.2 x -- every second proof ran 2^18 permutations where 2^17 do, and
    // the PoW search was 5.5 % of a transaction's vector instructions (profiles/r04l_...); round 2's 4 x first was 2^18 for every proof,
    // and before that 32 equal launches were 2 ms of launch latency per seven-table unit.
    u64 base = 0;
    u64 cuts[5];
    for (int k = 0; k < 4; k++) { const u64 c = (u64)1 << (min_zeros + k); cuts[k] = c < total ? c : total; }
    cuts[4] = total;
    for (int k = 0; k < 5; k++) {
        if (cuts[k] <= base) continue;
        u64 cnt = cuts[k] - base;
        while (cnt) {                                            // a launch is at most 2^31 - 256 blocks
            const u64 c = cnt < ((u64)1 << 38) ? cnt : ((u64)1 << 38);
            ZK_LAUNCH(ctx, "pow_grind", pow_grind_kernel, dim3((unsigned)((c + 255) / 256)), dim3(256), 0, d_pow_state, base, min_zeros);
            base += c; cnt -= c;
        }
    }
    return ZK_OK;
}

int32_t zk_pow_grind_dev(zk_ctx *ctx, DevChallenger *c, u32 min_zeros, u64 *d_pow_state, u64 *d_witness_out) {
    TOpList a = {}; a.n = 1; a.op[0].kind = TOP_POW_SETUP; a.op[0].dst = d_pow_state;
    zk_transcript_run(ctx, c, a);
    zk_pow_search_dev(ctx, min_zeros, d_pow_state);
    TOpList b = {}; b.n = 1; b.op[0].kind = TOP_POW_FINISH; b.op[0].src = d_pow_state; b.op[0].dst = d_witness_out; b.op[0].n = min_zeros;
    zk_transcript_run(ctx, c, b);
    return ZK_OK;
}

extern "C" {

int32_t zk_challenger_new(zk_ctx *ctx, zk_challenger **out) {
    if (!ctx || !out) return ZK_ERR_INVALID_ARG;
    hipSetDevice(ctx->device);
    zk_challenger *h = new zk_challenger();
    h->ctx = ctx;
    h->d = (DevChallenger *)ctx->alloc(sizeof(DevChallenger));
    h->scratch_words = 4096;
    h->scratch = (u64 *)ctx->alloc(h->scratch_words * sizeof(u64));
    if (!h->d || !h->scratch) { delete h; return ZK_ERR_OOM; }
    TOpList a = {}; a.n = 1; a.op[0].kind = TOP_INIT;
    zk_transcript_run(ctx, h->d, a);
    ctx->live_challengers.push_back(h->d);
    *out = h;
    return ZK_OK;
}
int32_t zk_challenger_free(zk_challenger *h) {
    if (!h) return ZK_ERR_INVALID_ARG;
    hipSetDevice(h->ctx->device);
    zk_ctx_synchronize(h->ctx);
    zk_host_transcript_forget(h->ctx, h->d);               // its host-side sponge, if the host path holds one
    { auto &v = h->ctx->live_challengers; for (size_t i = 0; i < v.size(); i++) if (v[i] == h->d) { v.erase(v.begin() + i); break; } }
    h->ctx->release(h->d); h->ctx->release(h->scratch);
    delete h;
    return ZK_OK;
}
int32_t zk_challenger_observe(zk_challenger *h, const uint64_t *elems, uint32_t n) {
    if (!h || (!elems && n)) return ZK_ERR_INVALID_ARG;
    hipSetDevice(h->ctx->device);
    while (n) {
        uint32_t m = n < h->scratch_words ? n : (uint32_t)h->scratch_words;
        int32_t rc = zk_dev_upload(h->ctx, h->scratch, elems, m * sizeof(u64));
        if (rc) return rc;
        TOpList a = {}; a.n = 1; a.op[0].kind = TOP_OBSERVE; a.op[0].src = h->scratch; a.op[0].n = m;
        zk_transcript_run(h->ctx, h->d, a);
        rc = zk_ctx_synchronize(h->ctx);
        if (rc) return rc;
        elems += m; n -= m;
    }
    return ZK_OK;
}
int32_t zk_challenger_get(zk_challenger *h, uint32_t n, uint64_t *out) {
    if (!h || (!out && n)) return ZK_ERR_INVALID_ARG;
    hipSetDevice(h->ctx->device);
    while (n) {
        uint32_t m = n < h->scratch_words ? n : (uint32_t)h->scratch_words;
        TOpList a = {}; a.n = 1; a.op[0].kind = TOP_GET; a.op[0].dst = h->scratch; a.op[0].n = m;
        zk_transcript_run(h->ctx, h->d, a);
        int32_t rc = zk_dev_download(h->ctx, out, h->scratch, m * sizeof(u64));
        if (rc) return rc;
        out += m; n -= m;
    }
    return ZK_OK;
}
int32_t zk_challenger_compact(zk_challenger *h, uint64_t out_state[12]) {
    if (!h) return ZK_ERR_INVALID_ARG;
    hipSetDevice(h->ctx->device);
    TOpList a = {}; a.n = 1; a.op[0].kind = TOP_COMPACT; a.op[0].dst = h->scratch;
    zk_transcript_run(h->ctx, h->d, a);
    if (out_state) return zk_dev_download(h->ctx, out_state, h->scratch, 12 * sizeof(u64));
    return zk_ctx_synchronize(h->ctx);
}

}  // extern "C"
