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

// The kernel runs as ONE wave.  The challenger is staged in LDS; every lane executes the (uniform) control flow, lane 0
// does the bookkeeping writes, and the permutation is the wave-wide one (poseidon_wave.cuh): state element l in lane l, the
// whole wave working on the one permutation -- a sponge is a serial chain, so the latency of ONE permutation is all that counts.
// (bit 2 of ZK_POSEIDON_MFMA selects the older 16-lane version, poseidon_coop.cuh, for comparison and for the tests.)
// One copy of the permutation code in the kernel (every call site jumps here): the instruction stream of a permutation is what
// the single wave waits for, so it should stay resident in the instruction cache.
struct ChPerm {
    const u64 *wtab; bool wave; u64 force_redo;
    __device__ __noinline__ u64 run(u64 s, int lane) const {
        if (wave) return psdw::permute(s, lane, wtab, force_redo);
        return psdc::permute16(s, lane & 15, lane & ~15);
    }
};
__device__ void ch_duplex(DevChallenger *c, ChPerm &P) {
    const int lane = threadIdx.x, l = lane & 15;
    u64 s = 0;
    if (l < 12) s = c->state[l];
    if ((u32)l < c->n_in) s = c->inbuf[l];
    __syncthreads();
    s = P.run(s, lane);
    if (lane < 12) c->state[lane] = s;
    if (lane < 8) c->outbuf[lane] = s;
    if (lane == 0) { c->n_in = 0; c->n_out = 8; }
    __syncthreads();
}
__device__ void ch_observe(DevChallenger *c, ChPerm &P, u64 x) {
    if (threadIdx.x == 0) { c->n_out = 0; c->inbuf[c->n_in] = gl_canon(x); c->n_in = c->n_in + 1; }
    __syncthreads();
    if (c->n_in == 8) ch_duplex(c, P);
}
__device__ u64 ch_get(DevChallenger *c, ChPerm &P) {
    if (c->n_in || !c->n_out) ch_duplex(c, P);
    u64 v = c->outbuf[c->n_out - 1];
    __syncthreads();
    if (threadIdx.x == 0) c->n_out = c->n_out - 1;
    __syncthreads();
    return v;
}

__global__ __launch_bounds__(64) void transcript_kernel(DevChallenger *gc, TOpList ops, u64 *err, int wave, u64 force_redo) {
    __shared__ DevChallenger L;
    __shared__ u64 wtab[psdw::TAB_WORDS];
    DevChallenger *c = &L;
    const int lane = threadIdx.x;
    ChPerm P; P.wave = wave != 0; P.wtab = wtab; P.force_redo = force_redo;
    if (P.wave) psdw::fill<64>(wtab);
    {
        const u64 *src = (const u64 *)gc; u64 *dst = (u64 *)c;
        for (u32 i = lane; i < sizeof(DevChallenger) / 8; i += 64) dst[i] = src[i];
    }
    __syncthreads();
    for (int k = 0; k < ops.n; k++) {
        const TOp &o = ops.op[k];
        switch (o.kind) {
        case TOP_INIT:
            if (lane < 12) c->state[lane] = 0;
            if (lane == 0) { c->n_in = 0; c->n_out = 0; }
            __syncthreads();
            break;
        case TOP_OBSERVE: {
            // element by element until the input buffer is empty, then whole rate-sized chunks with the sponge state kept in
            // registers: one coalesced (prefetched) load and one permutation per 8 elements -- wide opening sets (thousands of
            // columns) are a serial chain of permutations, so everything else must stay off that chain
            u32 i = 0;
            while (i < o.n && c->n_in != 0) ch_observe(c, P, o.src[i++]);
            if (i + 8 <= o.n) {
                const int l = lane & 15;
                u64 st = l < 12 ? c->state[l] : 0;
                u64 nxt = l < 8 ? gl_canon(o.src[i + l]) : 0;
                while (i + 8 <= o.n) {
                    const u64 cur = nxt;
                    i += 8;
                    if (i + 8 <= o.n && l < 8) nxt = gl_canon(o.src[i + l]);
                    if (l < 8) st = cur;
                    st = P.run(st, lane);
                    if (l >= 12) st = 0;
                }
                __syncthreads();
                if (lane < 12) c->state[lane] = st;
                if (lane < 8) c->outbuf[lane] = st;
                if (lane == 0) { c->n_in = 0; c->n_out = 8; }
                __syncthreads();
            }
            while (i < o.n) ch_observe(c, P, o.src[i++]);
            break;
        }
        case TOP_OBSERVE_BASE_AS_EXT:
            for (u32 i = 0; i < o.n; i++) { ch_observe(c, P, o.src[i]); ch_observe(c, P, 0); }
            break;
        case TOP_GET:
            for (u32 i = 0; i < o.n; i++) { u64 v = ch_get(c, P); if (lane == 0) o.dst[i] = v; }
            break;
        case TOP_COMPACT:
            if (c->n_in) ch_duplex(c, P);
            if (lane == 0) c->n_out = 0;
            __syncthreads();
            if (o.dst && lane < 12) o.dst[lane] = c->state[lane];
            break;
        case TOP_POW_SETUP: {
            // fri_proof_of_work: duplex_intermediate_state = sponge state with pending inputs written in; dst[12] = witness position
            if (lane < 12) o.dst[lane] = (u32)lane < c->n_in ? c->inbuf[lane] : c->state[lane];
            if (lane == 0) { o.dst[12] = c->n_in; o.dst[13] = ~0ULL; }    // dst[13] = best witness so far
            break;
        }
        case TOP_POW_FINISH: {
            u64 w = __hip_atomic_load((u64 *)&o.src[13], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (w == ~0ULL) { if (lane == 0) atomicCAS((unsigned long long *)err, 0ULL, (unsigned long long)(uint32_t)ZK_ERR_POW); w = 0; }
            ch_observe(c, P, w);
            u64 resp = ch_get(c, P);
            if (lane == 0) {
                if ((u32)__clzll((long long)resp) < o.n) atomicCAS((unsigned long long *)err, 0ULL, (unsigned long long)(uint32_t)ZK_ERR_POW);
                o.dst[0] = w;
            }
            break;
        }
        case TOP_GET_INDICES:
            // fri_prover_query_rounds: x_index = challenge.to_canonical_u64() as usize % lde_size (lde_size = 2^o.aux)
            for (u32 i = 0; i < o.n; i++) { u64 v = ch_get(c, P); if (lane == 0) o.dst[i] = v & (((u64)1 << o.aux) - 1); }
            break;
        }
    }
    __syncthreads();
    {
        const u64 *src = (const u64 *)c; u64 *dst = (u64 *)gc;
        for (u32 i = lane; i < sizeof(DevChallenger) / 8; i += 64) dst[i] = src[i];
    }
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
    const int wave = (ctx->poseidon_mfma & 4) ? 0 : 1;
    ZK_LAUNCH(ctx, "transcript", transcript_kernel, dim3(1), dim3(64), 0, c, ops, ctx->d_err, wave, ctx->force_redo);
    return ZK_OK;
}

int32_t zk_pow_search_dev(zk_ctx *ctx, u32 min_zeros, u64 *d_pow_state) {
    // expected 2^min_zeros candidates; 2^(min_zeros+6) of them fail with probability e^-64.  Later blocks exit early
    // once a smaller witness is recorded.
    u64 total = (u64)1 << (min_zeros + 6 > 40 ? 40 : min_zeros + 6);
    // Launches grow: the first covers the expected search (2^min_zeros candidates: one wave per SIMD at 16 bits, i.e. ONE permutation's
    // latency; misses with probability 1/e), the next 4 x, the last the rest.  A later launch costs its blocks one load each once a
    // witness is known.  (Round 2 started with 4 x the expected search: every proof paid 2^18 permutations, 160 us, where 2^16 + 0.37 x
    // 2^18 are enough on average; before that, 32 equal launches were 2 ms of launch latency per seven-table unit.)
    u64 base = 0;
    const u64 first = (u64)1 << min_zeros, second = (u64)1 << (min_zeros + 2);
    const u64 cuts[3] = {first < total ? first : total, second < total ? second : total, total};
    for (int k = 0; k < 3; k++) {
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
    *out = h;
    return ZK_OK;
}
int32_t zk_challenger_free(zk_challenger *h) {
    if (!h) return ZK_ERR_INVALID_ARG;
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
