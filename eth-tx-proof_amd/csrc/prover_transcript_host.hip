// Host-side Fiat-Shamir challenger (round 4; SURVEY.md 8a row a8 placed the challenger on the host).
// The device-resident transcript (transcript.hip) runs the duplex sponge on ONE wave: 7.5 us per permutation, and a proof's sponge is a
// serial chain -- 2 400 permutations = 18 ms of a seven-table unit's 44 ms alone, 20 ms of a transaction's 72.  A CPU core does the same
// permutation in about a microsecond.  With zk_ctx_set_host_transcript(ctx, 1) (or ZK_TRANSCRIPT_HOST=1) every zk_transcript_run of the
// context is served by a host thread instead, WITHOUT a host synchronisation of the proving thread:
//   proving thread (enqueues ahead, as before):   one kernel that writes the observed words to mapped pinned memory and then this job's
//       sequence number to a flag there  ->  [only if the job produces challenges]  hipStreamWaitValue64 on the reply flag  ->  one kernel
//       that moves the challenges from pinned memory to where the next kernels read them;
//   service thread (one per process, polls the mailboxes of all contexts):   waits for the job's flag, absorbs / squeezes with the sponge
//       state it keeps per challenger, writes the challenges to pinned memory, publishes the job's sequence number in the reply flag.
// A job that only observes never stalls the GPU; one that squeezes costs one round trip (3.6 us measured, tools/host_roundtrip.hip) plus
// its copies.  Same sponge, same words: Challenger<F, PoseidonHash> of plonky2 0.2.2 (overwrite-mode duplex, rate 8, challenges popped from
// the back of the output buffer), the permutation in the sparse partial-round form of poseidon_constants.inc (checked against the device
// kernels by every GPU test that runs with the switch on).  Failure rule: whatever goes wrong on the host side, the reply flag is still
// published -- a stream never waits for a value that will not come -- and the error reaches the proof's status word.
#include "zk_internal.h"
#include "transcript.h"
#include "transcript_host_core.h"       // the sponge, the mailbox, the service threads, the enqueue logic: everything that needs no GPU

namespace {
// The words a job observes go to the pinned ring by ONE kernel (the device writes mapped host memory directly) that publishes the job's
// sequence number when every word is out; the challenges come back by one kernel behind the stream-side wait.  Copy-engine transfers
// (hipMemcpyAsync) cost 10-20 us of latency each and a job has up to six segments; a kernel launch costs 3-5.
__global__ __launch_bounds__(256) void host_transcript_gather_kernel(HtSegs g, volatile u64 *flag, u64 seq) {
    for (u32 k = 0; k < g.count; k++)
        for (u32 i = threadIdx.x; i < g.n[k]; i += 256) g.dst[k][i] = g.src[k][i];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) { *flag = seq; __threadfence_system(); }
}
__global__ __launch_bounds__(256) void host_transcript_scatter_kernel(HtSegs g) {
    for (u32 k = 0; k < g.count; k++)
        for (u32 i = threadIdx.x; i < g.n[k]; i += 256) g.dst[k][i] = ((const volatile u64 *)g.src[k])[i];
}
// the device side of host_transcript_enqueue on a HIP stream
struct HipDev {
    zk_ctx *ctx;
    void sync() { hipStreamSynchronize(ctx->stream); }
    void gather(const HtSegs &g, u64 *flag, u64 seq) {
        ProfScope ps(ctx, "transcript_host_gather");
        ctx->note_launch("host_transcript_gather", 1, 256);
        hipLaunchKernelGGL(host_transcript_gather_kernel, dim3(1), dim3(256), 0, ctx->stream, g, (volatile u64 *)flag, seq);
    }
    void wait_scatter(u64 *flag, u64 seq, const HtSegs &g) {
        if (hipStreamWaitValue64(ctx->stream, (void *)flag, seq, hipStreamWaitValueGte, ~0ull) != hipSuccess) {
            // no stream wait on this stack: wait on the host instead (correct, slower)
            (void)hipGetLastError();
            hipStreamSynchronize(ctx->stream);
            while (ht_flag_load(flag) < seq) _mm_pause();
        }
        ProfScope ps(ctx, "transcript_host_scatter");
        ctx->note_launch("host_transcript_wait+scatter", 1, 256);
        hipLaunchKernelGGL(host_transcript_scatter_kernel, dim3(1), dim3(256), 0, ctx->stream, g);
    }
    int32_t fail(int32_t code, const char *msg) { return ctx->fail(code, msg); }
};
}  // namespace

static zk_host_transcript *mailbox(zk_ctx *ctx) {
    if (ctx->hmb) return ctx->hmb;
    zk_host_transcript *H = new zk_host_transcript();
    u64 *flags = nullptr;
    const size_t RING = (size_t)16 << 20;
    if (hipHostMalloc((void **)&flags, 256, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
        hipHostMalloc((void **)&H->ring, RING, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) { (void)hipGetLastError(); if (flags) hipHostFree(flags); delete H; return nullptr; }
    flags[0] = 0; flags[16] = 0;
    ZK_ALLOC_TRACE("host-challenger-flags", flags, 256); ZK_ALLOC_TRACE("host-challenger-ring", H->ring, RING);
    H->gpu_done = flags; H->reply = flags + 16; H->ring_cap = RING;
    ht_mailbox_publish(H);
    ctx->hmb = H;
    return H;
}

void zk_host_transcript_destroy(zk_ctx *ctx) {
    zk_host_transcript *H = ctx->hmb;
    if (!H) return;
    hipStreamSynchronize(ctx->stream);
    ht_mailbox_retire(H);                                  // queue drained, out of the service's list, nobody inside it
    ZK_ALLOC_TRACE("free-host-challenger-flags", H->gpu_done, 256); ZK_ALLOC_TRACE("free-host-challenger-ring", H->ring, 0);
    hipHostFree((void *)H->gpu_done); hipHostFree(H->ring);
    delete H;
    ctx->hmb = nullptr;
}

// start of a proof: every job of the previous one has been served (the proof ended with a host synchronisation, but a trailing job that
// only observed may still be in the queue), so the pinned ring can start over
void zk_host_transcript_begin_proof(zk_ctx *ctx) {
    zk_host_transcript *H = ctx->hmb;
    if (!H) return;
    ht_wait_idle(H);
    H->ring_used = 0;
}

int32_t zk_host_transcript_run(zk_ctx *ctx, DevChallenger *c, const TOpList &ops) {
    zk_host_transcript *H = mailbox(ctx);
    if (!H) return ctx->fail(ZK_ERR_OOM, "host transcript: pinned memory");
    HipDev dev{ctx};
    return host_transcript_enqueue(H, dev, c, ops);
}

// ---- the switch with challengers alive: the device path keeps a sponge in its DevChallenger block, the host path in
// H->sponges.  Both callers hold the context between proofs (stream synchronised, queue drained).  Upstream's output buffer is the first
// eight state words as of the last permutation, so (state, inbuf, n_in, n_out) is the whole challenger on either side.
int32_t zk_host_transcript_adopt(zk_ctx *ctx, DevChallenger *d) {
    zk_host_transcript *H = mailbox(ctx);
    if (!H) return ctx->fail(ZK_ERR_OOM, "host transcript: pinned memory");
    DevChallenger h;
    if (hipMemcpy(&h, d, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) return ctx->fail(ZK_ERR_HIP, "host transcript: challenger download");
    psdh::Sponge S;
    for (int i = 0; i < 12; i++) S.state[i] = h.state[i];
    for (int i = 0; i < 8; i++) S.inbuf[i] = h.inbuf[i];
    S.n_in = h.n_in; S.n_out = h.n_out;
    std::lock_guard<std::mutex> lk(H->mu);
    H->sponges[d] = S;
    return ZK_OK;
}
int32_t zk_host_transcript_hand_back(zk_ctx *ctx, DevChallenger *d) {
    zk_host_transcript *H = ctx->hmb;
    if (!H) return ZK_OK;                                  // the host path never ran: the device block is current
    DevChallenger h;
    {
        std::lock_guard<std::mutex> lk(H->mu);
        auto it = H->sponges.find(d);
        if (it == H->sponges.end()) return ZK_OK;
        const psdh::Sponge &S = it->second;
        for (int i = 0; i < 12; i++) h.state[i] = S.state[i];
        for (int i = 0; i < 8; i++) { h.inbuf[i] = S.inbuf[i]; h.outbuf[i] = S.state[i]; }
        h.n_in = S.n_in; h.n_out = S.n_out;
        H->sponges.erase(it);
    }
    if (hipMemcpy(d, &h, sizeof h, hipMemcpyHostToDevice) != hipSuccess) return ctx->fail(ZK_ERR_HIP, "host transcript: challenger upload");
    return ZK_OK;
}
void zk_host_transcript_forget(zk_ctx *ctx, const DevChallenger *d) {
    zk_host_transcript *H = ctx->hmb;
    if (!H) return;
    ht_wait_idle(H);                                                      // a trailing job may still name it
    std::lock_guard<std::mutex> lk(H->mu);
    H->sponges.erase(d);
}
int32_t zk_host_transcript_take_error(zk_ctx *ctx) { return ctx->hmb ? ctx->hmb->error.exchange(0) : 0; }

// ---- the host permutation and sponge by themselves (no GPU needed): what the CPU suite pins to upstream's known-answer vectors
extern "C" int32_t zk_host_poseidon_permute(uint64_t state[12]) {
    if (!state) return ZK_ERR_INVALID_ARG;
    u64 s[12];
    for (int i = 0; i < 12; i++) s[i] = state[i] >= psdh::P ? state[i] - psdh::P : state[i];
    psdh::permute(s);
    for (int i = 0; i < 12; i++) state[i] = s[i];
    return ZK_OK;
}
// one pass of the host challenger over a script: ops[i] > 0 = observe that many of the next `elems`, ops[i] < 0 = squeeze -ops[i] challenges
// into `out`, ops[i] == 0 = compact; returns the number of challenges written (or a negative ZK_ERR_*)
extern "C" int64_t zk_host_challenger_script(const int32_t *ops, uint32_t n_ops, const uint64_t *elems, size_t n_elems, uint64_t *out, size_t cap_out,
                                             uint64_t state_out[12]) {
    if ((!ops && n_ops) || (!elems && n_elems)) return ZK_ERR_INVALID_ARG;
    psdh::Sponge S;
    size_t at = 0, w = 0;
    for (uint32_t i = 0; i < n_ops; i++) {
        if (ops[i] > 0) { for (int32_t k = 0; k < ops[i]; k++) { if (at >= n_elems) return ZK_ERR_INVALID_ARG; S.observe(elems[at++]); } }
        else if (ops[i] < 0) { for (int32_t k = 0; k < -ops[i]; k++) { if (w >= cap_out) return ZK_ERR_BUFFER_TOO_SMALL; out[w++] = S.get(); } }
        else { if (S.n_in) S.duplex(); S.n_out = 0; }
    }
    if (state_out) for (int i = 0; i < 12; i++) state_out[i] = S.state[i];
    return (int64_t)w;
}
