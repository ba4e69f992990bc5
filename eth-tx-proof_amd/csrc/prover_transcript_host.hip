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
#include <stdlib.h>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <immintrin.h>
#include "zk_internal.h"
#include "transcript.h"

namespace psdh {
#define PSD_CONST static const
#include "poseidon_constants.inc"
#undef PSD_CONST
typedef unsigned __int128 u128;
static const u64 P = 0xFFFFFFFF00000001ULL;
// x mod p without a division: 2^64 = 2^32 - 1, 2^96 = -1 (mod p).  x = lo + hh 2^96 + hl 2^64  ->  lo - hh + hl (2^32 - 1)
static inline u64 red(u128 x) {
    const u64 lo = (u64)x, hi = (u64)(x >> 64);
    const u64 hh = hi >> 32, hl = hi & 0xFFFFFFFFull;
    u64 t0 = lo - hh;
    if (lo < hh) t0 -= 0xFFFFFFFFull;                 // borrow: + p = - (2^32 - 1) on the wrapped value
    const u64 t1 = hl * 0xFFFFFFFFull;
    u64 r = t0 + t1;
    if (r < t1) r += 0xFFFFFFFFull;                   // carry: 2^64 = 2^32 - 1
    return r >= P ? r - P : r;
}
static inline u64 mul(u64 a, u64 b) { return red((u128)a * b); }
static inline u64 add(u64 a, u64 b) { u128 s = (u128)a + b; return (u64)(s >= P ? s - P : s); }
static inline u64 sbox(u64 x) { const u64 x2 = mul(x, x), x3 = mul(x2, x), x4 = mul(x2, x2); return mul(x3, x4); }
static void mds(u64 s[12], const u64 *k) {
    static const u32 C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    u64 o[12];
    for (int r = 0; r < 12; r++) {
        u128 a = k ? k[r] : 0;
        for (int i = 0; i < 12; i++) a += (u128)s[(i + r) % 12] * C[i];
        if (r == 0) a += (u128)s[0] * 8u;
        o[r] = red(a);
    }
    for (int r = 0; r < 12; r++) s[r] = o[r];
}
static void full_rounds(u64 s[12], int first, const u64 *after) {
    for (int i = 0; i < 12; i++) s[i] = add(s[i], PSD_RC[12 * first + i]);
    for (int r = 0; r < 4; r++) {
        for (int i = 0; i < 12; i++) s[i] = sbox(s[i]);
        mds(s, r < 3 ? PSD_RC + 12 * (first + r + 1) : after);
    }
}
// canonical in, canonical out
static void permute(u64 s[12]) {
    full_rounds(s, 0, PSD_PR_FIRST);
    u64 t[11];
    for (int i = 0; i < 11; i++) { u128 a = 0; for (int j = 0; j < 11; j++) a += (u128)mul(PSD_PR_INIT[11 * i + j], s[1 + j]); t[i] = red(a); }
    for (int i = 0; i < 11; i++) s[1 + i] = t[i];
    for (int r = 0; r < 22; r++) {
        const u64 s0 = add(sbox(s[0]), PSD_PR_K[r]);
        u128 d = (u128)s0 * PSD_M00;
        const u64 *vh = PSD_PR_VHAT + 11 * r, *w = PSD_PR_W + 11 * r;
        for (int j = 0; j < 11; j++) d += (u128)mul(vh[j], s[1 + j]);
        for (int j = 0; j < 11; j++) s[1 + j] = add(s[1 + j], mul(w[j], s0));
        s[0] = red(d);
    }
    full_rounds(s, 26, nullptr);
}
struct Sponge {
    u64 state[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, inbuf[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    u32 n_in = 0, n_out = 0;
    void duplex() { for (u32 i = 0; i < n_in; i++) state[i] = inbuf[i]; permute(state); n_in = 0; n_out = 8; }
    void observe(u64 x) { n_out = 0; inbuf[n_in++] = x >= P ? x - P : x; if (n_in == 8) duplex(); }
    u64 get() { if (n_in || !n_out) duplex(); return state[--n_out]; }
};
}  // namespace psdh

namespace {
struct HostOp { int kind; u32 n, aux; const u64 *in; u64 *out; const DevChallenger *c; };
struct HostJob { u64 seq; bool needs_gpu; std::vector<HostOp> ops; };
}  // namespace

struct zk_host_transcript {
    std::mutex mu;                                    // queue + sponges are touched by the service thread; queue also by the proving thread
    std::deque<HostJob> q;
    std::condition_variable idle_cv;                  // signalled when the queue runs empty (begin_proof waits on it)
    std::unordered_map<const DevChallenger *, psdh::Sponge> sponges;
    volatile u64 *gpu_done = nullptr, *reply = nullptr;   // mapped pinned flags (two cache lines)
    char *ring = nullptr; size_t ring_cap = 0, ring_used = 0;
    u64 seq = 0;
    std::atomic<int32_t> error{0};
    bool busy = false;                                // the service thread is inside a job of this mailbox
};

namespace {
// A few service threads per process (ZK_TRANSCRIPT_HOST_THREADS, default 3), each polling the mailboxes of all contexts: jobs of ONE mailbox
// run in order (a mailbox is served by one thread at a time: `busy`), mailboxes are independent -- with a single thread the 1 250
// permutations of a wide table's opening set held up the shrink chains proved beside it on other contexts (44 instead of 14 ms).
struct Service {
    std::mutex mu; std::condition_variable cv;
    std::vector<zk_host_transcript *> boxes;
    bool started = false;
    std::atomic<u64> pending{0};
    void run();
    void ensure() {
        if (started) return;
        started = true;
        int n = getenv("ZK_TRANSCRIPT_HOST_THREADS") ? atoi(getenv("ZK_TRANSCRIPT_HOST_THREADS")) : 3;
        if (n < 1) n = 1;
        if (n > 16) n = 16;
        for (int i = 0; i < n; i++) std::thread([this] { run(); }).detach();
    }
};
Service &service() { static Service *s = new Service(); return *s; }     // leaked on purpose: the thread outlives static destructors

void process(zk_host_transcript *H, HostJob &job) {
    for (HostOp &o : job.ops) {
        psdh::Sponge &S = H->sponges[o.c];
        switch (o.kind) {
        case TOP_INIT: S = psdh::Sponge(); break;
        case TOP_OBSERVE: for (u32 i = 0; i < o.n; i++) S.observe(o.in[i]); break;
        case TOP_OBSERVE_BASE_AS_EXT: for (u32 i = 0; i < o.n; i++) { S.observe(o.in[i]); S.observe(0); } break;
        case TOP_GET: for (u32 i = 0; i < o.n; i++) o.out[i] = S.get(); break;
        case TOP_GET_INDICES: for (u32 i = 0; i < o.n; i++) o.out[i] = S.get() & (((u64)1 << o.aux) - 1); break;
        case TOP_COMPACT:
            if (S.n_in) S.duplex();
            S.n_out = 0;
            if (o.out) for (int i = 0; i < 12; i++) o.out[i] = S.state[i];
            break;
        case TOP_POW_SETUP:
            for (u32 i = 0; i < 12; i++) o.out[i] = i < S.n_in ? S.inbuf[i] : S.state[i];
            o.out[12] = S.n_in; o.out[13] = ~0ULL;
            break;
        case TOP_POW_FINISH: {
            u64 w = o.in[0];
            if (w == ~0ULL) { int32_t z = 0; H->error.compare_exchange_strong(z, ZK_ERR_POW); w = 0; }
            o.out[0] = w;
            S.observe(w);
            const u64 resp = S.get();
            const u32 lz = resp ? (u32)__builtin_clzll(resp) : 64u;
            if (lz < o.n) { int32_t z = 0; H->error.compare_exchange_strong(z, ZK_ERR_POW); }
            break;
        }
        default: break;
        }
    }
}

void Service::run() {
    u32 idle = 0;
    for (;;) {
        zk_host_transcript *H = nullptr;
        HostJob job;
        {
            // pick a mailbox whose head job is ready, under the service lock: a mailbox that is being destroyed has left `boxes` and
            // cannot be picked any more (zk_host_transcript_destroy)
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [this] { return pending.load() != 0; });
            for (zk_host_transcript *B : boxes) {
                std::lock_guard<std::mutex> bl(B->mu);
                if (B->busy || B->q.empty()) continue;                                       // another thread is inside this mailbox's head job
                if (B->q.front().needs_gpu && *B->gpu_done < B->q.front().seq) continue;      // its words have not landed yet
                job = std::move(B->q.front());
                B->q.pop_front();
                B->busy = true;
                H = B;
                break;
            }
        }
        if (!H) {
            // nothing ready: the jobs in the queues wait for words the GPU has not produced yet.  Spin for most of a millisecond (a
            // transcript phase answers within microseconds; napping after 100 us cost the seven-table unit 1.3 of 31 ms), then nap:
            // during a proof's long kernels nobody is waiting for this thread
            if (++idle < 512) { for (int i = 0; i < 32; i++) _mm_pause(); }
            else std::this_thread::sleep_for(std::chrono::microseconds(30));
            continue;
        }
        idle = 0;
        std::atomic_thread_fence(std::memory_order_acquire);
        process(H, job);
        std::atomic_thread_fence(std::memory_order_release);
        *H->reply = job.seq;                                       // ALWAYS published: a stream never waits for a value that will not come
        __sync_synchronize();
        pending.fetch_sub(1);
        {
            std::lock_guard<std::mutex> bl(H->mu);                 // last touch of the mailbox: a waiter (destroy, begin_proof) proceeds after the unlock
            H->busy = false;
            if (H->q.empty()) H->idle_cv.notify_all();
        }
    }
}

// The words a job observes go to the pinned ring by ONE kernel (the device writes mapped host memory directly) that publishes the job's
// sequence number when every word is out; the challenges come back by one kernel behind the stream-side wait.  Copy-engine transfers
// (hipMemcpyAsync) cost 10-20 us of latency each and a job has up to six segments; a kernel launch costs 3-5.
#define HT_MAX_SEG 8
struct HtSegs { const u64 *src[HT_MAX_SEG]; u64 *dst[HT_MAX_SEG]; u32 n[HT_MAX_SEG]; u32 count; };
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
    Service &S = service();
    { std::lock_guard<std::mutex> lk(S.mu); S.boxes.push_back(H); S.ensure(); }
    ctx->hmb = H;
    return H;
}

void zk_host_transcript_destroy(zk_ctx *ctx) {
    zk_host_transcript *H = ctx->hmb;
    if (!H) return;
    hipStreamSynchronize(ctx->stream);
    { std::unique_lock<std::mutex> lk(H->mu); H->idle_cv.wait(lk, [H] { return H->q.empty() && !H->busy; }); }
    Service &S = service();
    { std::lock_guard<std::mutex> lk(S.mu); for (size_t i = 0; i < S.boxes.size(); i++) if (S.boxes[i] == H) { S.boxes.erase(S.boxes.begin() + i); break; } }
    { std::unique_lock<std::mutex> lk(H->mu); H->idle_cv.wait(lk, [H] { return !H->busy; }); }       // nobody can pick it any more
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
    std::unique_lock<std::mutex> lk(H->mu);
    H->idle_cv.wait(lk, [H] { return H->q.empty() && !H->busy; });
    H->ring_used = 0;
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
    std::unique_lock<std::mutex> lk(H->mu);
    H->idle_cv.wait(lk, [H] { return H->q.empty() && !H->busy; });       // a trailing job may still name it
    H->sponges.erase(d);
}
int32_t zk_host_transcript_take_error(zk_ctx *ctx) { return ctx->hmb ? ctx->hmb->error.exchange(0) : 0; }

int32_t zk_host_transcript_run(zk_ctx *ctx, DevChallenger *c, const TOpList &ops) {
    zk_host_transcript *H = mailbox(ctx);
    if (!H) return ctx->fail(ZK_ERR_OOM, "host transcript: pinned memory");
    // the pinned ring: bump-allocated, started over at every proof; a call that would not fit first waits until everything enqueued so far
    // has been consumed (GPU and service thread) -- BEFORE it allocates anything, so a job's own buffers never move under it
    {
        size_t need = 0;
        for (int k = 0; k < ops.n; k++) need += (((size_t)ops.op[k].n + 16) * 8 + 63) & ~(size_t)63;
        if (need > H->ring_cap) return ctx->fail(ZK_ERR_OOM, "host transcript: one call exceeds the pinned ring");
        if (H->ring_used + need > H->ring_cap) {
            hipStreamSynchronize(ctx->stream);
            std::unique_lock<std::mutex> lk(H->mu);
            H->idle_cv.wait(lk, [H] { return H->q.empty() && !H->busy; });
            H->ring_used = 0;
        }
    }
    auto ring = [&](size_t words) -> u64 * {
        const size_t need = (words * 8 + 63) & ~(size_t)63;
        if (H->ring_used + need > H->ring_cap) return nullptr;
        u64 *p = (u64 *)(H->ring + H->ring_used);
        H->ring_used += need;
        return p;
    };
    HostJob job; job.seq = ++H->seq; job.needs_gpu = false;
    HtSegs gin = {}, gout = {};
    auto seg = [&](HtSegs &g, const u64 *src, u64 *dst, u32 n) -> bool {
        if (g.count >= HT_MAX_SEG) return false;
        g.src[g.count] = src; g.dst[g.count] = dst; g.n[g.count] = n; g.count++;
        return true;
    };
    for (int k = 0; k < ops.n; k++) {
        const TOp &o = ops.op[k];
        HostOp h = {o.kind, o.n, o.aux, nullptr, nullptr, c};
        switch (o.kind) {
        case TOP_OBSERVE: case TOP_OBSERVE_BASE_AS_EXT: {
            if (!o.n) continue;
            u64 *in = ring(o.n);
            if (!in) return ctx->fail(ZK_ERR_OOM, "host transcript: observed words exceed the pinned ring");
            if (!seg(gin, o.src, in, o.n)) return ctx->fail(ZK_ERR_INVALID_ARG, "host transcript: too many segments in one call");
            h.in = in; job.needs_gpu = true;
            break;
        }
        case TOP_GET: case TOP_GET_INDICES: {
            if (!o.n) continue;
            u64 *out = ring(o.n); if (!out) return ctx->fail(ZK_ERR_OOM, "host transcript: ring");
            h.out = out;
            if (!seg(gout, out, o.dst, o.n)) return ctx->fail(ZK_ERR_INVALID_ARG, "host transcript: too many segments in one call");
            break;
        }
        case TOP_COMPACT:
            if (o.dst) { u64 *out = ring(12); if (!out) return ctx->fail(ZK_ERR_OOM, "host transcript: ring"); h.out = out; if (!seg(gout, out, o.dst, 12)) return ctx->fail(ZK_ERR_INVALID_ARG, "host transcript: segments"); }
            break;
        case TOP_POW_SETUP: { u64 *out = ring(14); if (!out) return ctx->fail(ZK_ERR_OOM, "host transcript: ring"); h.out = out; if (!seg(gout, out, o.dst, 14)) return ctx->fail(ZK_ERR_INVALID_ARG, "host transcript: segments"); break; }
        case TOP_POW_FINISH: {
            u64 *in = ring(1), *out = ring(1); if (!in || !out) return ctx->fail(ZK_ERR_OOM, "host transcript: ring");
            if (!seg(gin, o.src + 13, in, 1) || !seg(gout, out, o.dst, 1)) return ctx->fail(ZK_ERR_INVALID_ARG, "host transcript: segments");
            h.in = in; h.out = out; job.needs_gpu = true;
            break;
        }
        default: break;
        }
        job.ops.push_back(h);
    }
    const u64 seq = job.seq;
    if (job.needs_gpu) {
        ProfScope ps(ctx, "transcript_host_gather");
        ctx->note_launch("host_transcript_gather", 1, 256);
        hipLaunchKernelGGL(host_transcript_gather_kernel, dim3(1), dim3(256), 0, ctx->stream, gin, H->gpu_done, seq);
    }
    Service &S = service();
    { std::lock_guard<std::mutex> lk(H->mu); H->q.push_back(std::move(job)); }
    S.pending.fetch_add(1);
    { std::lock_guard<std::mutex> lk(S.mu); S.cv.notify_all(); }
    if (gout.count) {
        if (hipStreamWaitValue64(ctx->stream, (void *)H->reply, seq, hipStreamWaitValueGte, ~0ull) != hipSuccess) {
            // no stream wait on this stack: wait on the host instead (correct, slower)
            (void)hipGetLastError();
            hipStreamSynchronize(ctx->stream);
            while (*H->reply < seq) _mm_pause();
        }
        ProfScope ps(ctx, "transcript_host_scatter");
        ctx->note_launch("host_transcript_wait+scatter", 1, 256);
        hipLaunchKernelGGL(host_transcript_scatter_kernel, dim3(1), dim3(256), 0, ctx->stream, gout);
    }
    return ZK_OK;
}

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
