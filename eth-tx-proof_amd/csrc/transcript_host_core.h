// Host-side Fiat-Shamir challenger: the part that needs no GPU (round 5: split off prover_transcript_host.hip so that the mailbox /
// service-thread protocol can be built with a plain C++ compiler and run under ThreadSanitizer, tests/cpp/transcript_host_tsan.cpp --
// a thread plays the device side of the two flags there).
//   * psdh: the scalar Poseidon permutation and the duplex sponge (Challenger<F, PoseidonHash> of plonky2 0.2.2: overwrite mode, rate 8,
//     challenges popped from the back of the output buffer), pinned by upstream's known-answer vectors in tests/test_abi_cpu.py;
//   * zk_host_transcript: one mailbox per context -- a queue of jobs, the sponges of its challengers, two flags in memory that host and
//     device both see (gpu_done: "the words job s observes have landed", written by the device; reply: "job s has been served", written by
//     the service thread), a bump-allocated ring for the words that travel;
//   * Service: a few threads per process that poll the mailboxes; jobs of ONE mailbox run in order, mailboxes are independent;
//   * host_transcript_enqueue<Dev>: what zk_transcript_run does on this path, with the three device-side actions (synchronise the stream,
//     gather observed words + publish gpu_done, wait for reply + scatter the challenges) behind a policy type.
// Flags are read and written with acquire / release atomics: on the real path the other side is the GPU (mapped pinned memory), in the
// sanitizer harness it is a CPU thread, and the same ordering argument covers both.
// Failure rule: whatever goes wrong on the host side, the reply flag is still published -- a stream never waits for a value that will not
// come -- and the error reaches the proof's status word.
#pragma once
#include <stdlib.h>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <vector>
#include <immintrin.h>
#include "../../include/zkgl.h"
#include "transcript_ops.h"

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
static inline void mds(u64 s[12], const u64 *k) {
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
static inline void full_rounds(u64 s[12], int first, const u64 *after) {
    for (int i = 0; i < 12; i++) s[i] = add(s[i], PSD_RC[12 * first + i]);
    for (int r = 0; r < 4; r++) {
        for (int i = 0; i < 12; i++) s[i] = sbox(s[i]);
        mds(s, r < 3 ? PSD_RC + 12 * (first + r + 1) : after);
    }
}
// canonical in, canonical out
static inline void permute(u64 s[12]) {
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

struct HostOp { int kind; u32 n, aux; const u64 *in; u64 *out; const DevChallenger *c; };
struct HostJob { u64 seq; bool needs_gpu; std::vector<HostOp> ops; };

struct zk_host_transcript {
    std::mutex mu;                                    // queue + sponges are touched by the service thread; queue also by the proving thread
    std::deque<HostJob> q;
    std::condition_variable idle_cv;                  // signalled when the queue runs empty (begin_proof waits on it)
    std::unordered_map<const DevChallenger *, psdh::Sponge> sponges;
    u64 *gpu_done = nullptr, *reply = nullptr;        // flags both sides see (two cache lines); acquire / release atomics only
    char *ring = nullptr; size_t ring_cap = 0, ring_used = 0;
    u64 seq = 0;
    std::atomic<int32_t> error{0};
    bool busy = false;                                // the service thread is inside a job of this mailbox
};
static inline u64 ht_flag_load(const u64 *f) { return __atomic_load_n(f, __ATOMIC_ACQUIRE); }
static inline void ht_flag_store(u64 *f, u64 v) { __atomic_store_n(f, v, __ATOMIC_RELEASE); }

// A few service threads per process (ZK_TRANSCRIPT_HOST_THREADS, default 3), each polling the mailboxes of all contexts: jobs of ONE mailbox
// run in order (a mailbox is served by one thread at a time: `busy`), mailboxes are independent -- with a single thread the 1 250
// permutations of a wide table's opening set held up the shrink chains proved beside it on other contexts (44 instead of 14 ms).
struct HtService {
    std::mutex mu; std::condition_variable cv;
    std::vector<zk_host_transcript *> boxes;
    bool started = false;
    std::atomic<u64> pending{0};
    inline void run();
    void ensure() {                                   // under mu
        if (started) return;
        started = true;
        int n = getenv("ZK_TRANSCRIPT_HOST_THREADS") ? atoi(getenv("ZK_TRANSCRIPT_HOST_THREADS")) : 3;
        if (n < 1) n = 1;
        if (n > 16) n = 16;
        for (int i = 0; i < n; i++) std::thread([this] { run(); }).detach();
    }
};
inline HtService &ht_service() { static HtService *s = new HtService(); return *s; }     // leaked on purpose: the threads outlive static destructors

inline void ht_process(zk_host_transcript *H, HostJob &job) {
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

inline void HtService::run() {
    u32 idle = 0;
    for (;;) {
        zk_host_transcript *H = nullptr;
        HostJob job;
        {
            // pick a mailbox whose head job is ready, under the service lock: a mailbox that is being destroyed has left `boxes` and
            // cannot be picked any more (ht_mailbox_retire)
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [this] { return pending.load() != 0; });
            for (zk_host_transcript *B : boxes) {
                std::lock_guard<std::mutex> bl(B->mu);
                if (B->busy || B->q.empty()) continue;                                       // another thread is inside this mailbox's head job
                if (B->q.front().needs_gpu && ht_flag_load(B->gpu_done) < B->q.front().seq) continue;      // its words have not landed yet
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
        ht_process(H, job);                                        // (the acquire load of gpu_done above ordered the observed words before this)
        ht_flag_store(H->reply, job.seq);                          // ALWAYS published: a stream never waits for a value that will not come
        pending.fetch_sub(1);
        {
            std::lock_guard<std::mutex> bl(H->mu);                 // last touch of the mailbox: a waiter (retire, wait_idle) proceeds after the unlock
            H->busy = false;
            if (H->q.empty()) H->idle_cv.notify_all();
        }
    }
}

// ---- mailbox life cycle (memory for flags and ring comes from the caller: mapped pinned memory on the real path)
inline void ht_mailbox_publish(zk_host_transcript *H) {
    HtService &S = ht_service();
    std::lock_guard<std::mutex> lk(S.mu);
    S.boxes.push_back(H);
    S.ensure();
}
inline void ht_wait_idle(zk_host_transcript *H) {
    std::unique_lock<std::mutex> lk(H->mu);
    H->idle_cv.wait(lk, [H] { return H->q.empty() && !H->busy; });
}
// after this nobody can pick the mailbox any more; the caller has made sure that nothing more will be enqueued
inline void ht_mailbox_retire(zk_host_transcript *H) {
    ht_wait_idle(H);
    HtService &S = ht_service();
    { std::lock_guard<std::mutex> lk(S.mu); for (size_t i = 0; i < S.boxes.size(); i++) if (S.boxes[i] == H) { S.boxes.erase(S.boxes.begin() + i); break; } }
    { std::unique_lock<std::mutex> lk(H->mu); H->idle_cv.wait(lk, [H] { return !H->busy; }); }
}

#define HT_MAX_SEG 8
struct HtSegs { const u64 *src[HT_MAX_SEG]; u64 *dst[HT_MAX_SEG]; u32 n[HT_MAX_SEG]; u32 count; };

// One zk_transcript_run on the host path.  Dev supplies:
//   void sync();                                          everything enqueued on the stream so far has executed
//   void gather(const HtSegs &, u64 *flag, u64 seq);      stream op: copy the segments (device -> ring), then publish flag = seq
//   void wait_scatter(u64 *flag, u64 seq, const HtSegs &);stream op: wait until flag >= seq, then copy the segments (ring -> device)
//   int32_t fail(int32_t code, const char *msg);
// The pinned ring is bump-allocated and started over at every proof; a call that would not fit first waits until everything enqueued so far
// has been consumed (device and service thread) -- BEFORE it allocates anything, so a job's own buffers never move under it.
template <class Dev> int32_t host_transcript_enqueue(zk_host_transcript *H, Dev &dev, const DevChallenger *c, const TOpList &ops) {
    {
        size_t need = 0;
        for (int k = 0; k < ops.n; k++) need += (((size_t)ops.op[k].n + 16) * 8 + 63) & ~(size_t)63;
        if (need > H->ring_cap) return dev.fail(ZK_ERR_OOM, "host transcript: one call exceeds the pinned ring");
        if (H->ring_used + need > H->ring_cap) {
            dev.sync();
            ht_wait_idle(H);
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
            if (!in) return dev.fail(ZK_ERR_OOM, "host transcript: observed words exceed the pinned ring");
            if (!seg(gin, o.src, in, o.n)) return dev.fail(ZK_ERR_INVALID_ARG, "host transcript: too many segments in one call");
            h.in = in; job.needs_gpu = true;
            break;
        }
        case TOP_GET: case TOP_GET_INDICES: {
            if (!o.n) continue;
            u64 *out = ring(o.n); if (!out) return dev.fail(ZK_ERR_OOM, "host transcript: ring");
            h.out = out;
            if (!seg(gout, out, o.dst, o.n)) return dev.fail(ZK_ERR_INVALID_ARG, "host transcript: too many segments in one call");
            break;
        }
        case TOP_COMPACT:
            if (o.dst) { u64 *out = ring(12); if (!out) return dev.fail(ZK_ERR_OOM, "host transcript: ring"); h.out = out; if (!seg(gout, out, o.dst, 12)) return dev.fail(ZK_ERR_INVALID_ARG, "host transcript: segments"); }
            break;
        case TOP_POW_SETUP: { u64 *out = ring(14); if (!out) return dev.fail(ZK_ERR_OOM, "host transcript: ring"); h.out = out; if (!seg(gout, out, o.dst, 14)) return dev.fail(ZK_ERR_INVALID_ARG, "host transcript: segments"); break; }
        case TOP_POW_FINISH: {
            u64 *in = ring(1), *out = ring(1); if (!in || !out) return dev.fail(ZK_ERR_OOM, "host transcript: ring");
            if (!seg(gin, o.src + 13, in, 1) || !seg(gout, out, o.dst, 1)) return dev.fail(ZK_ERR_INVALID_ARG, "host transcript: segments");
            h.in = in; h.out = out; job.needs_gpu = true;
            break;
        }
        default: break;
        }
        job.ops.push_back(h);
    }
    const u64 seq = job.seq;
    if (job.needs_gpu) dev.gather(gin, H->gpu_done, seq);
    HtService &S = ht_service();
    { std::lock_guard<std::mutex> lk(H->mu); H->q.push_back(std::move(job)); }
    S.pending.fetch_add(1);
    { std::lock_guard<std::mutex> lk(S.mu); S.cv.notify_all(); }
    if (gout.count) dev.wait_scatter(H->reply, seq, gout);
    return ZK_OK;
}
