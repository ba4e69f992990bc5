// GPU-less harness of the host-side challenger's mailbox / service-thread protocol (VERDICT r4 #4b).
// Built from the product's own header (eth-tx-proof_amd/csrc/transcript_host_core.h) with a plain C++ compiler, under ThreadSanitizer:
//     g++ -std=c++17 -O1 -g -fsanitize=thread -pthread -I eth-tx-proof_amd/csrc tests/cpp/transcript_host_tsan.cpp -o /tmp/ht_tsan
// A THREAD plays the device side of every context's stream: it executes the stream's operations in order -- "gather" copies the observed
// words into the ring and publishes gpu_done = seq (release), "wait + scatter" spins until reply >= seq (acquire) and copies the challenges
// out -- exactly what host_transcript_gather_kernel / hipStreamWaitValue64 / host_transcript_scatter_kernel do on the GPU.  The proving
// threads enqueue ahead of it through host_transcript_enqueue, as zk_transcript_run does.
// Checked: (1) every challenge, PoW state and compacted state equals a straight single-threaded run of the same script through
// psdh::Sponge; (2) the ring wraps (a small ring forces the drain-and-restart path in the middle of the stream of jobs); (3) mailboxes
// are created, retired and recreated while others are busy; (4) ThreadSanitizer sees no race and no lock-order inversion; (5) nothing
// stalls: a watchdog aborts with a dump of every mailbox if the run makes no progress for 20 s.
// usage: ht_tsan [contexts=6] [rounds=3] [jobs_per_round=400] [ring_bytes=65536]
#include <stdio.h>
#include <string.h>
#include <functional>
#include "transcript_host_core.h"

namespace {
struct Rng { u64 s; u64 next() { s += 0x9E3779B97F4A7C15ull; u64 z = s; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); } };

// the "stream" of one context: operations executed in order by the device thread
struct StreamOp { std::function<void()> fn; };
struct FakeDevice {
    std::mutex mu; std::condition_variable cv;
    std::deque<StreamOp> q;
    bool stop = false, idle = true;
    std::atomic<u64> executed{0};
    std::thread th;
    void start() { th = std::thread([this] { run(); }); }
    void run() {
        for (;;) {
            StreamOp op;
            {
                std::unique_lock<std::mutex> lk(mu);
                idle = q.empty();
                if (idle) cv.notify_all();
                cv.wait(lk, [this] { return stop || !q.empty(); });
                if (q.empty()) return;
                op = std::move(q.front()); q.pop_front();
                idle = false;
            }
            op.fn();
            executed.fetch_add(1);
        }
    }
    void push(std::function<void()> f) { { std::lock_guard<std::mutex> lk(mu); q.push_back({std::move(f)}); } cv.notify_all(); }
    void sync() { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [this] { return q.empty() && idle; }); }
    void join() { { std::lock_guard<std::mutex> lk(mu); stop = true; } cv.notify_all(); th.join(); }
};
// Dev policy of host_transcript_enqueue: the three device-side actions become stream operations of the fake device
struct FakeDev {
    FakeDevice *d;
    int32_t err = 0;
    void sync() { d->sync(); }
    void gather(const HtSegs &g, u64 *flag, u64 seq) {
        d->push([g, flag, seq] {
            for (u32 k = 0; k < g.count; k++) memcpy(g.dst[k], g.src[k], (size_t)g.n[k] * 8);
            ht_flag_store(flag, seq);
        });
    }
    void wait_scatter(u64 *flag, u64 seq, const HtSegs &g) {
        d->push([g, flag, seq] {
            while (ht_flag_load(flag) < seq) _mm_pause();
            for (u32 k = 0; k < g.count; k++) memcpy(g.dst[k], g.src[k], (size_t)g.n[k] * 8);
        });
    }
    int32_t fail(int32_t code, const char *msg) { fprintf(stderr, "enqueue failed: %d %s\n", code, msg); err = code; return code; }
};

struct Context {
    zk_host_transcript *H = nullptr;
    FakeDevice dev;
    u64 *flags = nullptr;
    void open(size_t ring_bytes) {
        H = new zk_host_transcript();
        flags = (u64 *)aligned_alloc(64, 256); memset(flags, 0, 256);
        H->gpu_done = flags; H->reply = flags + 16;
        H->ring = (char *)aligned_alloc(64, ring_bytes); H->ring_cap = ring_bytes;
        ht_mailbox_publish(H);
    }
    void close() {
        dev.sync();
        ht_mailbox_retire(H);
        free(H->ring); free(flags); delete H; H = nullptr;
    }
};
std::atomic<u64> g_progress{0};
std::atomic<int> g_failures{0};

// one proving thread: `rounds` "proofs" of `jobs` transcript calls each on one context; after every round the device-side results are
// compared with a plain sponge run of the same script
void prover(int id, int rounds, int jobs, size_t ring_bytes) {
    Context C;
    C.dev.start();
    Rng rng{(u64)id * 7919u + 1};
    for (int r = 0; r < rounds; r++) {
        C.open(ring_bytes);                       // a fresh mailbox per round: creation and retirement run beside the other contexts' traffic
        FakeDev dev{&C.dev};
        DevChallenger *ch[2] = {(DevChallenger *)(uintptr_t)(0x1000 + 64 * id), (DevChallenger *)(uintptr_t)(0x2000 + 64 * id)};     // map keys only
        psdh::Sponge ref[2];
        // "device memory": sources of observed words and destinations of challenges, one slab per job so that nothing is reused early
        std::vector<std::vector<u64>> srcs, dsts, want;
        srcs.reserve(jobs * 2); dsts.reserve(jobs * 2); want.reserve(jobs * 2);
        { TOpList a = {}; a.n = 1; a.op[0].kind = TOP_INIT; host_transcript_enqueue(C.H, dev, ch[0], a); host_transcript_enqueue(C.H, dev, ch[1], a); }
        for (int j = 0; j < jobs; j++) {
            const int w = (int)(rng.next() % 2);
            TOpList ops = {};
            const int nops = 1 + (int)(rng.next() % 3);
            for (int k = 0; k < nops; k++) {
                TOp &o = ops.op[ops.n];
                const u64 pick = rng.next() % 10;
                if (pick < 5) {                                        // observe a few words (sometimes a long opening set)
                    const u32 n = (u32)(pick == 0 ? 200 + rng.next() % 600 : 1 + rng.next() % 20);
                    srcs.emplace_back(n);
                    for (u32 i = 0; i < n; i++) { srcs.back()[i] = rng.next(); ref[w].observe(srcs.back()[i]); }
                    o.kind = TOP_OBSERVE; o.n = n; o.src = srcs.back().data();
                } else if (pick < 8) {                                 // squeeze
                    const u32 n = (u32)(1 + rng.next() % 12);
                    dsts.emplace_back(n, 0xDEADull); want.emplace_back(n);
                    for (u32 i = 0; i < n; i++) want.back()[i] = ref[w].get();
                    o.kind = TOP_GET; o.n = n; o.dst = dsts.back().data();
                } else if (pick == 8) {                                // query indices
                    const u32 n = (u32)(1 + rng.next() % 6), bits = 10;
                    dsts.emplace_back(n, 0xDEADull); want.emplace_back(n);
                    for (u32 i = 0; i < n; i++) want.back()[i] = ref[w].get() & ((1u << bits) - 1);
                    o.kind = TOP_GET_INDICES; o.n = n; o.aux = bits; o.dst = dsts.back().data();
                } else {                                               // compact, state read back
                    dsts.emplace_back(12, 0xDEADull); want.emplace_back(12);
                    if (ref[w].n_in) ref[w].duplex();
                    ref[w].n_out = 0;
                    for (int i = 0; i < 12; i++) want.back()[i] = ref[w].state[i];
                    o.kind = TOP_COMPACT; o.dst = dsts.back().data();
                }
                ops.n++;
            }
            if (host_transcript_enqueue(C.H, dev, ch[w], ops) != ZK_OK) { g_failures++; break; }
            g_progress.fetch_add(1);
        }
        C.dev.sync();                                                  // "hipStreamSynchronize" at the end of the proof
        ht_wait_idle(C.H);
        for (size_t i = 0; i < want.size(); i++)
            if (memcmp(want[i].data(), dsts[i].data(), want[i].size() * 8) != 0) { fprintf(stderr, "context %d round %d: result %zu differs\n", id, r, i); g_failures++; break; }
        if (C.H->error.load()) { fprintf(stderr, "context %d: host-side error %d\n", id, C.H->error.load()); g_failures++; }
        C.close();
    }
    C.dev.join();
}
}  // namespace

int main(int argc, char **argv) {
    const int contexts = argc > 1 ? atoi(argv[1]) : 6, rounds = argc > 2 ? atoi(argv[2]) : 3, jobs = argc > 3 ? atoi(argv[3]) : 400;
    const size_t ring_bytes = argc > 4 ? (size_t)atol(argv[4]) : 65536;
    std::atomic<bool> done{false};
    std::thread watchdog([&] {
        u64 last = ~0ull; int still = 0;
        while (!done.load()) {
            std::this_thread::sleep_for(std::chrono::milliseconds(500));
            const u64 now = g_progress.load();
            still = now == last ? still + 1 : 0;
            last = now;
            if (still >= 40) { fprintf(stderr, "STALL: no job enqueued for 20 s (%llu so far, %llu pending at the service)\n", (unsigned long long)now, (unsigned long long)ht_service().pending.load()); abort(); }
        }
    });
    std::vector<std::thread> th;
    for (int i = 0; i < contexts; i++) th.emplace_back(prover, i, rounds, jobs, ring_bytes);
    for (auto &t : th) t.join();
    done.store(true);
    watchdog.join();
    printf("%d contexts x %d rounds x %d transcript calls, ring %zu bytes: %s\n", contexts, rounds, jobs, ring_bytes, g_failures.load() ? "FAILED" : "ok");
    return g_failures.load() ? 1 : 0;
}
