// tools/host_roundtrip.hip -- what a HOST-side Fiat-Shamir challenger would pay per synchronisation point.
// SURVEY 8a row a8 placed the challenger on the host (a Poseidon permutation costs a core ~1 us against 7.5 us for the one-wave kernel); the
// price is one GPU -> host -> GPU round trip per challenge round.  This measures that round trip with no hipStreamSynchronize in it:
//   the stream holds, enqueued AHEAD:   kernel A_i (writes i to a flag in mapped pinned memory)  ->  hipStreamWaitValue64(reply >= i)  ->  kernel B_i
//   the host thread, per round i:       spins on the flag until it reads i,  (would absorb / squeeze here),  writes reply = i
// and, for reference, the same chain of kernels without the wait.  Per-round cost of the host challenger = (with - without) + its permutations.
// Build: hipcc -O2 --offload-arch=gfx950 tools/host_roundtrip.hip -o tools/bin/host_roundtrip      Run: tools/bin/host_roundtrip [rounds=2000]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <chrono>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void signal_kernel(volatile uint64_t *flag, uint64_t v, uint64_t *sink) {
    if (threadIdx.x == 0) { sink[0] += v; __threadfence_system(); *flag = v; __threadfence_system(); }
}
__global__ void consume_kernel(const volatile uint64_t *reply, uint64_t *sink) { if (threadIdx.x == 0) sink[1] += *reply; }

int main(int argc, char **argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 2000;
    uint64_t *flag, *reply, *sink;
    CK(hipHostMalloc((void **)&flag, 64, hipHostMallocMapped | hipHostMallocCoherent));
    hipError_t e;
    const bool sigmem = false;
    CK(hipHostMalloc((void **)&reply, 64, hipHostMallocMapped | hipHostMallocCoherent));
    CK(hipMalloc(&sink, 64)); CK(hipMemset(sink, 0, 64));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    // reference: the kernel chain alone
    *flag = 0;
    *reply = 0;
    CK(hipDeviceSynchronize());
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 1; i <= rounds; i++) {
        hipLaunchKernelGGL(signal_kernel, dim3(1), dim3(64), 0, st, flag, (uint64_t)i, sink);
        hipLaunchKernelGGL(consume_kernel, dim3(1), dim3(64), 0, st, reply, sink);
    }
    CK(hipStreamSynchronize(st));
    const double base_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / rounds;
    // with the host in the loop
    *flag = 0;
    CK(hipDeviceSynchronize());
    const int AHEAD = 64;                    // rounds enqueued ahead of the host side
    int enq = 0;
    auto push = [&](int i) -> hipError_t {
        hipLaunchKernelGGL(signal_kernel, dim3(1), dim3(64), 0, st, flag, (uint64_t)i, sink);
        hipError_t r = hipStreamWaitValue64(st, reply, (uint64_t)i, hipStreamWaitValueGte, ~0ull);
        if (r != hipSuccess) return r;
        hipLaunchKernelGGL(consume_kernel, dim3(1), dim3(64), 0, st, reply, sink);
        return hipSuccess;
    };
    t0 = std::chrono::steady_clock::now();
    double spin_us = 0;
    for (int i = 1; i <= rounds; i++) {
        while (enq < rounds && enq < i + AHEAD) { e = push(++enq); if (e != hipSuccess) { fprintf(stderr, "hipStreamWaitValue64: %s (signal memory %d)\n", hipGetErrorString(e), (int)sigmem); return 2; } }
        auto s0 = std::chrono::steady_clock::now();
        bool stuck = false;
        while (*(volatile uint64_t *)flag < (uint64_t)i) {
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - s0).count() > 2.0) { stuck = true; break; }
        }
        spin_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - s0).count();
        if (stuck) {                 // never leave a wait behind: release every enqueued round, drain, report
            *(volatile uint64_t *)reply = ~0ull >> 1; __sync_synchronize();
            hipStreamSynchronize(st);
            fprintf(stderr, "host_roundtrip: round %d: the flag never arrived (mapped memory not visible?)\n", i);
            return 3;
        }
        *(volatile uint64_t *)reply = (uint64_t)i; __sync_synchronize();
    }
    CK(hipStreamSynchronize(st));
    const double loop_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / rounds;
    printf("host_roundtrip: %d rounds; kernel chain alone %.2f us/round; with the host in the loop %.2f us/round (host spinning %.2f us of it); "
           "round trip = %.2f us; reply in %s\n", rounds, base_us, loop_us, spin_us / rounds, loop_us - base_us, sigmem ? "signal memory" : "mapped pinned memory");
    return 0;
}
