// Micro-benchmark of the Poseidon permutation core (poseidon.cuh) in isolation: permutations/s with the state in VGPRs and
// no memory traffic, at several grid sizes (a power- or clock-limited chip shows up as per-CU throughput that rises when
// fewer CUs are busy).  build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/psd_ubench.hip -o tools/bin/psd_ubench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#ifdef PSD_ALT
#include "poseidon.cuh"
#else
#include "../eth-tx-proof_amd/csrc/poseidon.cuh"
#endif

#ifndef REPS
#define REPS 8
#endif

__global__ __launch_bounds__(256) void k_perm(u64 *out, u64 seed) {
    u64 s[12];
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = (seed + threadIdx.x * 12 + i + blockIdx.x * 4096) * 0x9E3779B97F4A7C15ULL % GL_P;
    psd::core<false> pc; pc.fl = 0;
    for (int r = 0; r < REPS; r++) pc.run(s);
    u64 x = pc.fl;
#pragma unroll
    for (int i = 0; i < 12; i++) x ^= s[i];
    if (x == 0x1234567) out[0] = x;
}

int main(int argc, char **argv) {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    int cus = prop.multiProcessorCount;
    u64 *d; hipMalloc(&d, 256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("%s: %d CUs\n", argc > 1 ? argv[1] : "variant", cus);
    int waves_per_simd[] = {8, 4, 2, 1};
    for (int w : waves_per_simd) {
        for (int frac = 1; frac <= 4; frac *= 2) {
            int blocks = cus * w / frac;                 // one 256-thread block = one wave on each SIMD of a CU
            int rounds = 4 * frac * (8 / w);
            hipLaunchKernelGGL(k_perm, dim3(blocks), dim3(256), 0, 0, d, 1ull);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int r = 0; r < rounds; r++) hipLaunchKernelGGL(k_perm, dim3(blocks), dim3(256), 0, 0, d, (u64)r);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double perms = (double)blocks * 256 * REPS * rounds;
            printf("  blocks %5d (%d resident waves/SIMD on %3d%% of the CUs): %8.3f ms/launch  %.3e perms/s  %.3e per CU\n", blocks, w, 100 / frac,
                   ms / rounds, perms / (ms * 1e-3), perms / (ms * 1e-3) / (cus / frac));
        }
    }
    return 0;
}
