// Matrix-core Poseidon permutation (poseidon_mfma.cuh) against the vector-ALU one (poseidon.cuh) on random and edge states,
// then its throughput with the state in registers.  build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/psdm_test.hip -o tools/bin/psdm_test
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../eth-tx-proof_amd/csrc/poseidon_mfma.cuh"

// states[P][12]; wave w handles permutations 64w .. 64w+63
template <bool EXACT> __global__ __launch_bounds__(256) void k_new(const PsdmTables *T, u64 *states, u64 *flags) {
    const u32 lane = threadIdx.x & 63, n = lane & 31, h = lane >> 5;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    __shared__ psdm_v4i shA[PSDM_LDS_V4];
    psdm::lds_fill(T, shA); __syncthreads();
    psdm::perm<EXACT> pm; pm.init(T, lane, shA);
    u64 s[2][6];
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
        for (int i = 0; i < 6; i++) s[c][i] = states[(wave * 64 + 32 * c + n) * 12 + 6 * h + i];
    pm.run(s);
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
        for (int i = 0; i < 6; i++) states[(wave * 64 + 32 * c + n) * 12 + 6 * h + i] = s[c][i];
    if (flags) flags[wave * 64 + lane] = pm.pc.fl;
}
__global__ __launch_bounds__(256) void k_ref(u64 *states) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    u64 s[12];
    for (int k = 0; k < 12; k++) s[k] = states[12 * i + k];
    psd::permute(s);
    for (int k = 0; k < 12; k++) states[12 * i + k] = s[k];
}
// stage tests: 0 = MDS layer 7 (no constants), 1 = MDS layer 0 (+RC[1]), 2 = the 22 partial rounds
template <int STAGE> __global__ __launch_bounds__(256) void k_stage_new(const PsdmTables *T, u64 *states) {
    const u32 lane = threadIdx.x & 63, n = lane & 31, h = lane >> 5;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    __shared__ psdm_v4i shA[PSDM_LDS_V4];
    psdm::lds_fill(T, shA); __syncthreads();
    psdm::perm<true> pm; pm.init(T, lane, shA);
    u64 s[2][6];
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
        for (int i = 0; i < 6; i++) s[c][i] = states[(wave * 64 + 32 * c + n) * 12 + 6 * h + i];
    if (STAGE == 0) pm.mds(s, 7);
    if (STAGE == 1) pm.mds(s, 0);
    if (STAGE == 2) pm.partial_rounds(s);
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
        for (int i = 0; i < 6; i++) states[(wave * 64 + 32 * c + n) * 12 + 6 * h + i] = gl_canon(s[c][i]);
}
template <int STAGE> __global__ __launch_bounds__(256) void k_stage_ref(u64 *states) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    u64 s[12];
    for (int k = 0; k < 12; k++) s[k] = states[12 * i + k];
    psd::core<true> pc; pc.fl = 0;
    if (STAGE == 0) pc.mds_layer(s, psd::PSD_ZERO12);
    if (STAGE == 1) pc.mds_layer(s, PSD_RC + 12);
    if (STAGE == 2) { pc.partial_rounds(s); for (int k = 0; k < 12; k++) s[k] = psd::wadd_c(s[k], PSD_RC[12 * 26 + k]); }
    for (int k = 0; k < 12; k++) states[12 * i + k] = gl_canon(s[k]);
}
// the Merkle-level shape of the product kernels (poseidon_kernels.hip two_to_one_pair): eight words from the children, four zeros
#ifndef KLB
#define KLB 3
#endif
template <bool EXACT> __global__ __launch_bounds__(256, KLB) void k_level(const PsdmTables *T, const u64 *child, u64 *parent, u64 *flags) {
    const u32 lane = threadIdx.x & 63, n = lane & 31, h = lane >> 5;
    const size_t parent0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) & ~(size_t)63;
    __shared__ psdm_v4i shA[PSDM_LDS_V4];
    psdm::lds_fill(T, shA); __syncthreads();
    psdm::perm<EXACT> pm; pm.init(T, lane, shA);
    u64 s[2][6];
#pragma unroll
    for (int c = 0; c < 2; c++) {
        const u64 *p = child + 8 * (parent0 + 32 * c + n);
        if (h == 0) {
            const ulonglong2 a = ((const ulonglong2 *)p)[0], b = ((const ulonglong2 *)p)[1], d = ((const ulonglong2 *)p)[2];
            s[c][0] = a.x; s[c][1] = a.y; s[c][2] = b.x; s[c][3] = b.y; s[c][4] = d.x; s[c][5] = d.y;
        } else {
            const ulonglong2 e = ((const ulonglong2 *)p)[3];
            s[c][0] = e.x; s[c][1] = e.y; s[c][2] = 0; s[c][3] = 0; s[c][4] = 0; s[c][5] = 0;
        }
    }
    pm.run(s);
    if (lane < 32) {
#pragma unroll
        for (int c = 0; c < 2; c++) { u64 *o = parent + 4 * (parent0 + 32 * c + lane); o[0] = s[c][0]; o[1] = s[c][1]; o[2] = s[c][2]; o[3] = s[c][3]; }
    }
    if (flags) flags[parent0 + lane] = pm.pc.fl;
}
#ifndef LB
#define LB 2
#endif
#ifndef REPS
#define REPS 8
#endif
__global__ __launch_bounds__(256, LB) void k_time(const PsdmTables *T, u64 *out, u64 seed) {
    const u32 lane = threadIdx.x & 63;
    __shared__ psdm_v4i shA[PSDM_LDS_V4];
    psdm::lds_fill(T, shA); __syncthreads();
    psdm::perm<false> pm; pm.init(T, lane, shA);
    u64 s[2][6];
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
        for (int i = 0; i < 6; i++) s[c][i] = (seed + threadIdx.x * 12 + i + 6 * c + blockIdx.x * 4096) * 0x9E3779B97F4A7C15ULL % GL_P;
    const u64 c0 = clock64(), w0 = wall_clock64();          // shader cycles and the constant-rate counter: their ratio is the clock the kernel was given
    for (int r = 0; r < REPS; r++) pm.run(s);
    if (blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) { out[8] = clock64() - c0; out[9] = wall_clock64() - w0; }
    u64 x = pm.pc.fl;
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
        for (int i = 0; i < 6; i++) x ^= s[c][i];
    if (x == 0x1234567) out[0] = x;
}

static u64 splitmix(u64 &z) { z += 0x9E3779B97F4A7C15ULL; u64 x = z; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL; x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL; return x ^ (x >> 31); }

int main() {
    static PsdmTables tab; psdm_host::build(tab);
    PsdmTables *dT; hipMalloc(&dT, sizeof tab); hipMemcpy(dT, &tab, sizeof tab, hipMemcpyHostToDevice);
    const size_t P = 64 * 256;
    std::vector<u64> in(P * 12), a(P * 12), b(P * 12), fl(P);
    u64 z = 42;
    const u64 edge[8] = {0, 1, GL_P - 1, 0xFFFFFFFFULL, 1ULL << 32, 1ULL << 63, GL_P, ~0ULL};
    for (size_t i = 0; i < P * 12; i++) in[i] = (i < 12 * 64 * 4) ? edge[splitmix(z) & 7] : (i < 12 * 64 * 8 ? i / 12 + (i % 12 == 3) : splitmix(z));
    u64 *dA, *dB, *dF; hipMalloc(&dA, P * 96); hipMalloc(&dB, P * 96); hipMalloc(&dF, P * 8);
    int rc = 0;
    for (int st = 0; st < 3; st++) {
        hipMemcpy(dA, in.data(), P * 96, hipMemcpyHostToDevice); hipMemcpy(dB, in.data(), P * 96, hipMemcpyHostToDevice);
        if (st == 0) { hipLaunchKernelGGL(k_stage_new<0>, dim3(P / 256), dim3(256), 0, 0, dT, dA); hipLaunchKernelGGL(k_stage_ref<0>, dim3(P / 256), dim3(256), 0, 0, dB); }
        if (st == 1) { hipLaunchKernelGGL(k_stage_new<1>, dim3(P / 256), dim3(256), 0, 0, dT, dA); hipLaunchKernelGGL(k_stage_ref<1>, dim3(P / 256), dim3(256), 0, 0, dB); }
        if (st == 2) { hipLaunchKernelGGL(k_stage_new<2>, dim3(P / 256), dim3(256), 0, 0, dT, dA); hipLaunchKernelGGL(k_stage_ref<2>, dim3(P / 256), dim3(256), 0, 0, dB); }
        hipDeviceSynchronize();
        hipMemcpy(a.data(), dA, P * 96, hipMemcpyDeviceToHost); hipMemcpy(b.data(), dB, P * 96, hipMemcpyDeviceToHost);
        size_t bad = 0; size_t badw[12] = {0};
        for (size_t i = 0; i < P * 12; i++) if (a[i] != b[i]) { if (bad < 4) printf("  stage %d perm %zu word %zu: got %016llx want %016llx (in %016llx)\n", st, i / 12, i % 12, (unsigned long long)a[i], (unsigned long long)b[i], (unsigned long long)in[i]); bad++; badw[i % 12]++; }
        printf("stage %d: %s (%zu mismatches; per word:", st, bad ? "FAIL" : "OK", bad);
        for (int k = 0; k < 12; k++) printf(" %zu", badw[k]);
        printf(")\n");
    }
    for (int exact = 0; exact < 2; exact++) {
        hipMemcpy(dA, in.data(), P * 96, hipMemcpyHostToDevice); hipMemcpy(dB, in.data(), P * 96, hipMemcpyHostToDevice);
        if (exact) hipLaunchKernelGGL(k_new<true>, dim3(P / 256), dim3(256), 0, 0, dT, dA, dF);
        else hipLaunchKernelGGL(k_new<false>, dim3(P / 256), dim3(256), 0, 0, dT, dA, dF);
        hipLaunchKernelGGL(k_ref, dim3(P / 256), dim3(256), 0, 0, dB);
        hipError_t e = hipDeviceSynchronize();
        if (e != hipSuccess) { printf("HIP error %s\n", hipGetErrorString(e)); return 2; }
        hipMemcpy(a.data(), dA, P * 96, hipMemcpyDeviceToHost); hipMemcpy(b.data(), dB, P * 96, hipMemcpyDeviceToHost);
        hipMemcpy(fl.data(), dF, P * 8, hipMemcpyDeviceToHost);
        size_t bad = 0, flagged = 0;
        for (size_t i = 0; i < P; i++) {
            bool f = fl[i] != 0; flagged += f;
            for (int k = 0; k < 12; k++) if (a[12 * i + k] != b[12 * i + k]) {
                if (!f || exact) { if (bad < 6) printf("  mismatch perm %zu word %d: got %016llx want %016llx\n", i, k, (unsigned long long)a[12 * i + k], (unsigned long long)b[12 * i + k]); bad++; }
            }
        }
        printf("matrix-core permutation (%s products) vs psd::permute on %zu states: %s (%zu word mismatches, %zu flagged lanes)\n", exact ? "exact" : "fast", P, bad ? "FAIL" : "OK", bad, flagged);
        rc |= bad != 0;
    }
    {   // Merkle-level shape: P parents from 2P random canonical child digests
        std::vector<u64> ch(P * 8), st(P * 12), got(P * 4);
        for (size_t i = 0; i < P * 8; i++) ch[i] = splitmix(z) % GL_P;
        for (size_t i = 0; i < P; i++) for (int k = 0; k < 12; k++) st[12 * i + k] = k < 8 ? ch[8 * i + k] : 0;
        u64 *dC, *dP; hipMalloc(&dC, P * 64); hipMalloc(&dP, P * 32);
        hipMemcpy(dC, ch.data(), P * 64, hipMemcpyHostToDevice); hipMemcpy(dB, st.data(), P * 96, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_level<false>, dim3(P / 256), dim3(256), 0, 0, dT, dC, dP, dF);
        hipLaunchKernelGGL(k_ref, dim3(P / 256), dim3(256), 0, 0, dB);
        hipDeviceSynchronize();
        hipMemcpy(got.data(), dP, P * 32, hipMemcpyDeviceToHost); hipMemcpy(b.data(), dB, P * 96, hipMemcpyDeviceToHost); hipMemcpy(fl.data(), dF, P * 8, hipMemcpyDeviceToHost);
        size_t bad = 0; size_t by_lane[64] = {0};
        for (size_t i = 0; i < P; i++) for (int k = 0; k < 4; k++) if (got[4 * i + k] != b[12 * i + k] && !fl[i & ~(size_t)63]) {
            if (bad < 4) printf("  level parent %zu word %d: got %016llx want %016llx\n", i, k, (unsigned long long)got[4 * i + k], (unsigned long long)b[12 * i + k]);
            bad++; by_lane[i & 63]++;
        }
        printf("Merkle-level shape (two children + zero capacity) vs psd::permute on %zu parents: %s (%zu word mismatches)\n", P, bad ? "FAIL" : "OK", bad);
        if (bad) { printf("  mismatches by position in the wave:"); for (int k = 0; k < 64; k++) printf(" %zu", by_lane[k]); printf("\n"); }
        rc |= bad != 0;
    }
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    int cus = prop.multiProcessorCount;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int wl[] = {8, 4, 3, 2, 1};
    for (int w : wl) {
        int blocks = cus * w, rounds = 4 * (8 / w);
        hipLaunchKernelGGL(k_time, dim3(blocks), dim3(256), 0, 0, dT, dA, 1ull);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < rounds; r++) hipLaunchKernelGGL(k_time, dim3(blocks), dim3(256), 0, 0, dT, dA, (u64)r);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double perms = (double)blocks * 256 * REPS * rounds;
        u64 ck[2] = {0, 1}; hipMemcpy(ck, dA + 8, 16, hipMemcpyDeviceToHost);
        int wrate = 0; hipDeviceGetAttribute(&wrate, hipDeviceAttributeWallClockRate, 0);          // kHz
        printf("  blocks %5d (%d waves/SIMD launched): %8.3f ms/launch  %.3e perms/s   shader clock under the kernel %.2f GHz (%llu cycles of one wave for %d permutations)\n", blocks, w,
               ms / rounds, perms / (ms * 1e-3), (double)ck[0] / (double)ck[1] * wrate * 1e-6, (unsigned long long)ck[0], REPS);
    }
    return rc;
}
