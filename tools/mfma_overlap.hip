// tools/mfma_overlap.hip -- can vector instructions run in the shadow of an i8 matrix instruction on gfx950, and which ones?
// One loop body = ONE v_mfma_i32_32x32x32_i8 (its own accumulator, no reader) + N independent vector instructions of one kind, for
// N = 0, 4, 8, 12, 16, 24, 32, 48; kinds: v_mad_u64_u32 (what Poseidon's s-boxes are made of), v_add_u32, v_lshl_add_u32, v_perm_b32.
// If the fillers hide, cycles per iteration stay at the instruction's own time (64) until their issue time exceeds it; if they do not,
// cycles grow from N = 1.  Reported per kind and N: shader cycles per iteration for one wave per SIMD and for two (the second wave is the
// question "does ANOTHER wave's vector work hide it").  A second table runs the fillers alone (no matrix instruction).
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/mfma_overlap.hip -o tools/bin/mfma_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef unsigned long long u64;
typedef unsigned int u32;

// the shape the Poseidon kernel has: the matrix instruction's RESULT is read right away (the compiler pads the read with s_nop wait
// states), then N v_mad_u64_u32 that do not depend on it.  DEP: 0 = no reader (as above), 1 = one compiler-generated reader of acc[0]
template <int N, int DEP> __global__ __launch_bounds__(256) void kdep(u64 *out, int iters, u32 seed) {
    v4i a = {(int)seed, 1, 2, 3}, b = {4, 5, (int)threadIdx.x, 7};
    u64 m[8]; u32 x[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { m[i] = seed + i + threadIdx.x; x[i] = seed * 3 + i; }
    u32 sink = 0;
    const u64 c0 = clock64();
    for (int it = 0; it < iters; it++) {
        v16i acc = {};
        acc[0] = (int)x[0];
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc, 0, 0, 0);
        if (DEP) sink += (u32)acc[0] + (u32)acc[15];
        else asm volatile("" : : "v"(acc));
#pragma unroll
        for (int j = 0; j < N; j++) { const int q = j & 7; asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(m[q]) : "v"(x[q]), "v"(x[(q + 1) & 7]) : "vcc"); }
    }
    const u64 c1 = clock64();
    u64 s = sink;
#pragma unroll
    for (int i = 0; i < 8; i++) s += m[i] + x[i];
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = c1 - c0; out[1] = s; }
}
template <int N, int DEP> static double rundep(u64 *d, int waves_per_simd, int cus) {
    const int iters = 4000;
    for (int r = 0; r < 2; r++) { hipLaunchKernelGGL((kdep<N, DEP>), dim3(cus * waves_per_simd), dim3(256), 0, 0, d, iters, 7u + r); hipDeviceSynchronize(); }
    u64 h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    return (double)h[0] / iters;
}

template <int KIND, int N, bool MFMA> __global__ __launch_bounds__(256) void k(u64 *out, int iters, u32 seed) {
    v4i a = {(int)seed, 1, 2, 3}, b = {4, 5, (int)threadIdx.x, 7};
    v16i acc = {};
    u64 m[8]; u32 x[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { m[i] = seed + i + threadIdx.x; x[i] = seed * 3 + i; }
    const u64 c0 = clock64();
    for (int it = 0; it < iters; it++) {
        if (MFMA) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < N; j++) {
            const int q = j & 7;
            if (KIND == 0) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(m[q]) : "v"(x[q]), "v"(x[(q + 1) & 7]) : "vcc");
            if (KIND == 1) asm volatile("v_add_u32 %0, %1, %0" : "+v"(x[q]) : "v"(x[(q + 3) & 7]));
            if (KIND == 2) asm volatile("v_lshl_add_u32 %0, %1, 8, %0" : "+v"(x[q]) : "v"(x[(q + 3) & 7]));
            if (KIND == 3) asm volatile("v_perm_b32 %0, %1, %0, %2" : "+v"(x[q]) : "v"(x[(q + 3) & 7]), "v"(x[(q + 5) & 7]));
        }
    }
    const u64 c1 = clock64();
    u64 s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += m[i] + x[i];
#pragma unroll
    for (int i = 0; i < 16; i++) s += (u32)acc[i];
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = c1 - c0; out[1] = s; }
}

template <int KIND, int N, bool MFMA> static double run(u64 *d, int waves_per_simd, int cus) {
    const int iters = 4000;
    hipLaunchKernelGGL((k<KIND, N, MFMA>), dim3(cus * waves_per_simd), dim3(256), 0, 0, d, iters, 7u);
    hipDeviceSynchronize();
    hipLaunchKernelGGL((k<KIND, N, MFMA>), dim3(cus * waves_per_simd), dim3(256), 0, 0, d, iters, 9u);
    hipDeviceSynchronize();
    u64 h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    return (double)h[0] / iters;
}
template <int KIND> static void table(const char *name, u64 *d, int cus) {
    printf("%-16s %8s %8s %8s %8s %8s %8s %8s %8s\n", name, "N=0", "4", "8", "12", "16", "24", "32", "48");
    for (int mf = 1; mf >= 0; mf--)
        for (int w = 1; w <= 2; w++) {
            printf("  %s, %d wave%s/SIMD:", mf ? "with the matrix instruction" : "fillers alone            ", w, w > 1 ? "s" : " ");
            if (mf) printf(" %7.1f %7.1f %7.1f %7.1f %7.1f %7.1f %7.1f %7.1f\n", run<KIND, 0, true>(d, w, cus), run<KIND, 4, true>(d, w, cus), run<KIND, 8, true>(d, w, cus), run<KIND, 12, true>(d, w, cus),
                           run<KIND, 16, true>(d, w, cus), run<KIND, 24, true>(d, w, cus), run<KIND, 32, true>(d, w, cus), run<KIND, 48, true>(d, w, cus));
            else printf(" %7.1f %7.1f %7.1f %7.1f %7.1f %7.1f %7.1f %7.1f\n", run<KIND, 0, false>(d, w, cus), run<KIND, 4, false>(d, w, cus), run<KIND, 8, false>(d, w, cus), run<KIND, 12, false>(d, w, cus),
                        run<KIND, 16, false>(d, w, cus), run<KIND, 24, false>(d, w, cus), run<KIND, 32, false>(d, w, cus), run<KIND, 48, false>(d, w, cus));
        }
}
int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    u64 *d; hipMalloc(&d, 64);
    printf("# shader cycles per loop iteration (one v_mfma_i32_32x32x32_i8 + N independent vector instructions), wave 0 of block 0; %d CUs\n", prop.multiProcessorCount);
    table<0>("v_mad_u64_u32", d, prop.multiProcessorCount);
    table<1>("v_add_u32", d, prop.multiProcessorCount);
    table<2>("v_lshl_add_u32", d, prop.multiProcessorCount);
    table<3>("v_perm_b32", d, prop.multiProcessorCount);
    printf("# one matrix instruction whose result is %s, then N independent v_mad_u64_u32: cycles per iteration at 1 / 2 / 4 / 8 waves per SIMD\n", "read at once (DEP 1) or not read (DEP 0)");
    for (int dep = 0; dep < 2; dep++) {
        printf("  DEP %d  N=16:", dep);
        for (int w : {1, 2, 4, 8}) printf(" %7.1f", dep ? rundep<16, 1>(d, w, prop.multiProcessorCount) : rundep<16, 0>(d, w, prop.multiProcessorCount));
        printf("   N=48:");
        for (int w : {1, 2, 4, 8}) printf(" %7.1f", dep ? rundep<48, 1>(d, w, prop.multiProcessorCount) : rundep<48, 0>(d, w, prop.multiProcessorCount));
        printf("\n");
    }
    printf("  fillers alone N=16:");
    for (int w : {1, 2, 4, 8}) printf(" %7.1f", run<0, 16, false>(d, w, prop.multiProcessorCount));
    printf("   N=48:");
    for (int w : {1, 2, 4, 8}) printf(" %7.1f", run<0, 48, false>(d, w, prop.multiProcessorCount));
    printf("\n");
    return 0;
}
