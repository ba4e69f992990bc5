// Micro-benchmark of ONE radix-16 NTT stage in two formulations ("NTT: change the formulation"):
//   (V) the register butterfly network of the product kernel (csrc/ntt_radix.cuh: radix_net<4>, weak residues, shift twiddles):
//       one lane = one 16-point transform, 32 butterflies x 9 instructions + 17 shift twiddles;
//   (M) the same 16-point DFT as a dense contraction on the i8 matrix cores, with the byte-limb trick of poseidon_mfma.cuh: the DFT
//       matrix (entries w_16^(jk) = 2^(12jk)) times the bytes of the inputs, K = 16 points x 8 bytes = 4 k-steps, 16 outputs x 8 limb
//       sums = 4 row tiles -> 16 v_mfma_i32_32x32x32_i8 per 32 transforms, then the 10-instruction limb recombination per output word.
//       One wave = 32 transforms (lane (n, h) holds 8 inputs and gets 8 outputs of transform n).
// Both are checked against a host DFT, then timed on data that stays in registers (a chain of stages, so nothing but the stage
// itself is measured), with 1, 2 and 4 waves per SIMD.  Prints SIMD-cycles per element.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I eth-tx-proof_amd/csrc tools/ntt_mfma_stage.hip -o /tmp/ntt_mfma_stage ; run on the GPU box
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#include "gl64.cuh"
#include "ntt_radix.cuh"
#include "poseidon_mfma.cuh"

#define ITERS 256

// ---- tables of the matrix-core stage: 4 row tiles x 4 k-steps of operand fragments, accumulator starts per tile
struct StageTables { int8_t A[4][4][64][16]; int32_t Cin[4][2][16]; };
static void build_tables(StageTables &t) {
    const u64 w16 = gl_pow(gl_root(4), 1);
    // lane half h, k-step ks, word wsel: input point j = 4 ks + 2 h + wsel;  lane half hp, slot (2T + i): output point k = 2 slot + hp
    psdm_host::fill(4, 4, &t.A[0][0][0][0], &t.Cin[0][0][0],
                    [](int hp, int slot) { return slot < 8 ? 2 * slot + hp : -1; },
                    [](int ks, int h, int wsel) { return 4 * ks + 2 * h + wsel; },
                    [&](int k, int j) { return gl_pow(w16, (u64)((j * k) % 16)); }, [](int) { return (u64)0; });
}

// (M) one stage: s[2 ks + wsel] = point 4 ks + 2 h + wsel  ->  s[slot] = output 2 slot + h
__device__ __forceinline__ void stage_mfma(const psdm::rk &K, const psdm_v4i *__restrict__ A /* LDS or global: [T][ks][lane] */, const psdm_v16i (&cin)[4], u64 (&s)[8], u32 lane) {
    psdm_v4i b[4];
#pragma unroll
    for (int ks = 0; ks < 4; ks++) b[ks] = psdm::bfrag(s[2 * ks], s[2 * ks + 1]);
#pragma unroll
    for (int T = 0; T < 4; T++) {
        psdm_v16i acc = cin[T];
#pragma unroll
        for (int ks = 0; ks < 4; ks++) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[(T * 4 + ks) * 64 + lane], b[ks], acc, 0, 0, 0);
        psdm::recombine2(K, acc, s[2 * T], s[2 * T + 1]);
    }
}
__global__ __launch_bounds__(256) void mfma_check_kernel(const StageTables *tab, const u64 *in, u64 *out) {
    __shared__ psdm_v4i shA[16 * 64];
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (u32 i = threadIdx.x; i < 16 * 64; i += blockDim.x) shA[i] = ((const psdm_v4i *)&tab->A[0][0][0][0])[i];
    __syncthreads();
    psdm_v16i cin[4];
    for (int T = 0; T < 4; T++) cin[T] = psdm::ld16(&tab->Cin[T][lane >> 5][0]);
    psdm::rk K; K.init();
    const u32 n = (blockIdx.x * 4 + wave) * 32 + (lane & 31), h = lane >> 5;
    u64 s[8];
    for (int ks = 0; ks < 4; ks++) for (int w = 0; w < 2; w++) s[2 * ks + w] = in[(size_t)n * 16 + 4 * ks + 2 * h + w];
    stage_mfma(K, shA, cin, s, lane);
    for (int slot = 0; slot < 8; slot++) out[(size_t)n * 16 + 2 * slot + h] = gl_canon(s[slot]);
}
template <int WPS> __global__ __launch_bounds__(64 * 4 * WPS) void mfma_time_kernel(const StageTables *tab, u64 *sink, u64 seed) {
    __shared__ psdm_v4i shA[16 * 64];
    const u32 lane = threadIdx.x & 63;
    for (u32 i = threadIdx.x; i < 16 * 64; i += blockDim.x) shA[i] = ((const psdm_v4i *)&tab->A[0][0][0][0])[i];
    __syncthreads();
    psdm_v16i cin[4];
    for (int T = 0; T < 4; T++) cin[T] = psdm::ld16(&tab->Cin[T][lane >> 5][0]);
    psdm::rk K; K.init();
    u64 s[8];
    for (int i = 0; i < 8; i++) s[i] = (seed + threadIdx.x * 0x9E3779B97F4A7C15ULL + i * 0xD1B54A32D192ED03ULL + blockIdx.x) | 1;
#pragma unroll 1
    for (int it = 0; it < ITERS; it++) stage_mfma(K, shA, cin, s, lane);
    u64 r = 0;
    for (int i = 0; i < 8; i++) r ^= s[i];
    if (r == 0x1234567) sink[0] = r;
}

// (V) the product's register network: one lane = one transform
__global__ __launch_bounds__(256) void valu_check_kernel(const u64 *in, u64 *out, const u64 *tw) {
    const u32 n = blockIdx.x * 256 + threadIdx.x;
    u64 x[16], tw16[8];
    for (int j = 0; j < 8; j++) tw16[j] = tw[j];
    for (int j = 0; j < 16; j++) x[j] = in[(size_t)n * 16 + j];
    glx<false> g; g.fl = 0;
    radix_net<4, false, false>(g, x, 0, tw16);               // DIF: natural in, bit-reversed out
    for (int j = 0; j < 16; j++) out[(size_t)n * 16 + BR<4>::rev(j)] = gl_canon(x[j]);
    if (g.fl) out[(size_t)n * 16] = ~0ull;                    // a flagged rare repair would need the exact redo: report it
}
template <int WPS> __global__ __launch_bounds__(64 * 4 * WPS) void valu_time_kernel(u64 *sink, u64 seed) {
    u64 x[16], tw16[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 16; i++) x[i] = (seed + threadIdx.x * 0x9E3779B97F4A7C15ULL + i * 0xD1B54A32D192ED03ULL + blockIdx.x) | 1;
    // a fresh sticky word per stage, folded into an ordinary variable: glx<false>'s SGPR-pair asm operand is never carried around a loop
    // (a generated kernel that did so faulted)
    u64 flv = 0;
#pragma unroll 1
    for (int it = 0; it < ITERS; it++) { glx<false> g; g.fl = 0; radix_net<4, false, false>(g, x, 0, tw16); flv |= g.fl; }
    u64 r = flv;
    for (int i = 0; i < 16; i++) r ^= x[i];
    if (r == 0x1234567) sink[0] = r;
}

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(_e)); return 1; } } while (0)
template <class F> static double time_ms(F launch) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    launch(); hipDeviceSynchronize();
    hipEventRecord(a); for (int i = 0; i < 5; i++) launch(); hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    return ms / 5;
}
int main() {
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    const int cus = pr.multiProcessorCount; const double ghz = pr.clockRate * 1e-6;
    printf("device %s, %d CUs, %.2f GHz\n", pr.gcnArchName, cus, ghz);
    StageTables *ht = new StageTables; build_tables(*ht);
    StageTables *dt; CK(hipMalloc(&dt, sizeof *dt)); CK(hipMemcpy(dt, ht, sizeof *ht, hipMemcpyHostToDevice));
    // ---- correctness: 1024 random transforms (+ edge values) against a host DFT
    const int NT = 1024;
    std::vector<u64> in((size_t)NT * 16), ref((size_t)NT * 16), o1(in.size()), o2(in.size());
    u64 st = 0x243F6A8885A308D3ULL;
    for (auto &v : in) { st = st * 6364136223846793005ULL + 1442695040888963407ULL; v = (st ^ (st >> 29)) % GL_P; }
    const u64 edge[8] = {0, 1, GL_P - 1, 0xFFFFFFFFULL, 0x100000000ULL, 1ULL << 63, GL_P - 2, 0xFFFFFFFF00000000ULL};
    for (int i = 0; i < 16; i++) in[i] = edge[i & 7];
    const u64 w16 = gl_root(4);
    for (int n = 0; n < NT; n++) for (int k = 0; k < 16; k++) { u64 a = 0; for (int j = 0; j < 16; j++) a = gl_add(a, gl_mul(in[(size_t)n * 16 + j], gl_pow(w16, (u64)((j * k) % 16)))); ref[(size_t)n * 16 + k] = a; }
    u64 *din, *dout, *dtw, *sink; CK(hipMalloc(&din, in.size() * 8)); CK(hipMalloc(&dout, in.size() * 8)); CK(hipMalloc(&dtw, 64)); CK(hipMalloc(&sink, 64));
    CK(hipMemcpy(din, in.data(), in.size() * 8, hipMemcpyHostToDevice));
    u64 tw[8]; for (int j = 0; j < 8; j++) tw[j] = gl_pow(w16, j);
    CK(hipMemcpy(dtw, tw, 64, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(mfma_check_kernel, dim3(NT / 128), dim3(256), 0, 0, dt, din, dout); CK(hipMemcpy(o1.data(), dout, in.size() * 8, hipMemcpyDeviceToHost));
    hipLaunchKernelGGL(valu_check_kernel, dim3(NT / 256), dim3(256), 0, 0, din, dout, dtw); CK(hipMemcpy(o2.data(), dout, in.size() * 8, hipMemcpyDeviceToHost));
    size_t bad1 = 0, bad2 = 0;
    for (size_t i = 0; i < ref.size(); i++) { bad1 += o1[i] != ref[i]; bad2 += o2[i] != ref[i]; }
    printf("radix-16 stage vs host DFT: matrix-core %zu wrong of %zu, register network %zu wrong\n", bad1, ref.size(), bad2);
    if (bad1 || bad2) return 2;
    // ---- timing: every SIMD of the chip busy, data in registers
    printf("%-44s %10s %14s %18s\n", "formulation", "waves/SIMD", "ms", "SIMD-cycles/element");
    auto report = [&](const char *name, int wps, double ms, double elems_per_wave) {
        const double waves = (double)cus * 4 * wps * 8;       // 8 workgroups per CU-slot, see the grids below
        const double cyc = ms * 1e-3 * ghz * 1e9 * (cus * 4) / (waves * elems_per_wave * ITERS);
        printf("%-44s %10d %14.3f %18.3f\n", name, wps, ms, cyc);
    };
#define RUN(WPS) \
    report("(M) matrix cores + limb recombination", WPS, time_ms([&] { hipLaunchKernelGGL(mfma_time_kernel<WPS>, dim3(cus * 8), dim3(64 * 4 * WPS), 0, 0, dt, sink, 7ull); }), 32 * 16); \
    report("(V) register butterfly network", WPS, time_ms([&] { hipLaunchKernelGGL(valu_time_kernel<WPS>, dim3(cus * 8), dim3(64 * 4 * WPS), 0, 0, sink, 7ull); }), 64 * 16);
    RUN(1) RUN(2) RUN(4)
    printf("(M): 16 v_mfma_i32_32x32x32_i8 + 8 recombinations (10 vector instructions each) + 16 operand XORs per 512 elements; (V): 32 butterflies x 9 + 17 shift twiddles per 16 elements\n");
    return 0;
}
