// Integer-ALU micro-benchmark for gfx950: issue rate of the instructions a Goldilocks modmul / Poseidon round is
// built from.  Poseidon hashing is integer-ALU-bound, so these rates -- not HBM -- are its roofline.
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench_int.hip -o /tmp/ubench_int ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../eth-tx-proof_amd/csrc/gl64.cuh"

#define ITERS 2048
#define CHAINS 8

#define DEF_KERNEL32(NAME, ASM)                                                        \
    __global__ void NAME(uint32_t *out, uint32_t seed) {                              \
        uint32_t a[CHAINS], b = seed | 1, c = seed ^ 0x9E3779B9u;                     \
        for (int i = 0; i < CHAINS; i++) a[i] = threadIdx.x * 2654435761u + i + seed; \
        for (int it = 0; it < ITERS; it++) {                                          \
            _Pragma("unroll") for (int i = 0; i < CHAINS; i++) { ASM; }              \
        }                                                                             \
        uint32_t r = 0;                                                               \
        for (int i = 0; i < CHAINS; i++) r ^= a[i];                                   \
        if (r == 0x12345) out[0] = r;                                                 \
    }
#define DEF_KERNEL64(NAME, ASM)                                                        \
    __global__ void NAME(uint32_t *out, uint32_t seed) {                              \
        uint64_t a[CHAINS]; uint32_t b = seed | 1, c = seed ^ 0x9E3779B9u;            \
        uint64_t b64 = ((uint64_t)b << 32) | c;                                       \
        for (int i = 0; i < CHAINS; i++) a[i] = (uint64_t)(threadIdx.x * 2654435761u + i + seed) * 0x9E3779B97F4A7C15ULL; \
        for (int it = 0; it < ITERS; it++) {                                          \
            _Pragma("unroll") for (int i = 0; i < CHAINS; i++) { ASM; }              \
        }                                                                             \
        uint64_t r = 0;                                                               \
        for (int i = 0; i < CHAINS; i++) r ^= a[i];                                   \
        if (r == 0x12345 + b64) out[0] = (uint32_t)r;                                 \
    }

DEF_KERNEL32(k_add_u32, asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[i]) : "v"(b)))
DEF_KERNEL32(k_mul_lo_u32, asm volatile("v_mul_lo_u32 %0, %1, %0" : "+v"(a[i]) : "v"(b)))
DEF_KERNEL32(k_mul_hi_u32, asm volatile("v_mul_hi_u32 %0, %1, %0" : "+v"(a[i]) : "v"(b)))
DEF_KERNEL32(k_mad_u32_u24, asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c)))
DEF_KERNEL32(k_mul_u32_u24, asm volatile("v_mul_u32_u24 %0, %1, %0" : "+v"(a[i]) : "v"(b)))
DEF_KERNEL32(k_mul_hi_u32_u24, asm volatile("v_mul_hi_u32_u24 %0, %1, %0" : "+v"(a[i]) : "v"(b)))
DEF_KERNEL32(k_dot4_u32_u8, asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c)))
DEF_KERNEL32(k_dot2_u32_u16, asm volatile("v_dot2_u32_u16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c)))
DEF_KERNEL32(k_perm_b32, asm volatile("v_perm_b32 %0, %1, %0, %2" : "+v"(a[i]) : "v"(b), "v"(c)))
DEF_KERNEL32(k_alignbit, asm volatile("v_alignbit_b32 %0, %1, %0, 7" : "+v"(a[i]) : "v"(b)))
DEF_KERNEL32(k_add3_u32, asm volatile("v_add3_u32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c)))
DEF_KERNEL32(k_addc_pair, asm volatile("v_add_co_u32 %0, vcc, %1, %0\n\tv_addc_co_u32 %0, vcc, %2, %0, vcc" : "+v"(a[i]) : "v"(b), "v"(c) : "vcc"))
DEF_KERNEL32(k_mov_b32, asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(b)))
DEF_KERNEL32(k_and_b32, asm volatile("v_and_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b)))
DEF_KERNEL32(k_lshlrev_b32, asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(a[i])))
DEF_KERNEL32(k_cndmask_vcc, asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc"))
DEF_KERNEL32(k_cndmask_e64, asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[10:11]" : "+v"(a[i]) : "v"(b) : "s10", "s11"))
DEF_KERNEL32(k_add_co_e32, asm volatile("v_add_co_u32 %0, vcc, %1, %0" : "+v"(a[i]) : "v"(b) : "vcc"))
DEF_KERNEL32(k_sub_u32, asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b)))
DEF_KERNEL32(k_xor_b32, asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b)))
DEF_KERNEL32(k_snop_add, asm volatile("s_nop 1\n\tv_add_u32 %0, %1, %0" : "+v"(a[i]) : "v"(b)))
DEF_KERNEL64(k_cmp_lt_u64, asm volatile("v_cmp_lt_u64 vcc, %0, %1" : : "v"(a[i]), "v"(b64) : "vcc"))
DEF_KERNEL64(k_mad_u64_u32, asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c) : "vcc"))
DEF_KERNEL64(k_lshl_add_u64, asm volatile("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(a[i]) : "v"(b64)))
DEF_KERNEL64(k_lshlrev_b64, asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(a[i])))
DEF_KERNEL64(k_fma_f64, asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b64)))
DEF_KERNEL64(k_mul_f64, asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b64)))
DEF_KERNEL64(k_cmp_cndmask64, { uint64_t t = a[i] + b64; a[i] = t < b64 ? t + 0xFFFFFFFFULL : t; })
DEF_KERNEL64(k_gl_mul, a[i] = gl_mul(a[i], b64 | 1))
DEF_KERNEL64(k_gl_sqr, a[i] = gl_mul(a[i], a[i]))
DEF_KERNEL64(k_gl_add, a[i] = gl_add(a[i], b64 >> 1))
DEF_KERNEL64(k_gl_mul_add, a[i] = gl_mul_add(a[i], b64 | 1, c))
DEF_KERNEL64(k_umul64hi, a[i] = __umul64hi(a[i], b64) + a[i] * b64)

typedef void (*kern_t)(uint32_t *, uint32_t);
struct Entry { const char *name; kern_t k; int insts_per_op; };

int main() {
    Entry es[] = {
        {"v_add_u32", k_add_u32, 1}, {"v_add3_u32", k_add3_u32, 1}, {"v_mul_lo_u32", k_mul_lo_u32, 1}, {"v_mul_hi_u32", k_mul_hi_u32, 1},
        {"v_mad_u32_u24", k_mad_u32_u24, 1}, {"v_mul_u32_u24", k_mul_u32_u24, 1}, {"v_mul_hi_u32_u24", k_mul_hi_u32_u24, 1},
        {"v_dot4_u32_u8", k_dot4_u32_u8, 1}, {"v_dot2_u32_u16", k_dot2_u32_u16, 1}, {"v_perm_b32", k_perm_b32, 1}, {"v_alignbit_b32", k_alignbit, 1},
        {"v_mov_b32", k_mov_b32, 1}, {"v_and_b32", k_and_b32, 1}, {"v_xor_b32", k_xor_b32, 1}, {"v_sub_u32", k_sub_u32, 1}, {"v_lshlrev_b32", k_lshlrev_b32, 1},
        {"v_cndmask_b32 (vcc)", k_cndmask_vcc, 1}, {"v_cndmask_b32_e64 (sgpr)", k_cndmask_e64, 1}, {"v_add_co_u32 (vcc out)", k_add_co_e32, 1},
        {"s_nop 1 + v_add_u32", k_snop_add, 1}, {"v_cmp_lt_u64", k_cmp_lt_u64, 1},
        {"v_add_co+v_addc (64b add)", k_addc_pair, 1}, {"v_mad_u64_u32", k_mad_u64_u32, 1}, {"v_lshl_add_u64", k_lshl_add_u64, 1},
        {"v_lshlrev_b64", k_lshlrev_b64, 1}, {"v_fma_f64", k_fma_f64, 1}, {"v_mul_f64", k_mul_f64, 1},
        {"u64 add + carry fix (C++)", k_cmp_cndmask64, 1}, {"__umul64hi + mul (C++)", k_umul64hi, 1},
        {"gl_mul (C++)", k_gl_mul, 1}, {"gl_sqr (C++)", k_gl_sqr, 1}, {"gl_add (C++)", k_gl_add, 1}, {"gl_mul_add (C++)", k_gl_mul_add, 1},
    };
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    int cus = prop.multiProcessorCount;
    double clk = prop.clockRate * 1e3;
    printf("device %s CUs %d clock %.0f MHz\n", prop.name, cus, clk / 1e6);
    uint32_t *d; hipMalloc(&d, 256);
    int blocks = cus * 8, threads = 256;    // 8 waves per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("%-30s %12s %14s %18s\n", "op", "ms", "Gop/s (lane)", "cyc/wave-op/SIMD");
    for (auto &e : es) {
        hipLaunchKernelGGL(e.k, dim3(blocks), dim3(threads), 0, 0, d, 12345u);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 5; r++) hipLaunchKernelGGL(e.k, dim3(blocks), dim3(threads), 0, 0, d, 12345u + r);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        double ops = (double)blocks * threads * ITERS * CHAINS;
        double waveops = ops / 64.0;
        double cyc = (ms * 1e-3 * clk) * (cus * 4.0) / waveops;   // SIMD-cycles per wave-level op
        printf("%-30s %12.4f %14.1f %18.2f\n", e.name, ms, ops / ms / 1e6, cyc);
    }
    return 0;
}
