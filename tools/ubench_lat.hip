// Single-wave latency micro-benchmark for gfx950: what ONE wave pays per instruction in a dependent chain, with 1 / 2 / 4 independent
// chains interleaved.  A serial sponge (the Fiat-Shamir transcript) and the small Merkle levels run one permutation per wave, so their
// time is this, not the chip's issue rate (tools/ubench_int.hip).
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench_lat.hip -o /tmp/ubench_lat ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define ITERS 512

#define DEF_LAT(NAME, ...)                                                          \
    template <int CH> __global__ void NAME(uint64_t *out, uint32_t seed) {                \
        uint64_t a[4]; uint32_t b = seed | 1, c = seed ^ 0x9E3779B9u; uint64_t b64 = ((uint64_t)b << 32) | c; (void)b64; (void)c; \
        uint32_t m[4]; uint64_t k[4]; uint32_t s[4] = {seed, seed + 1, seed + 2, seed + 3}; (void)m; (void)k; (void)s; \
        for (int i = 0; i < 4; i++) a[i] = (uint64_t)(threadIdx.x * 2654435761u + i + seed) * 0x9E3779B97F4A7C15ULL; \
        uint64_t t0 = clock64();                                                           \
        for (int it = 0; it < ITERS; it++) {                                               \
            _Pragma("unroll") for (int u = 0; u < 8; u++) {                                \
                _Pragma("unroll") for (int i = 0; i < CH; i++) { __VA_ARGS__; }                    \
            }                                                                              \
        }                                                                                  \
        uint64_t t1 = clock64();                                                           \
        uint64_t r = 0;                                                                    \
        for (int i = 0; i < 4; i++) r ^= a[i];                                             \
        if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = r; }                            \
    }

DEF_LAT(l_add_u32, asm volatile("v_add_u32 %0, %1, %0" : "+v"(*(uint32_t *)&a[i]) : "v"(b)))
DEF_LAT(l_mad_acc, asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c) : "vcc"))
DEF_LAT(l_mad_mul, { uint32_t lo = (uint32_t)a[i]; asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(a[i]) : "v"(lo), "v"(c), "v"(b64) : "vcc"); })
DEF_LAT(l_lshl_add_u64, asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(a[i]) : "v"(b64)))
DEF_LAT(l_addc_pair, { uint32_t lo = (uint32_t)a[i], hi = (uint32_t)(a[i] >> 32); asm volatile("v_add_co_u32 %0, vcc, %2, %0\n\tv_addc_co_u32 %1, vcc, %3, %1, vcc" : "+v"(lo), "+v"(hi) : "v"(b), "v"(c) : "vcc"); a[i] = ((uint64_t)hi << 32) | lo; })
DEF_LAT(l_mad_eps, { uint32_t h = (uint32_t)(a[i] >> 32); asm volatile("v_mad_u64_u32 %[r], %[k], %[h], -1, %[r]\n\ts_nop 1\n\tv_cndmask_b32_e64 %[m], 0, -1, %[k]\n\tv_mad_u64_u32 %[r], %[k], %[m], 1, %[r]" : [r] "+v"(a[i]), [k] "=&s"(k[i]), [m] "=&v"(m[i]) : [h] "v"(h)); })
DEF_LAT(l_readlane_mad, { uint32_t lo = (uint32_t)a[i]; uint32_t s; asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(s) : "v"(lo)); asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[i]) : "s"(s), "v"(c) : "vcc"); })
DEF_LAT(l_snop0, asm volatile("s_nop 0" : "+v"(a[i])))
DEF_LAT(l_snop1, asm volatile("s_nop 1" : "+v"(a[i])))
DEF_LAT(l_mad_snop1, asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\ts_nop 1" : "+v"(a[i]) : "v"(b), "v"(c) : "vcc"))
DEF_LAT(l_cndmask_sgpr, asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[10:11]" : "+v"(*(uint32_t *)&a[i]) : "v"(b) : "s10", "s11"))
DEF_LAT(l_smul, { asm volatile("s_mul_i32 %0, %0, %1" : "+s"(s[i]) : "s"(b)); a[i] ^= 0; if (it == ITERS - 1 && u == 7) a[i] += s[i]; })
DEF_LAT(l_saddc, { asm volatile("s_add_u32 %0, %0, %1\n\ts_addc_u32 %0, %0, %1" : "+s"(s[i]) : "s"(b) : "scc"); if (it == ITERS - 1 && u == 7) a[i] += s[i]; })
DEF_LAT(l_mov_dpp, asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(*(uint32_t *)&a[i])))
DEF_LAT(l_permlane32, { uint32_t lo = (uint32_t)a[i], hi = (uint32_t)(a[i] >> 32); asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(lo), "+v"(hi)); a[i] = ((uint64_t)hi << 32) | lo; })

typedef void (*kern_t)(uint64_t *, uint32_t);
struct Entry { const char *name; kern_t k1, k2, k4; int insts; };
#define E(n, f, c) {n, f<1>, f<2>, f<4>, c}

int main() {
    Entry es[] = {
        E("v_add_u32", l_add_u32, 1), E("v_mad_u64_u32 (dep: accumulator)", l_mad_acc, 1), E("v_mad_u64_u32 (dep: multiplicand)", l_mad_mul, 1),
        E("v_lshl_add_u64", l_lshl_add_u64, 1), E("v_add_co + v_addc", l_addc_pair, 2), E("mad, s_nop 1, cndmask(sgpr), mad", l_mad_eps, 4),
        E("v_readlane -> v_mad (sgpr operand)", l_readlane_mad, 2), E("s_nop 0", l_snop0, 1), E("s_nop 1", l_snop1, 1), E("v_mad + s_nop 1", l_mad_snop1, 2),
        E("v_cndmask_b32_e64 (sgpr mask)", l_cndmask_sgpr, 1), E("s_mul_i32 (dep)", l_smul, 1), E("s_add_u32 + s_addc_u32 (dep)", l_saddc, 2),
        E("s_nop 1 + v_mov_b32_dpp", l_mov_dpp, 2), E("s_nop 1 + v_permlane32_swap", l_permlane32, 2),
    };
    uint64_t *d; hipMalloc(&d, 64);
    printf("one wave alone on the chip; cycles (s_memtime, shader clock) per GROUP of instructions, by number of independent chains interleaved\n");
    printf("%-40s %6s %10s %10s %10s\n", "group", "instrs", "1 chain", "2 chains", "4 chains");
    for (auto &e : es) {
        double cyc[3]; kern_t ks[3] = {e.k1, e.k2, e.k4}; int ch[3] = {1, 2, 4};
        for (int j = 0; j < 3; j++) {
            uint64_t h[2], best = ~0ull;
            for (int rep = 0; rep < 5; rep++) {
                hipLaunchKernelGGL(ks[j], dim3(1), dim3(64), 0, 0, d, 12345u + rep);
                hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
                if (h[0] < best) best = h[0];
            }
            cyc[j] = (double)best / (ITERS * 8.0 * ch[j]);
        }
        printf("%-40s %6d %10.2f %10.2f %10.2f\n", e.name, e.insts, cyc[0], cyc[1], cyc[2]);
    }
    return 0;
}
