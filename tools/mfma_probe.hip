// Probe of v_mfma_i32_32x32x32_i8 on gfx950: operand lane maps, signedness, C/D map -- checked with exact integer data.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__global__ void k_probe(const int8_t *A /*[64][16]*/, const int8_t *B /*[64][16]*/, int *D /*[64][16]*/) {
    int l = threadIdx.x;
    v4i a = *(const v4i *)(A + 16 * l), b = *(const v4i *)(B + 16 * l);
    v16i c; for (int i = 0; i < 16; i++) c[i] = 0;
    c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0);
    for (int i = 0; i < 16; i++) D[16 * l + i] = c[i];
}
__global__ void k_swap(unsigned *out) {
    unsigned l = threadIdx.x;
    unsigned x = 1000 + l, y = 2000 + l;
    auto r = __builtin_amdgcn_permlane32_swap(y, x, false, false);     // (vdst_old = y, src0_old = x)
    out[2 * l] = r[0]; out[2 * l + 1] = r[1];
}
int main() {
    {
        unsigned *d, h[128]; hipMalloc(&d, sizeof h);
        hipLaunchKernelGGL(k_swap, dim3(1), dim3(64), 0, 0, d);
        hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        printf("permlane32_swap(vdst=y=2000+l, src=x=1000+l): lane0 -> (%u, %u)  lane31 -> (%u, %u)  lane32 -> (%u, %u)  lane63 -> (%u, %u)\n",
               h[0], h[1], h[62], h[63], h[64], h[65], h[126], h[127]);
    }
    int8_t hA[64 * 16], hB[64 * 16]; int hD[64 * 16];
    srand(1);
    for (int i = 0; i < 64 * 16; i++) { hA[i] = (int8_t)(rand() & 255); hB[i] = (int8_t)(rand() & 255); }
    int8_t *dA, *dB; int *dD;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
    // hypothesis: lane l, byte j <-> k = 16*(l>>5) + j for both operands (any common k order gives the same sums),
    // A row = l&31, B col = l&31; D[reg] at lane l: row = (reg&3) + 8*(reg>>2) + 4*(l>>5), col = l&31
    int bad = 0;
    for (int l = 0; l < 64; l++) for (int r = 0; r < 16; r++) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
        long s = 0;
        for (int h = 0; h < 2; h++) for (int j = 0; j < 16; j++) s += (long)hA[16 * (row + 32 * h) + j] * hB[16 * (col + 32 * h) + j];
        if (s != hD[16 * l + r]) { if (bad < 5) printf("mismatch lane %d reg %d: got %d want %ld\n", l, r, hD[16 * l + r], s); bad++; }
    }
    printf("mfma_i32_32x32x32_i8 signed x signed, pairing (lane half, byte j): %s (%d mismatches)\n", bad ? "FAIL" : "OK", bad);
    return bad != 0;
}
