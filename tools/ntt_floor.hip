// tools/ntt_floor.hip -- the MEASURED issue-bound floor of one NTT digit pass.
// The four W = 10 passes of `from_values` at 2^20 rows (inverse DIF: TILE then ROW; forward coset DIT: ROW then TILE with the permuted
// store) are instantiated from the product kernel's own source (csrc/ntt_fast_kernel.cuh) three times:
//   mode 0  as shipped: real loads, tables and stores                                   (must reproduce bench.py's pass times)
//   mode 1  no global data traffic: loads are values made from their addresses, stores never happen   (NTT_FLOOR=1)
//   mode 2  additionally no LDS exchange between the register phases                      (NTT_FLOOR=2)
// Mode 2 is the pure instruction stream -- butterflies, shift twiddles, inter-phase and factor products, canonical checks -- at the
// pass's own occupancy (two 512-thread workgroups per CU): its run time is the floor no memory-side change can go below; mode 1 adds
// what the LDS exchanges and their barriers cost; mode 0 - mode 1 is what the pass waits for HBM.
// Build (three binaries, one per mode):  hipcc -O3 -std=c++17 --offload-arch=gfx950 -DNTT_FLOOR=<m> -Ieth-tx-proof_amd/csrc tools/ntt_floor.hip -o tools/bin/ntt_floor<m>
// Run:  tools/bin/ntt_floor<m> [n_cols=93] [reps=20] [col_group=0] [persist=0] [stride_pad=0] [nt_store=1]      -> one line per pass: us per launch, ns per element, SIMD-cycles per element
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "ntt_fast_kernel.cuh"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void fill_random(u64 *p, size_t n, u64 seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        u64 z = (i + 1) * 0x9E3779B97F4A7C15ull + seed * 0xD1B54A32D192ED03ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
        p[i] = z >= GL_P ? z - GL_P : z;
    }
}
static int g_persist = 0;
template <bool DIT, bool ROW, bool INV>
static float run(const PassParams &p, int reps, hipStream_t st) {
    constexpr int W = 10;
    const size_t n = (size_t)1 << p.k;
    const size_t blocks = ROW ? ((n >> W) * (size_t)p.n_cols) >> p.TL : ((n >> W) >> p.TL) * (size_t)p.n_cols;
    const size_t shmem = (NttGeom<ROW, W>::tile_elems + ((size_t)1 << W)) * sizeof(u64);
    PassParams pp = p; size_t grid = blocks;
    if (g_persist > 0 && blocks > (size_t)g_persist) { pp.n_blocks = (unsigned)blocks; grid = (size_t)g_persist; }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL((ntt_fast_kernel<DIT, ROW, W, INV, false>), dim3((unsigned)grid), dim3(512), shmem, st, pp);
    hipEventRecord(e0, st);
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL((ntt_fast_kernel<DIT, ROW, W, INV, false>), dim3((unsigned)grid), dim3(512), shmem, st, pp);
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return ms / reps;
}

int main(int argc, char **argv) {
    const int n_cols = argc > 1 ? atoi(argv[1]) : 93, reps = argc > 2 ? atoi(argv[2]) : 20, col_group = argc > 3 ? atoi(argv[3]) : 0;
    g_persist = argc > 4 ? atoi(argv[4]) : 0;
    const unsigned nt_store = argc > 6 ? (unsigned)atoi(argv[6]) : 1;      // whole-line stores with the non-temporal hint (the library sets it for big passes)
    const size_t pad = argc > 5 ? (size_t)atol(argv[5]) : 0;       // column stride = n + pad elements (are power-of-two strides a channel hot spot?)
    const int k = 20;
    const size_t n = (size_t)1 << k;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const double simds = 4.0 * prop.multiProcessorCount, ghz = prop.clockRate * 1e-6;
    u64 *in, *mid, *out, *tab, *rw;
    const size_t cs = n + pad;
    CK(hipMalloc(&in, cs * n_cols * 8)); CK(hipMalloc(&mid, cs * n_cols * 8)); CK(hipMalloc(&out, 2 * cs * n_cols * 8));
    CK(hipMalloc(&tab, n * 8)); CK(hipMalloc(&rw, 1024 * 8));
    // random canonical field elements everywhere: on zeros (or on values made from addresses) the weak arithmetic meets its
    // rare-repair paths -- and the exact redo -- far more often than on real data
    fill_random<<<4096, 256>>>(in, cs * n_cols, 1); fill_random<<<4096, 256>>>(mid, cs * n_cols, 2);
    fill_random<<<1024, 256>>>(tab, n, 3); fill_random<<<4, 256>>>(rw, 1024, 4);
    CK(hipDeviceSynchronize());
    hipStream_t st; CK(hipStreamCreate(&st));
    PassParams b = {};
    b.k = k; b.W = 10; b.TL = 3; b.n_cols = n_cols; b.tw_full = rw; b.col_group = col_group; b.load_tab_mask = b.store_tab_mask = ~0u; b.nt_store = nt_store;
    struct { const char *name; float ms; } res[4];
    { PassParams p = b; p.in = in; p.in_stride = cs; p.out = mid; p.out_stride = cs; p.S = 10; p.inverse = 1; p.canon_in = 1; p.store_tab = tab; p.col_fast = 2;
      res[0] = {"inverse DIF pass 1 (TILE, inter-pass factors at the store)", run<false, false, true>(p, reps, st)}; }
    { PassParams p = b; p.in = mid; p.in_stride = cs; p.out = mid; p.out_stride = cs; p.S = 0; p.inverse = 1; p.store_tab = tab; p.col_fast = 1;
      res[1] = {"inverse DIF pass 2 (ROW, 1/n at the store)", run<false, true, true>(p, reps, st)}; }
    { PassParams p = b; p.in = mid; p.in_stride = cs; p.out = in; p.out_stride = cs; p.S = 0; p.load_tab = tab; p.col_fast = 1;
      res[2] = {"forward DIT pass 1 (ROW, coset factors at the load)", run<true, true, false>(p, reps, st)}; }
    { PassParams p = b; p.in = in; p.in_stride = cs; p.out = out; p.out_stride = 2 * cs; p.S = 10; p.load_tab = tab; p.permute_out = 1; p.col_fast = 2;
      res[3] = {"forward DIT pass 2 (TILE, inter-pass factors at the load, permuted store)", run<true, false, false>(p, reps, st)}; }
    CK(hipStreamSynchronize(st));
    printf("# ntt_floor mode %d: %d columns x 2^%d, W = 10, col_group %d, persist %d, stride pad %zu, nt stores %u, %d SIMDs at %.2f GHz, %d launches each\n", (int)NTT_FLOOR, n_cols, k, col_group, g_persist, pad, nt_store, (int)simds, ghz, reps);
    const double elems = (double)n * n_cols;
    for (auto &r : res)
        printf("%-78s %8.1f us  %6.3f ns/element  %6.2f SIMD-cycles/element  %7.1f GB/s algorithmic\n", r.name, r.ms * 1e3, r.ms * 1e6 / elems,
               r.ms * 1e-3 * ghz * 1e9 * simds / elems, 16.0 * elems / (r.ms * 1e-3) * 1e-9);
    return 0;
}
