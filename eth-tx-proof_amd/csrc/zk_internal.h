// Internal host-side declarations shared by the .hip translation units of libzkgl_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>
#include "../../include/zkgl.h"
#include "gl64.cuh"

struct ProfRec {
    std::string name;
    uint64_t launches = 0;
    double total_ms = 0.0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};

struct TwoLevel { u64 *A = nullptr, *B = nullptr; };   // x -> A[x & 2047] * B[x >> 11]

struct zk_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    // test hook (env ZK_FORCE_EXACT_REDO=1): the hashing kernels take their exact-redo branch on every wave (poseidon.cuh)
    u64 force_redo = 0;
    // constraint-program JIT (prover_quotient_jit.hip): env ZK_QUOTIENT_JIT = -1 never (default) / 0 only kernels compiled ahead
    // of time by zk_prover_state_prepare / 1 compile at first use for tables with >= 2^ZK_QUOTIENT_JIT_MIN_LOG coset points
    // (default 17) / 2 always; read when the context is created.  Off by default: in bench.py (first proof of a process in a
    // worker thread) the compiled kernel ended in a GPU memory fault in about one process out of three -- also when compiled
    // ahead of time from the main thread, also with the previous round's field arithmetic; never with the interpreter (20 of 20
    // processes), never in the test-suite or single-threaded drivers.  Cause not found: DESIGN.md "known issues".
    int quotient_jit_mode = -1, quotient_jit_min_log = 17;
    // matrix-core Poseidon for big leaf / Merkle-level launches (poseidon_mfma.cuh): env ZK_POSEIDON_MFMA: bit 0 leaf
    // hashing, bit 1 Merkle levels (default 3, 0 = vector-ALU kernels only)
    int poseidon_mfma = 3;
    int sync_launch = 0;
    int efence = 0;                                                // debug allocator, see zk_ctx::alloc
    void *d_psdm = nullptr;                                        // this device's PsdmTables (shared, never freed; poseidon_kernels.hip)
    std::string last_error;
    // caching device allocator (stream-ordered reuse on the ctx stream)
    std::multimap<size_t, void *> free_blocks;
    std::unordered_map<void *, size_t> block_size;
    size_t bytes_allocated = 0;
    // per-kernel hipEvent profiler
    bool prof = false;
    std::vector<ProfRec> prof_recs;
    std::unordered_map<std::string, int> prof_index;
    std::vector<hipEvent_t> event_pool;
    // twiddle / scale tables (device)
    std::map<std::pair<int, int>, u64 *> digit_tw;                 // (W, inverse)
    std::map<std::pair<int, int>, TwoLevel> root_tab;              // (m, inverse): w_{2^m}^{+-x}
    std::map<std::tuple<u64, u64, int>, TwoLevel> scale_tab;       // (base, premul, log_count): premul*base^j
    std::map<std::pair<int, int>, u64 *> lagrange_tab;             // (log_n, rate_bits): L_first|L_last on the coset, R order
    std::map<std::string, u64 *> table_cache;                      // fused per-position NTT pass tables (ntt_fast.hip)
    u64 *d_err = nullptr;                                          // device error word (first error wins)

    // Pinned host memory for the asynchronous copies of a proof.  hipMemcpyAsync on pageable memory makes the runtime register
    // the caller's pages with the GPU on the fly and drop them again when the copy retires; with several small host buffers
    // of several threads sharing heap pages, that ended in sporadic "memory access fault by GPU" at HOST heap addresses
    // (bench.py, four streams).  Small uploads are therefore copied into h_stage first (bump allocator, reset per proof) and
    // the proof words come back through h_out.
    char *h_stage = nullptr; size_t h_stage_cap = 0, h_stage_used = 0;
    char *h_out = nullptr; size_t h_out_cap = 0;
    const void *stage_pinned(const void *src, size_t bytes);      // nullptr when it does not fit (caller copies synchronously)
    void *pinned_out(size_t bytes);                                // download buffer of at least `bytes`
    int32_t upload_small(void *d_dst, const void *src, size_t bytes);   // stream-ordered H2D of a small host object

    void *alloc(size_t bytes);
    void release(void *p);
    int32_t fail(int32_t code, const std::string &msg) { last_error = msg; return code; }
    void prof_flush();
};

struct ProfScope {
    zk_ctx *ctx; int idx; hipEvent_t e0, e1;
    ProfScope(zk_ctx *c, const char *name);
    ~ProfScope();
};

// debug hook (env ZK_SYNC_LAUNCH=1): every launch is named on stderr and waited for, so a faulting kernel is the last name printed
#define ZK_LAUNCH(ctx, name, kern, grid, block, shmem, ...)                                   \
    do {                                                                                      \
        ProfScope _ps((ctx), (name));                                                         \
        if ((ctx)->sync_launch) fprintf(stderr, "[zk %p] %s grid %u block %u\n", (void *)(ctx), (name), (unsigned)(grid).x, (unsigned)(block).x); \
        hipLaunchKernelGGL(kern, grid, block, shmem, (ctx)->stream, __VA_ARGS__);             \
        if ((ctx)->sync_launch) hipStreamSynchronize((ctx)->stream);                          \
    } while (0)

#define HIP_TRY(ctx, expr)                                                                    \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) return (ctx)->fail(ZK_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

// ---- tables (ntt.hip)
u64 *zk_digit_twiddles(zk_ctx *ctx, int W, int inverse);
TwoLevel zk_root_table(zk_ctx *ctx, int m, int inverse);
TwoLevel zk_scale_table(zk_ctx *ctx, u64 base, u64 premul, int log_count);

// ---- NTT engines (ntt.hip); all device pointers, column-major with explicit column strides (in elements)
// natural order in -> bit-reversed order out.  pre (optional): multiply input j by pre.A/B(j).  post_const: multiply outputs.
int32_t zk_ntt_dif(zk_ctx *ctx, const u64 *in, size_t in_stride, u64 *out, size_t out_stride, int k, int n_cols,
                   int inverse, const TwoLevel *pre, u64 post_const, int canon_in);
// bit-reversed order in -> natural order out (permute_out=0) or bit-reversed out (permute_out=1; needs tmp of n_cols*2^k
// when k needs more than one pass).  pre (optional) is indexed by the coefficient index j = bitrev(pos); post by output index.
int32_t zk_ntt_dit(zk_ctx *ctx, const u64 *in, size_t in_stride, u64 *out, size_t out_stride, int k, int n_cols,
                   int inverse, const TwoLevel *pre, const TwoLevel *post, int permute_out, u64 *tmp, size_t tmp_stride);
int32_t zk_bitrev_copy(zk_ctx *ctx, const u64 *in, size_t in_stride, u64 *out, size_t out_stride, int k, int n_cols);

// ---- Poseidon / Merkle (poseidon_kernels.hip)
// digests of n_leaves leaves held column-major: leaf i = (cols[c*stride + i])_c
int32_t zk_hash_leaves_colmajor(zk_ctx *ctx, const u64 *cols, size_t stride, uint32_t n_cols, size_t n_leaves, u64 *digests);
// digests of n_leaves leaves held row-major (leaf_len contiguous elements each)
int32_t zk_hash_leaves_rowmajor(zk_ctx *ctx, const u64 *rows, uint32_t leaf_len, size_t n_leaves, u64 *digests);
// levels: level 0 = n_leaves digests at `levels`, level k follows; builds up to the cap level
int32_t zk_merkle_levels(zk_ctx *ctx, u64 *levels, size_t n_leaves, uint32_t cap_height);
// levels (level order) -> upstream's recursive `digests` layout, device to device
int32_t zk_digests_upstream_layout(zk_ctx *ctx, const u64 *levels, size_t n_leaves, uint32_t n_levels, u64 *d_digests);
// evaluate n_cols bit-reversed-coefficient polynomials (2^k coeffs each) at the extension point stored at d_point[0..1]
// (device memory); writes n_cols ext values to d_out.  scratch is managed internally.
int32_t zk_eval_cols_at(zk_ctx *ctx, const u64 *coeffs, size_t stride, int k, int n_cols, const u64 *d_point, u64 *d_out);
static inline size_t zk_level_offset(size_t n_leaves, uint32_t k) {   // in digests
    size_t off = 0;
    for (uint32_t j = 0; j < k; j++) off += n_leaves >> j;
    return off;
}

struct zk_batch {
    zk_ctx *ctx;
    uint32_t log_n, n_cols, rate_bits, cap_height;
    u64 *coeffs;    // n_cols x n, column-major, BIT-REVERSED coefficient order: coeffs[c*n + bitrev(j)] = c_j
    u64 *lde;       // n_cols x (n<<rate_bits), column-major, bit-reversed row order (= leaf order)
    u64 *levels;    // Merkle levels 0..(log L - cap_height), 4 words per digest
    size_t n() const { return (size_t)1 << log_n; }
    size_t L() const { return (size_t)1 << (log_n + rate_bits); }
    uint32_t n_levels() const { return log_n + rate_bits - cap_height + 1; }
    const u64 *cap() const { return levels + 4 * zk_level_offset(L(), n_levels() - 1); }
};
// values_natural == 1: cols are evaluations on the subgroup (from_values); 0: natural-order coefficients (from_coeffs)
int32_t zk_batch_commit_dev(zk_ctx *ctx, const u64 *d_cols, uint32_t log_n, uint32_t n_cols, uint32_t rate_bits,
                            uint32_t cap_height, int values_natural, zk_batch **out);
