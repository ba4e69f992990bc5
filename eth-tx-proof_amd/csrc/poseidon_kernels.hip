// PoseidonGoldilocks hashing kernels: leaf sponges over column-major LDE values, Merkle interior levels, caps.
// Mirrors plonky2 0.2.2 hash::{hashing::hash_n_to_m_no_pad, hashing::hash_or_noop, poseidon::PoseidonHash::two_to_one,
// merkle_tree::MerkleTree::new} (crate pinned at /root/reference/Cargo.lock:3441-3443), reached from
// the reference at /root/reference/ops/src/lib.rs:52 through PolynomialBatch::from_values.
//
// Layout: one thread = one leaf.  Leaves are rows of a column-major matrix, so lane i reads element i of each
// column: every global load is a fully coalesced 512-byte wave access.  The 12-word sponge state stays in VGPRs.
//
// Big launches (whole commits, the wide Merkle levels) run the matrix-core permutation of poseidon_mfma.cuh: one wave =
// 64 leaves / parents in "pair layout" (lane (n, h) holds words 6h..6h+5 of leaves n and 32+n of the wave's 64), column
// loads still walk consecutive leaves (256-byte runs per column and half-wave).  Everything else keeps the one-thread-per-
// permutation kernels below.
#include <mutex>
#include "zk_internal.h"
#include "poseidon_coop.cuh"
#include "poseidon_wave.cuh"
#include "poseidon_mfma.cuh"

// sponge over the columns of one leaf; returns the sticky word of psd::core<false> (0 for EXACT)
template <bool EXACT> __device__ __forceinline__ u64 leaf_sponge_colmajor(const u64 *__restrict__ p, size_t stride, u32 n_cols, u64 s[12]) {
    psd::core<EXACT> pc; pc.fl = 0;
#pragma unroll
    for (int k = 0; k < 12; k++) s[k] = 0;
    u32 c = 0;
    for (; c + 8 <= n_cols; c += 8) {
#pragma unroll
        for (int k = 0; k < 8; k++) s[k] = p[(size_t)(c + k) * stride];
        pc.run(s);
    }
    if (c < n_cols) {
#pragma unroll
        for (int k = 0; k < 8; k++) if (c + k < n_cols) s[k] = p[(size_t)(c + k) * stride];
        pc.run(s);
    }
    return pc.fl;
}
// leaf i = (cols[c*stride + i])_{c < n_cols};  digest -> out[4*i .. 4*i+3]
__global__ __launch_bounds__(256) void hash_leaves_colmajor_kernel(const u64 *__restrict__ cols, size_t stride, u32 n_cols,
                                                                   size_t n_leaves, u64 *__restrict__ out, u64 force_redo) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_leaves) return;
    u64 s[12];
    if (n_cols <= 4) {            // hash_or_noop: short leaves are copied, zero padded
#pragma unroll
        for (int k = 0; k < 12; k++) s[k] = 0;
        for (u32 c = 0; c < n_cols; c++) s[c] = cols[(size_t)c * stride + i];
    } else if (leaf_sponge_colmajor<false>(cols + i, stride, n_cols, s) | force_redo) {
        leaf_sponge_colmajor<true>(cols + i, stride, n_cols, s);      // a skipped rare repair somewhere in this wave: redo exactly
    }
    ulonglong2 *o = (ulonglong2 *)(out + 4 * i);
    o[0] = make_ulonglong2(s[0], s[1]);
    o[1] = make_ulonglong2(s[2], s[3]);
}

// ---- matrix-core variants ------------------------------------------------------------------------------------------
// sponge over the columns of the wave's 64 leaves (leaf0 = first of them); state left in pair layout
template <bool EXACT> __device__ __forceinline__ u64 leaf_sponge_pair(const PsdmTables *__restrict__ T, const u64 *__restrict__ cols, size_t stride,
                                                                      u32 n_cols, size_t leaf0, u32 lane, const psdm_v4i *lds, u64 (&s)[2][6]) {
    psdm::perm<EXACT> pm; pm.init(T, lane, lds);
    const u32 n = lane & 31, h = lane >> 5;
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
        for (int i = 0; i < 6; i++) s[c][i] = 0;
    for (u32 c0 = 0; c0 < n_cols; c0 += 8) {
#pragma unroll
        for (int i = 0; i < 6; i++) {
            const u32 w = 6 * h + i;                       // sponge word held in register i of this lane half
            if (w < 8 && c0 + w < n_cols) {
                const u64 *p = cols + (size_t)(c0 + w) * stride + leaf0 + n;
                s[0][i] = p[0]; s[1][i] = p[32];
            }
        }
        pm.run(s);
    }
    return pm.pc.fl;
}
// 32 leaves per wave (psdm::perm<.., 1>): for commits of 16384 .. 65535 leaves, where the pair layout would leave SIMDs without a wave
template <bool EXACT> __device__ __forceinline__ u64 leaf_sponge_single(const PsdmTables *__restrict__ T, const u64 *__restrict__ cols, size_t stride,
                                                                        u32 n_cols, size_t leaf0, u32 lane, const psdm_v4i *lds, u64 (&s)[1][6]) {
    psdm::perm<EXACT, 1> pm; pm.init(T, lane, lds);
    const u32 n = lane & 31, h = lane >> 5;
#pragma unroll
    for (int i = 0; i < 6; i++) s[0][i] = 0;
    for (u32 c0 = 0; c0 < n_cols; c0 += 8) {
#pragma unroll
        for (int i = 0; i < 6; i++) {
            const u32 w = 6 * h + i;
            if (w < 8 && c0 + w < n_cols) s[0][i] = cols[(size_t)(c0 + w) * stride + leaf0 + n];
        }
        pm.run(s);
    }
    return pm.pc.fl;
}
__global__ __launch_bounds__(256, 3) void hash_leaves_colmajor_mfma32_kernel(const PsdmTables *__restrict__ T, const u64 *__restrict__ cols, size_t stride,
                                                                            u32 n_cols, size_t n_leaves, u64 *__restrict__ out, u64 force_redo) {
    const u32 lane = threadIdx.x & 63;
    const size_t leaf0 = ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 32;                 // n_leaves is a multiple of 128
    __shared__ psdm_v4i shA[PSDM_LDS_V4];
    psdm::lds_fill(T, shA); __syncthreads();
    u64 s[1][6];
    if (leaf_sponge_single<false>(T, cols, stride, n_cols, leaf0, lane, shA, s) | force_redo)
        leaf_sponge_single<true>(T, cols, stride, n_cols, leaf0, lane, shA, s);
    if (lane < 32) {
        ulonglong2 *o = (ulonglong2 *)(out + 4 * (leaf0 + lane));
        o[0] = make_ulonglong2(s[0][0], s[0][1]);
        o[1] = make_ulonglong2(s[0][2], s[0][3]);
    }
}
__global__ __launch_bounds__(256, 3) void hash_leaves_colmajor_mfma_kernel(const PsdmTables *__restrict__ T, const u64 *__restrict__ cols, size_t stride,
                                                                          u32 n_cols, size_t n_leaves, u64 *__restrict__ out, u64 force_redo) {
    const u32 lane = threadIdx.x & 63;
    const size_t leaf0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) & ~(size_t)63;     // n_leaves is a multiple of 256
    __shared__ psdm_v4i shA[PSDM_LDS_V4];
    psdm::lds_fill(T, shA); __syncthreads();
    u64 s[2][6];
    if (leaf_sponge_pair<false>(T, cols, stride, n_cols, leaf0, lane, shA, s) | force_redo)
        leaf_sponge_pair<true>(T, cols, stride, n_cols, leaf0, lane, shA, s);      // a skipped rare repair somewhere in this wave: redo exactly
    if (lane < 32) {
#pragma unroll
        for (int c = 0; c < 2; c++) {
            ulonglong2 *o = (ulonglong2 *)(out + 4 * (leaf0 + 32 * c + lane));
            o[0] = make_ulonglong2(s[c][0], s[c][1]);
            o[1] = make_ulonglong2(s[c][2], s[c][3]);
        }
    }
}
template <bool EXACT> __device__ __forceinline__ u64 two_to_one_pair(const PsdmTables *__restrict__ T, const u64 *__restrict__ child, size_t parent0,
                                                                     u32 lane, const psdm_v4i *lds, u64 (&s)[2][6]) {
    psdm::perm<EXACT> pm; pm.init(T, lane, lds);
    const u32 n = lane & 31, h = lane >> 5;
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
    return pm.pc.fl;
}
// 32 parents per wave, for levels of 16384 .. 65535 parents (see hash_leaves_colmajor_mfma32_kernel)
template <bool EXACT> __device__ __forceinline__ u64 two_to_one_single(const PsdmTables *__restrict__ T, const u64 *__restrict__ child, size_t parent0,
                                                                       u32 lane, const psdm_v4i *lds, u64 (&s)[1][6]) {
    psdm::perm<EXACT, 1> pm; pm.init(T, lane, lds);
    const u32 n = lane & 31, h = lane >> 5;
    const u64 *p = child + 8 * (parent0 + n);
    if (h == 0) {
        const ulonglong2 a = ((const ulonglong2 *)p)[0], b = ((const ulonglong2 *)p)[1], d = ((const ulonglong2 *)p)[2];
        s[0][0] = a.x; s[0][1] = a.y; s[0][2] = b.x; s[0][3] = b.y; s[0][4] = d.x; s[0][5] = d.y;
    } else {
        const ulonglong2 e = ((const ulonglong2 *)p)[3];
        s[0][0] = e.x; s[0][1] = e.y; s[0][2] = 0; s[0][3] = 0; s[0][4] = 0; s[0][5] = 0;
    }
    pm.run(s);
    return pm.pc.fl;
}
__global__ __launch_bounds__(256, 3) void merkle_level_mfma32_kernel(const PsdmTables *__restrict__ T, const u64 *__restrict__ child, u64 *__restrict__ parent,
                                                                    size_t n_parents, u64 force_redo) {
    const u32 lane = threadIdx.x & 63;
    const size_t parent0 = ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 32;               // n_parents is a multiple of 128
    __shared__ psdm_v4i shA[PSDM_LDS_V4];
    psdm::lds_fill(T, shA); __syncthreads();
    u64 s[1][6];
    if (two_to_one_single<false>(T, child, parent0, lane, shA, s) | force_redo) two_to_one_single<true>(T, child, parent0, lane, shA, s);
    if (lane < 32) {
        ulonglong2 *o = (ulonglong2 *)(parent + 4 * (parent0 + lane));
        o[0] = make_ulonglong2(s[0][0], s[0][1]);
        o[1] = make_ulonglong2(s[0][2], s[0][3]);
    }
}
__global__ __launch_bounds__(256, 3) void merkle_level_mfma_kernel(const PsdmTables *__restrict__ T, const u64 *__restrict__ child, u64 *__restrict__ parent,
                                                                  size_t n_parents, u64 force_redo) {
    const u32 lane = threadIdx.x & 63;
    const size_t parent0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) & ~(size_t)63;   // n_parents is a multiple of 256
    __shared__ psdm_v4i shA[PSDM_LDS_V4];
    psdm::lds_fill(T, shA); __syncthreads();
    u64 s[2][6];
    if (two_to_one_pair<false>(T, child, parent0, lane, shA, s) | force_redo) two_to_one_pair<true>(T, child, parent0, lane, shA, s);
    if (lane < 32) {
#pragma unroll
        for (int c = 0; c < 2; c++) {
            ulonglong2 *o = (ulonglong2 *)(parent + 4 * (parent0 + 32 * c + lane));
            o[0] = make_ulonglong2(s[c][0], s[c][1]);
            o[1] = make_ulonglong2(s[c][2], s[c][3]);
        }
    }
}
// leaves of a FRI commit-phase layer: `arity` consecutive extension values flattened [a0,b0,a1,b1,...] from the re|im columns
// (plonky2 0.2.2 fri::prover::fri_committed_trees; prover_fri.hip holds the one-thread-per-leaf version for small layers)
template <bool EXACT> __device__ __forceinline__ u64 fri_leaf_sponge_pair(const PsdmTables *__restrict__ T, const u64 *__restrict__ re, const u64 *__restrict__ im,
                                                                          u32 arity, size_t leaf0, u32 lane, const psdm_v4i *lds, u64 (&s)[2][6]) {
    psdm::perm<EXACT> pm; pm.init(T, lane, lds);
    const u32 n = lane & 31, h = lane >> 5;
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
        for (int i = 0; i < 6; i++) s[c][i] = 0;
    for (u32 c0 = 0; c0 < arity; c0 += 4) {          // four extension values = eight sponge words per permutation
#pragma unroll
        for (int i = 0; i < 6; i++) {
            const u32 w = 6 * h + i;
            if (w < 8) {
                const u64 *col = (w & 1) ? im : re;
#pragma unroll
                for (int c = 0; c < 2; c++) s[c][i] = col[(leaf0 + 32 * c + n) * arity + c0 + (w >> 1)];
            }
        }
        pm.run(s);
    }
    return pm.pc.fl;
}
__global__ __launch_bounds__(256, 3) void fri_leaf_hash_mfma_kernel(const PsdmTables *__restrict__ T, const u64 *__restrict__ re, const u64 *__restrict__ im,
                                                                   size_t n_leaves, u32 arity, u64 *__restrict__ out, u64 force_redo) {
    const u32 lane = threadIdx.x & 63;
    const size_t leaf0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) & ~(size_t)63;     // n_leaves is a multiple of 256
    __shared__ psdm_v4i shA[PSDM_LDS_V4];
    psdm::lds_fill(T, shA); __syncthreads();
    u64 s[2][6];
    if (fri_leaf_sponge_pair<false>(T, re, im, arity, leaf0, lane, shA, s) | force_redo) fri_leaf_sponge_pair<true>(T, re, im, arity, leaf0, lane, shA, s);
    if (lane < 32) {
#pragma unroll
        for (int c = 0; c < 2; c++) {
            ulonglong2 *o = (ulonglong2 *)(out + 4 * (leaf0 + 32 * c + lane));
            o[0] = make_ulonglong2(s[c][0], s[c][1]);
            o[1] = make_ulonglong2(s[c][2], s[c][3]);
        }
    }
}
// Small commits (fewer leaves than would give every SIMD a matrix-core wave): the sponge of ONE leaf is a serial chain of
// ceil(C/8) permutations, and one thread per leaf walks it at ~60 us per permutation.  Here a leaf gets 16 lanes (poseidon_coop.cuh,
// ~20 us per permutation, up to 8192 leaves) or a whole wave (poseidon_wave.cuh, ~10 us, up to 1024 leaves): word l of the state in
// lane l, lanes 0..7 fetch the next eight columns.  430 columns x 2^10 leaves: 3.4 -> 0.6 ms.
template <bool WAVE> __global__ __launch_bounds__(256) void hash_leaves_colmajor_small_kernel(const u64 *__restrict__ cols, size_t stride, u32 n_cols,
                                                                                              size_t n_leaves, u64 *__restrict__ out, u64 force_redo) {
    __shared__ u64 wtab[WAVE ? psdw::TAB_WORDS : 1];
    if (WAVE) { psdw::fill<256>(wtab); __syncthreads(); }
    const int lane = threadIdx.x & 63, l = WAVE ? lane : (lane & 15), gb = lane & ~15;
    const size_t leaf = WAVE ? (size_t)blockIdx.x * 4 + (threadIdx.x >> 6) : ((size_t)blockIdx.x * 256 + threadIdx.x) >> 4;
    const bool live = leaf < n_leaves;
    u64 s = 0;
    for (u32 c0 = 0; c0 < n_cols; c0 += 8) {
        if (live && l < 8 && c0 + l < n_cols) s = cols[(size_t)(c0 + l) * stride + leaf];
        s = WAVE ? psdw::permute(s, lane, wtab, force_redo) : psdc::permute16(s, l, gb);
        if (l >= 12) s = 0;
    }
    if (live && l < 4) out[4 * leaf + l] = s;
}
// the same for the leaves of a FRI commit-phase layer (`arity` extension values flattened [a0,b0,a1,b1,...])
template <bool WAVE> __global__ __launch_bounds__(256) void fri_leaf_hash_small_kernel(const u64 *__restrict__ re, const u64 *__restrict__ im, size_t n_leaves,
                                                                                       u32 arity, u64 *__restrict__ out, u64 force_redo) {
    __shared__ u64 wtab[WAVE ? psdw::TAB_WORDS : 1];
    if (WAVE) { psdw::fill<256>(wtab); __syncthreads(); }
    const int lane = threadIdx.x & 63, l = WAVE ? lane : (lane & 15), gb = lane & ~15;
    const size_t leaf = WAVE ? (size_t)blockIdx.x * 4 + (threadIdx.x >> 6) : ((size_t)blockIdx.x * 256 + threadIdx.x) >> 4;
    const bool live = leaf < n_leaves;
    u64 s = 0;
    for (u32 w0 = 0; w0 < 2 * arity; w0 += 8) {
        const u32 w = w0 + (u32)l;
        if (live && l < 8 && w < 2 * arity) s = ((w & 1) ? im : re)[leaf * arity + (w >> 1)];
        s = WAVE ? psdw::permute(s, lane, wtab, force_redo) : psdc::permute16(s, l, gb);
        if (l >= 12) s = 0;
    }
    if (live && l < 4) out[4 * leaf + l] = s;
}
#define PSD_SMALL_WAVE_MAX 1024u      // leaves: one wave per leaf up to here (one wave per SIMD), 16 lanes per leaf up to PSD_SMALL_COOP_MAX
#define PSD_SMALL_COOP_MAX 8192u

// device copy of the matrix-core tables: built on the host once per process, one copy per device shared by all contexts,
// uploaded with a blocking hipMemcpy under a lock (complete before any launch that uses it can be enqueued)
static const PsdmTables *psdm_tables(zk_ctx *ctx) {
    if (ctx->d_psdm) return (const PsdmTables *)ctx->d_psdm;
    static PsdmTables host_tab;
    static std::mutex mu;
    static bool built = false;
    static void *dev_tab[64] = {nullptr};
    std::lock_guard<std::mutex> lk(mu);
    if (!built) { psdm_host::build(host_tab); built = true; }
    if (ctx->device < 0 || ctx->device >= 64) return nullptr;
    if (!dev_tab[ctx->device]) {
        void *d = nullptr;
        if (hipMalloc(&d, sizeof(PsdmTables)) != hipSuccess) return nullptr;
        if (hipMemcpy(d, &host_tab, sizeof(PsdmTables), hipMemcpyHostToDevice) != hipSuccess) { hipFree(d); return nullptr; }
        dev_tab[ctx->device] = d;
        ZK_ALLOC_TRACE("dev-poseidon-tables", d, sizeof(PsdmTables));
        if (ctx->sync_launch) fprintf(stderr, "[zk] PsdmTables at %p .. %p\n", d, (void *)((char *)d + sizeof(PsdmTables)));
    }
    ctx->d_psdm = dev_tab[ctx->device];
    return (const PsdmTables *)ctx->d_psdm;
}
// From how many units (leaves, parents) a launch goes to the matrix-core kernels (32 per wave below 65 536).  Rounds 2-3 drew the line at
// 16 384 ("below this a launch no longer fills the chip two waves per SIMD deep") and gave smaller launches to the lane-cooperative /
// wave-wide kernels, which answer sooner but cost 6-22 x the vector instructions per permutation (16 lanes: 1 525, a wave: 3 000, against
// 134-268).  Round 4 counted a transaction's instructions (tools/pmc_prove_root_valu.sh: those kernels were 20 % of them) and moved the
// line to 2 048: +5 % transactions/s with eight in flight, and a transaction ALONE is no slower (62.6 against 63.5 ms) -- between 2 048 and
// 16 384 units the cooperative kernels' extra instructions cost as much time as their parallelism saves.  ZK_PSDM_MIN_UNITS overrides.
static size_t psdm_min_units() {
    static const size_t v = getenv("ZK_PSDM_MIN_UNITS") ? (size_t)atol(getenv("ZK_PSDM_MIN_UNITS")) : 2048;
    return v < 256 ? 256 : v;
}
#define PSDM_MIN_UNITS psdm_min_units()

template <bool EXACT> __device__ __forceinline__ u64 leaf_sponge_rowmajor(const u64 *__restrict__ p, u32 leaf_len, int canon, u64 s[12]) {
    psd::core<EXACT> pc; pc.fl = 0;
#pragma unroll
    for (int k = 0; k < 12; k++) s[k] = 0;
    for (u32 c = 0; c < leaf_len; c += 8) {
#pragma unroll
        for (int k = 0; k < 8; k++) if (c + k < leaf_len) s[k] = canon ? gl_canon(p[c + k]) : p[c + k];
        pc.run(s);
    }
    return pc.fl;
}
// leaf i = rows[i*leaf_len .. (i+1)*leaf_len)
__global__ __launch_bounds__(256) void hash_leaves_rowmajor_kernel(const u64 *__restrict__ rows, u32 leaf_len, size_t n_leaves,
                                                                   u64 *__restrict__ out, int canon, u64 force_redo) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_leaves) return;
    const u64 *p = rows + i * (size_t)leaf_len;
    u64 s[12];
    if (leaf_len <= 4) {
#pragma unroll
        for (int k = 0; k < 12; k++) s[k] = 0;
        for (u32 c = 0; c < leaf_len; c++) s[c] = canon ? gl_canon(p[c]) : p[c];
    } else if (leaf_sponge_rowmajor<false>(p, leaf_len, canon, s) | force_redo) {
        leaf_sponge_rowmajor<true>(p, leaf_len, canon, s);
    }
#pragma unroll
    for (int k = 0; k < 4; k++) out[4 * i + k] = s[k];
}

// parent j = two_to_one(child 2j, child 2j+1) = permute(L | R | 0^4)[0..4]
__device__ __forceinline__ void load_children(const u64 *__restrict__ child, size_t j, u64 s[12]) {
    const ulonglong2 *c = (const ulonglong2 *)(child + 8 * j);
    ulonglong2 a = c[0], b = c[1], d = c[2], e = c[3];
    s[0] = a.x; s[1] = a.y; s[2] = b.x; s[3] = b.y; s[4] = d.x; s[5] = d.y; s[6] = e.x; s[7] = e.y;
    s[8] = s[9] = s[10] = s[11] = 0;
}
__global__ __launch_bounds__(256, 4) void merkle_level_kernel(const u64 *__restrict__ child, u64 *__restrict__ parent, size_t n_parents, u64 force_redo) {
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_parents) return;
    u64 s[12];
    load_children(child, j, s);
    psd::core<false> pc; pc.fl = 0;
    pc.run(s);
    if (pc.fl | force_redo) {      // a skipped rare repair somewhere in this wave: redo exactly (poseidon.cuh)
        load_children(child, j, s);
        psd::permute(s);
    }
    ulonglong2 *o = (ulonglong2 *)(parent + 4 * j);
    o[0] = make_ulonglong2(s[0], s[1]);
    o[1] = make_ulonglong2(s[2], s[3]);
}

// small levels: one parent per 16-lane group (lane-cooperative permutation) -- about 7x lower latency than one thread per
// parent, which is what matters once a level no longer fills the chip
__global__ __launch_bounds__(256) void merkle_level_coop_kernel(const u64 *__restrict__ child, u64 *__restrict__ parent, size_t n_parents) {
    const int lane = threadIdx.x & 63, l = lane & 15, gb = lane & ~15;
    size_t j = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const bool live = j < n_parents;
    u64 s = 0;
    if (live && l < 8) s = child[8 * j + l];
    s = psdc::permute16(s, l, gb);
    if (live && l < 4) parent[4 * j + l] = s;
}

// smallest levels: one parent per WAVE (poseidon_wave.cuh) -- the lowest latency per permutation there is on this chip, for the
// levels that no longer give every SIMD a wave anyway
__global__ __launch_bounds__(256) void merkle_level_wave_kernel(const u64 *__restrict__ child, u64 *__restrict__ parent, size_t n_parents, u64 force_redo) {
    __shared__ u64 wtab[psdw::TAB_WORDS];
    psdw::fill<256>(wtab); __syncthreads();
    const int lane = threadIdx.x & 63;
    const size_t j = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);        // wave-uniform
    if (j >= n_parents) return;
    u64 s = lane < 8 ? child[8 * j + lane] : 0;
    s = psdw::permute(s, lane, wtab, force_redo);
    if (lane < 4) parent[4 * j + lane] = s;
}

// Several of the smallest levels in ONE launch: a workgroup of 2^(LV-1) waves owns 2^LV adjacent nodes of the level below and walks up to
// their one ancestor -- wave w hashes parent w, the digests meet in LDS, half the waves go on.  Every level is also written to the
// levels array (the tree keeps all of them for the openings).  A level of the one-parent-per-wave kernel is one permutation (7.5 us)
// plus a launch and the 28 KB table image (as much again): the eight levels below 2 048 parents of a tree are two launches here.
template <int LV>
__global__ __launch_bounds__(64 << (LV - 1)) void merkle_tail_kernel(const u64 *__restrict__ child, u64 *__restrict__ parent, size_t n_parents, u64 force_redo) {
    constexpr int NW = 1 << (LV - 1);
    __shared__ u64 wtab[psdw::TAB_WORDS];
    __shared__ u64 dig[2][NW * 4];
    psdw::fill<(64 << (LV - 1))>(wtab); __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;            // w: wave-uniform
    size_t cnt = n_parents;                                              // a multiple of NW (launcher)
    u64 *out = parent;
    {
        const size_t j = (size_t)blockIdx.x * NW + w;
        u64 s = lane < 8 ? child[8 * j + lane] : 0;
        s = psdw::permute(s, lane, wtab, force_redo);
        if (lane < 4) { out[4 * j + lane] = s; dig[0][4 * w + lane] = s; }
    }
#pragma unroll
    for (int lv = 1; lv < LV; lv++) {
        __syncthreads();
        out += 4 * cnt; cnt >>= 1;
        const int active = NW >> lv;
        if (w < active) {
            u64 s = lane < 8 ? dig[(lv - 1) & 1][8 * w + lane] : 0;
            s = psdw::permute(s, lane, wtab, force_redo);
            const size_t j = (size_t)blockIdx.x * active + w;
            if (lane < 4) { out[4 * j + lane] = s; dig[lv & 1][4 * w + lane] = s; }
        }
    }
}

__global__ __launch_bounds__(256) void permute_batch_kernel(u64 *states, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u64 s[12];
#pragma unroll
    for (int k = 0; k < 12; k++) s[k] = gl_canon(states[12 * i + k]);
    psd::permute(s);
#pragma unroll
    for (int k = 0; k < 12; k++) states[12 * i + k] = s[k];
}

__global__ __launch_bounds__(256) void two_to_one_batch_kernel(const u64 *l, const u64 *r, size_t n, u64 *out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u64 s[12];
#pragma unroll
    for (int k = 0; k < 4; k++) { s[k] = gl_canon(l[4 * i + k]); s[4 + k] = gl_canon(r[4 * i + k]); s[8 + k] = 0; }
    psd::permute(s);
#pragma unroll
    for (int k = 0; k < 4; k++) out[4 * i + k] = s[k];
}

// upstream's recursive digest layout: position of node j at `depth` below a cap node over m leaves
__device__ __forceinline__ size_t layout_off(size_t m, u32 depth, size_t j) {
    size_t off = 0;
    while (depth > 1) {
        size_t half_nodes = (size_t)1 << (depth - 1);
        if (j >= half_nodes) { off += m; j -= half_nodes; }
        m >>= 1; depth--;
    }
    return off + (m - 2) + j;
}
// levels (level-order) -> digests (upstream layout)
__global__ void digests_layout_kernel(const u64 *levels, size_t n_leaves, u32 n_levels, u64 *digests) {
    // thread = one digest of levels 0 .. n_levels-2
    size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t total = 2 * (n_leaves - (n_leaves >> (n_levels - 1)));
    if (g >= total) return;
    u32 k = 0; size_t off = 0, cnt = n_leaves;
    while (g >= off + cnt) { off += cnt; cnt >>= 1; k++; }
    size_t i = g - off;
    u32 depth = n_levels - 1 - k;
    size_t ncap = n_leaves >> (n_levels - 1), m = n_leaves / ncap, sub_len = 2 * m - 2;
    size_t per_sub = (size_t)1 << depth;
    size_t sub = i / per_sub, j = i % per_sub;
    size_t dst = sub * sub_len + layout_off(m, depth, j);
#pragma unroll
    for (int w = 0; w < 4; w++) digests[4 * dst + w] = levels[4 * g + w];
}

static inline dim3 grid1(size_t n) { return dim3((unsigned)((n + 255) / 256)); }

int32_t zk_hash_leaves_colmajor(zk_ctx *ctx, const u64 *cols, size_t stride, uint32_t n_cols, size_t n_leaves, u64 *digests) {
    // one profiler label per launch shape (the sponge length is what a launch costs): poseidon_leaf_hash_c93, .._c40, and
    // poseidon_leaf_copy_c4 for the hash_or_noop copy branch, so that an average launch duration means one thing
    char label[48];
    snprintf(label, sizeof label, n_cols > 4 ? "poseidon_leaf_hash_c%u" : "poseidon_leaf_copy_c%u", n_cols);
    if ((ctx->poseidon_mfma & 1) && n_cols > 4 && n_leaves >= PSDM_MIN_UNITS && n_leaves % 256 == 0) {
        const PsdmTables *T = psdm_tables(ctx);
        if (!T) return ZK_ERR_OOM;
        if (n_leaves < 65536 && !(ctx->poseidon_mfma & 16))      // fewer pair-layout waves than SIMDs: 32 leaves per wave (bit 4 of the switch: off)
            ZK_LAUNCH(ctx, label, hash_leaves_colmajor_mfma32_kernel, dim3((unsigned)(n_leaves / 128)), dim3(256), 0, T, cols, stride, n_cols,
                      n_leaves, digests, ctx->force_redo);
        else
            ZK_LAUNCH(ctx, label, hash_leaves_colmajor_mfma_kernel, dim3((unsigned)(n_leaves / 256)), dim3(256), 0, T, cols, stride, n_cols,
                      n_leaves, digests, ctx->force_redo);
        return ZK_OK;
    }
    if (n_cols > 4 && !(ctx->poseidon_mfma & 8)) {       // bit 3 of ZK_POSEIDON_MFMA: no lane-cooperative kernels (thread per leaf instead)
        if (n_leaves <= PSD_SMALL_WAVE_MAX) {
            ZK_LAUNCH(ctx, label, hash_leaves_colmajor_small_kernel<true>, dim3((unsigned)((n_leaves + 3) / 4)), dim3(256), 0, cols, stride, n_cols, n_leaves, digests, ctx->force_redo);
            return ZK_OK;
        }
        if (n_leaves <= PSD_SMALL_COOP_MAX) {
            ZK_LAUNCH(ctx, label, hash_leaves_colmajor_small_kernel<false>, grid1(n_leaves * 16), dim3(256), 0, cols, stride, n_cols, n_leaves, digests, ctx->force_redo);
            return ZK_OK;
        }
    }
    ZK_LAUNCH(ctx, label, hash_leaves_colmajor_kernel, grid1(n_leaves), dim3(256), 0, cols, stride, n_cols, n_leaves, digests, ctx->force_redo);
    return ZK_OK;
}
int32_t zk_hash_leaves_rowmajor(zk_ctx *ctx, const u64 *rows, uint32_t leaf_len, size_t n_leaves, u64 *digests) {
    ZK_LAUNCH(ctx, "poseidon_leaf_hash_rows", hash_leaves_rowmajor_kernel, grid1(n_leaves), dim3(256), 0, rows, leaf_len, n_leaves, digests, 1, ctx->force_redo);
    return ZK_OK;
}
// true when the matrix-core kernel took the launch (big layers whose arity is a multiple of four); false: caller uses its own kernel
bool zk_fri_leaf_hash_mfma(zk_ctx *ctx, const u64 *re, const u64 *im, size_t n_leaves, uint32_t arity, u64 *digests) {
    if (arity * 2 <= 4) return false;                                       // hash_or_noop's copy branch: the caller's kernel
    if (!(ctx->poseidon_mfma & 8) && n_leaves <= PSD_SMALL_COOP_MAX) {      // small layers: a wave / 16 lanes per leaf (see above)
        if (n_leaves <= PSD_SMALL_WAVE_MAX)
            ZK_LAUNCH(ctx, "fri_leaf_hash", fri_leaf_hash_small_kernel<true>, dim3((unsigned)((n_leaves + 3) / 4)), dim3(256), 0, re, im, n_leaves, arity, digests, ctx->force_redo);
        else
            ZK_LAUNCH(ctx, "fri_leaf_hash", fri_leaf_hash_small_kernel<false>, grid1(n_leaves * 16), dim3(256), 0, re, im, n_leaves, arity, digests, ctx->force_redo);
        return true;
    }
    if (!(ctx->poseidon_mfma & 1) || n_leaves < PSDM_MIN_UNITS || n_leaves % 256 != 0 || arity % 4 != 0) return false;
    const PsdmTables *T = psdm_tables(ctx);
    if (!T) return false;
    ZK_LAUNCH(ctx, "fri_leaf_hash", fri_leaf_hash_mfma_kernel, dim3((unsigned)(n_leaves / 256)), dim3(256), 0, T, re, im, n_leaves, arity, digests, ctx->force_redo);
    return true;
}
int32_t zk_merkle_levels(zk_ctx *ctx, u64 *levels, size_t n_leaves, uint32_t cap_height) {
    size_t cnt = n_leaves;
    u64 *cur = levels;
    while (cnt > ((size_t)1 << cap_height)) {
        u64 *nxt = cur + 4 * cnt;
        if ((ctx->poseidon_mfma & 2) && cnt / 2 >= PSDM_MIN_UNITS && (cnt / 2) % 256 == 0) {
            const PsdmTables *T = psdm_tables(ctx);
            if (!T) return ZK_ERR_OOM;
            if (cnt / 2 < 65536 && !(ctx->poseidon_mfma & 16))
                ZK_LAUNCH(ctx, "poseidon_merkle_level", merkle_level_mfma32_kernel, dim3((unsigned)(cnt / 2 / 128)), dim3(256), 0, T, (const u64 *)cur, nxt, cnt / 2,
                          ctx->force_redo);
            else
                ZK_LAUNCH(ctx, "poseidon_merkle_level", merkle_level_mfma_kernel, dim3((unsigned)(cnt / 2 / 256)), dim3(256), 0, T, (const u64 *)cur, nxt, cnt / 2,
                          ctx->force_redo);
        } else if (cnt / 2 <= 2048 && !(ctx->poseidon_mfma & 8)) {
            // levels left down to the cap; up to five of them in one launch while the parents divide into whole workgroups
            // ZK_MERKLE_TAIL = most levels per launch (2..5; 0 = one launch per level).  OFF by default: measured, a transaction alone gains
            // 0.05 ms per Plonk proof, and with eight transactions in flight the mostly idle 16-wave workgroups block whole compute units:
            // 23.0 -> 17.5 tx/s at 5 levels
            static const int tail = getenv("ZK_MERKLE_TAIL") ? atoi(getenv("ZK_MERKLE_TAIL")) : 0;
            int left = 0;
            for (size_t c = cnt; c > ((size_t)1 << cap_height); c >>= 1) left++;
            int lv = left < tail ? left : tail;
            if (lv > 5) lv = 5;
            while (lv > 1 && (cnt / 2) % ((size_t)1 << (lv - 1))) lv--;
            if (tail >= 2 && lv >= 2) {
                const unsigned nw = 1u << (lv - 1), grid = (unsigned)((cnt / 2) / nw);
                switch (lv) {
                case 5: ZK_LAUNCH(ctx, "poseidon_merkle_level_wave", merkle_tail_kernel<5>, dim3(grid), dim3(64 * nw), 0, (const u64 *)cur, nxt, cnt / 2, ctx->force_redo); break;
                case 4: ZK_LAUNCH(ctx, "poseidon_merkle_level_wave", merkle_tail_kernel<4>, dim3(grid), dim3(64 * nw), 0, (const u64 *)cur, nxt, cnt / 2, ctx->force_redo); break;
                case 3: ZK_LAUNCH(ctx, "poseidon_merkle_level_wave", merkle_tail_kernel<3>, dim3(grid), dim3(64 * nw), 0, (const u64 *)cur, nxt, cnt / 2, ctx->force_redo); break;
                default: ZK_LAUNCH(ctx, "poseidon_merkle_level_wave", merkle_tail_kernel<2>, dim3(grid), dim3(64 * nw), 0, (const u64 *)cur, nxt, cnt / 2, ctx->force_redo); break;
                }
                for (int q = 0; q < lv; q++) { cur += 4 * cnt; cnt >>= 1; }
                continue;
            }
            ZK_LAUNCH(ctx, "poseidon_merkle_level_wave", merkle_level_wave_kernel, dim3((unsigned)((cnt / 2 + 3) / 4)), dim3(256), 0, (const u64 *)cur, nxt, cnt / 2, ctx->force_redo);
        } else if (cnt / 2 <= 16384)
            ZK_LAUNCH(ctx, "poseidon_merkle_level_coop", merkle_level_coop_kernel, grid1(cnt / 2 * 16), dim3(256), 0, (const u64 *)cur, nxt, cnt / 2);
        else
            ZK_LAUNCH(ctx, "poseidon_merkle_level", merkle_level_kernel, grid1(cnt / 2), dim3(256), 0, (const u64 *)cur, nxt, cnt / 2, ctx->force_redo);
        cur = nxt; cnt >>= 1;
    }
    return ZK_OK;
}
int32_t zk_digests_upstream_layout(zk_ctx *ctx, const u64 *levels, size_t n_leaves, uint32_t n_levels, u64 *d_digests) {
    size_t total = 2 * (n_leaves - (n_leaves >> (n_levels - 1)));
    if (total) ZK_LAUNCH(ctx, "digests_layout", digests_layout_kernel, grid1(total), dim3(256), 0, levels, n_leaves, n_levels, d_digests);
    return ZK_OK;
}

static uint32_t ilog2_exact(size_t x) { uint32_t l = 0; while (((size_t)1 << l) < x) l++; return ((size_t)1 << l) == x ? l : 0xFFFFFFFFu; }

extern "C" {

int32_t zk_poseidon_permute_batch(zk_ctx *ctx, uint64_t *states, size_t n) {
    if (!ctx || (!states && n)) return ZK_ERR_INVALID_ARG;
    if (!n) return ZK_OK;
    hipSetDevice(ctx->device);
    size_t bytes = n * 12 * sizeof(u64);
    u64 *d = (u64 *)ctx->alloc(bytes);
    if (!d) return ZK_ERR_OOM;
    int32_t rc = zk_dev_upload(ctx, d, states, bytes);
    if (rc == ZK_OK) {
        ZK_LAUNCH(ctx, "poseidon_permute_batch", permute_batch_kernel, grid1(n), dim3(256), 0, d, n);
        rc = zk_dev_download(ctx, states, d, bytes);
    }
    ctx->release(d);
    return rc;
}

int32_t zk_hash_no_pad_batch(zk_ctx *ctx, const uint64_t *rows, size_t n_rows, uint32_t row_len, uint64_t *out) {
    if (!ctx || !rows || !out || row_len == 0) return ZK_ERR_INVALID_ARG;
    if (!n_rows) return ZK_OK;
    hipSetDevice(ctx->device);
    size_t bytes = n_rows * row_len * sizeof(u64);
    u64 *d = (u64 *)ctx->alloc(bytes), *o = (u64 *)ctx->alloc(n_rows * 4 * sizeof(u64));
    int32_t rc = (d && o) ? zk_dev_upload(ctx, d, rows, bytes) : ZK_ERR_OOM;
    if (rc == ZK_OK) rc = zk_hash_leaves_rowmajor(ctx, d, row_len, n_rows, o);
    if (rc == ZK_OK) rc = zk_dev_download(ctx, out, o, n_rows * 4 * sizeof(u64));
    ctx->release(d); ctx->release(o);
    return rc;
}

int32_t zk_two_to_one_batch(zk_ctx *ctx, const uint64_t *left, const uint64_t *right, size_t n, uint64_t *out) {
    if (!ctx || !left || !right || !out) return ZK_ERR_INVALID_ARG;
    if (!n) return ZK_OK;
    hipSetDevice(ctx->device);
    size_t bytes = n * 4 * sizeof(u64);
    u64 *l = (u64 *)ctx->alloc(bytes), *r = (u64 *)ctx->alloc(bytes), *o = (u64 *)ctx->alloc(bytes);
    int32_t rc = (l && r && o) ? ZK_OK : ZK_ERR_OOM;
    if (rc == ZK_OK) rc = zk_dev_upload(ctx, l, left, bytes);
    if (rc == ZK_OK) rc = zk_dev_upload(ctx, r, right, bytes);
    if (rc == ZK_OK) {
        ZK_LAUNCH(ctx, "poseidon_two_to_one", two_to_one_batch_kernel, grid1(n), dim3(256), 0, (const u64 *)l, (const u64 *)r, n, o);
        rc = zk_dev_download(ctx, out, o, bytes);
    }
    ctx->release(l); ctx->release(r); ctx->release(o);
    return rc;
}

int32_t zk_merkle_build(zk_ctx *ctx, const uint64_t *leaves, size_t n_leaves, uint32_t leaf_len, uint32_t cap_height,
                        uint64_t *cap_out, uint64_t *digests_out) {
    if (!ctx || !leaves || !cap_out || leaf_len == 0) return ZK_ERR_INVALID_ARG;
    uint32_t lg = ilog2_exact(n_leaves);
    if (lg == 0xFFFFFFFFu || cap_height > lg) return ZK_ERR_INVALID_ARG;
    hipSetDevice(ctx->device);
    uint32_t n_levels = lg - cap_height + 1;
    size_t n_dig = zk_level_offset(n_leaves, n_levels);
    size_t lbytes = n_leaves * leaf_len * sizeof(u64);
    u64 *d_leaves = (u64 *)ctx->alloc(lbytes), *levels = (u64 *)ctx->alloc(n_dig * 4 * sizeof(u64));
    int32_t rc = (d_leaves && levels) ? zk_dev_upload(ctx, d_leaves, leaves, lbytes) : ZK_ERR_OOM;
    if (rc == ZK_OK) rc = zk_hash_leaves_rowmajor(ctx, d_leaves, leaf_len, n_leaves, levels);
    if (rc == ZK_OK) rc = zk_merkle_levels(ctx, levels, n_leaves, cap_height);
    if (rc == ZK_OK) rc = zk_dev_download(ctx, cap_out, levels + 4 * zk_level_offset(n_leaves, n_levels - 1), ((size_t)4 << cap_height) * sizeof(u64));
    if (rc == ZK_OK && digests_out && n_levels > 1) {
        size_t total = 2 * (n_leaves - ((size_t)1 << cap_height));
        u64 *dd = (u64 *)ctx->alloc(total * 4 * sizeof(u64));
        if (!dd) rc = ZK_ERR_OOM;
        if (rc == ZK_OK) rc = zk_digests_upstream_layout(ctx, levels, n_leaves, n_levels, dd);
        if (rc == ZK_OK) rc = zk_dev_download(ctx, digests_out, dd, total * 4 * sizeof(u64));
        ctx->release(dd);
    }
    ctx->release(d_leaves); ctx->release(levels);
    return rc;
}

}  // extern "C"
