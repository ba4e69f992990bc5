// Context, stream-ordered caching allocator, per-kernel hipEvent profiler, device-memory helpers.
// Boundary: include/zkgl.h "context" group.  One ctx = one device + one HIP stream = one worker thread of the
// reference's runtime (paladin `--num-workers N`, /root/reference/README.md:60-63).
#include <stdlib.h>
#include <string.h>
#include "zk_internal.h"
#include "transcript.h"

void *zk_ctx::alloc(size_t bytes) {
    if (bytes == 0) bytes = 256;
    bytes = (bytes + 255) & ~(size_t)255;
    auto it = free_blocks.lower_bound(bytes);
    if (it != free_blocks.end() && it->first <= bytes + bytes / 8) {
        void *p = it->second;
        free_blocks.erase(it);
        return p;
    }
    void *p = nullptr;
    if (efence) {
        // debug allocator (env ZK_ALLOC_EFENCE=1): every block ends exactly at the end of its own mapping and the next
        // virtual page is reserved but unmapped, so any read or write past the end of a buffer faults at the kernel that does it
        // (use with ZK_SYNC_LAUNCH=1 to get its name).  Blocks are never returned to the driver.
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = device;
        size_t gran = 0;
        if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum) != hipSuccess || !gran) gran = 2u << 20;
        size_t mapped = (bytes + gran - 1) / gran * gran;
        void *va = nullptr; hipMemGenericAllocationHandle_t h;
        hipMemAccessDesc ad = {}; ad.location = prop.location; ad.flags = hipMemAccessFlagsProtReadWrite;
        if (efence == 2) {
            // ZK_ALLOC_EFENCE=2: every block STRADDLES a 4 GiB boundary, the boundary in its middle.  The faults of rounds 1-4 were pointers that
            // lost the carry into their upper half (an asm block overwrote SCC between the compiler's s_add_u32 and s_addc_u32, DESIGN.md
            // section 9.4): they needed a block that happens to straddle such a boundary, about one allocation in a thousand.  Under this
            // allocator every kernel of a test run meets that layout in every buffer.
            const size_t G4 = (size_t)1 << 32, half_lo = (bytes / 2) & ~(size_t)255;
            const size_t k_lo = (half_lo + gran - 1) / gran, k_hi = (bytes - half_lo + gran - 1) / gran;
            mapped = (k_lo + k_hi) * gran;
            if (hipMemAddressReserve(&va, G4 + mapped, gran, nullptr, 0) != hipSuccess) { last_error = "straddle allocation: address range"; return nullptr; }
            const size_t B = ((size_t)va + k_lo * gran + G4 - 1) & ~(G4 - 1);
            void *m = (void *)(B - k_lo * gran);
            if (hipMemCreate(&h, mapped, &prop, 0) != hipSuccess || hipMemMap(m, mapped, 0, h, 0) != hipSuccess || hipMemSetAccess(m, mapped, &ad, 1) != hipSuccess) {
                (void)hipGetLastError();
                hipMemAddressFree(va, G4 + mapped);            // the reservation does not outlive a failed mapping
                last_error = "straddle allocation failed"; return nullptr;
            }
            p = (void *)(B - half_lo);
            block_size[p] = bytes; bytes_allocated += bytes;
            ZK_ALLOC_TRACE("dev-block-straddling", p, bytes);
            if (sync_launch) fprintf(stderr, "[zk %p] straddle %zu at %p .. %p\n", (void *)this, bytes, p, (void *)((char *)p + bytes));
            return p;
        }
        if (hipMemAddressReserve(&va, mapped + gran, gran, nullptr, 0) != hipSuccess || hipMemCreate(&h, mapped, &prop, 0) != hipSuccess ||
            hipMemMap(va, mapped, 0, h, 0) != hipSuccess || hipMemSetAccess(va, mapped, &ad, 1) != hipSuccess) {
            last_error = "efence allocation failed"; return nullptr;
        }
        p = (char *)va + (mapped - bytes);
        block_size[p] = bytes; bytes_allocated += bytes;
        if (sync_launch) fprintf(stderr, "[zk %p] efence %zu at %p .. %p\n", (void *)this, bytes, p, (void *)((char *)p + bytes));
        return p;
    }
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {
        // give cached blocks back to the driver and retry once
        for (auto &kv : free_blocks) { ZK_ALLOC_TRACE("free-dev-oom", kv.second, kv.first); hipFree(kv.second); block_size.erase(kv.second); bytes_allocated -= kv.first; }
        free_blocks.clear();
        e = hipMalloc(&p, bytes);
        if (e != hipSuccess) { last_error = std::string("hipMalloc: ") + hipGetErrorString(e); return nullptr; }
    }
    block_size[p] = bytes;
    bytes_allocated += bytes;
    if (sync_launch) fprintf(stderr, "[zk] hipMalloc %zu at %p .. %p\n", bytes, p, (void *)((char *)p + bytes));
    ZK_ALLOC_TRACE("dev-block", p, bytes);
    return p;
}

const void *zk_ctx::stage_pinned(const void *src, size_t bytes) {
    if (!h_stage) {
        if (hipHostMalloc((void **)&h_stage, (size_t)1 << 20, hipHostMallocDefault) != hipSuccess) { h_stage = nullptr; return nullptr; }
        h_stage_cap = (size_t)1 << 20; h_stage_used = 0; ZK_ALLOC_TRACE("host-stage", h_stage, h_stage_cap);
    }
    size_t need = (bytes + 63) & ~(size_t)63;
    if (h_stage_used + need > h_stage_cap) return nullptr;
    char *p = h_stage + h_stage_used;
    h_stage_used += need;
    memcpy(p, src, bytes);
    return p;
}
void *zk_ctx::pinned_out(size_t bytes) {
    if (h_out_cap >= bytes) return h_out;
    if (h_out) { hipStreamSynchronize(stream); ZK_ALLOC_TRACE("free-host-out", h_out, h_out_cap); hipHostFree(h_out); h_out = nullptr; h_out_cap = 0; }
    size_t cap = (bytes + ((size_t)1 << 16)) & ~(((size_t)1 << 16) - 1);
    if (hipHostMalloc((void **)&h_out, cap, hipHostMallocDefault) != hipSuccess) { h_out = nullptr; return nullptr; }
    h_out_cap = cap; ZK_ALLOC_TRACE("host-out", h_out, cap);
    return h_out;
}
int32_t zk_ctx::upload_small(void *d_dst, const void *src, size_t bytes) {
    if (!bytes) return ZK_OK;
    if (const void *p = stage_pinned(src, bytes)) {
        if (hipMemcpyAsync(d_dst, p, bytes, hipMemcpyHostToDevice, stream) != hipSuccess) return fail(ZK_ERR_HIP, "upload_small: hipMemcpyAsync");
        return ZK_OK;
    }
    // staging full: through the bounce buffer, and wait for it
    return copy_h2d_sync(d_dst, src, bytes);
}

// Synchronous copies between caller memory and the device.  The caller's memory is pageable as a rule (numpy arrays, std::vector): handed
// to hipMemcpyAsync directly, the runtime pins its pages on the fly and releases them when the copy retires -- with dozens of threads doing
// that at once (eight transactions x four contexts: Merkle roots of child proofs, public-input hashes, caps coming back) that is a lot of
// registration traffic in the driver.  Every such copy bounces through pinned memory owned by the context, 4 MiB at a time; big uploads keep
// the double-buffered path of upload_large.  (Round 4 suspected these copies of a GPU memory fault; the fault was a lost pointer carry in
// three kernels, DESIGN.md section 9 -- the bounce buffers stay because they keep the runtime's pin / unpin path out of the proofs.)
#define ZK_IO_BYTES ((size_t)4 << 20)
static bool io_buffer(zk_ctx *c) {
    if (c->h_io) return true;
    if (hipHostMalloc((void **)&c->h_io, ZK_IO_BYTES, hipHostMallocDefault) != hipSuccess) { c->h_io = nullptr; (void)hipGetLastError(); return false; }
    ZK_ALLOC_TRACE("host-io", c->h_io, ZK_IO_BYTES);
    return true;
}
int32_t zk_ctx::copy_h2d_sync(void *d_dst, const void *src, size_t bytes) {
    if (!bytes) return ZK_OK;
    if (bytes > 4 * ZK_IO_BYTES) {           // a trace-sized array: pinned source -> one copy, pageable -> two 32 MiB chunks in turn
        int32_t rc = upload_large(d_dst, src, bytes, stream);
        if (rc != ZK_OK) return rc;
        return hipStreamSynchronize(stream) == hipSuccess ? ZK_OK : fail(ZK_ERR_HIP, "copy_h2d_sync: stream sync");
    }
    if (!io_buffer(this)) return fail(ZK_ERR_OOM, "copy_h2d_sync: pinned bounce buffer");
    for (size_t off = 0; off < bytes; off += ZK_IO_BYTES) {
        const size_t len = bytes - off < ZK_IO_BYTES ? bytes - off : ZK_IO_BYTES;
        memcpy(h_io, (const char *)src + off, len);
        if (hipMemcpyAsync((char *)d_dst + off, h_io, len, hipMemcpyHostToDevice, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess)
            return fail(ZK_ERR_HIP, "copy_h2d_sync: copy");
    }
    return ZK_OK;
}
int32_t zk_ctx::copy_d2h_sync(void *dst, const void *d_src, size_t bytes) {
    if (!bytes) return ZK_OK;
    if (!io_buffer(this)) return fail(ZK_ERR_OOM, "copy_d2h_sync: pinned bounce buffer");
    for (size_t off = 0; off < bytes; off += ZK_IO_BYTES) {
        const size_t len = bytes - off < ZK_IO_BYTES ? bytes - off : ZK_IO_BYTES;
        if (hipMemcpyAsync(h_io, (const char *)d_src + off, len, hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess)
            return fail(ZK_ERR_HIP, "copy_d2h_sync: copy");
        memcpy((char *)dst + off, h_io, len);
    }
    return ZK_OK;
}

#define ZK_CHUNK_BYTES ((size_t)32 << 20)
int32_t zk_ctx::upload_large(void *d_dst, const void *src, size_t bytes, hipStream_t on) {
    if (!bytes) return ZK_OK;
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, src) == hipSuccess && (at.type == hipMemoryTypeHost || at.type == hipMemoryTypeDevice || at.type == hipMemoryTypeManaged)) {
        // pinned (or device) source: asynchronous copies, the caller keeps the buffer alive until the work has been waited for.
        // One copy on one stream by default: on the MI355X boxes measured it carries the link's full rate (53 GB/s for a 780 MB trace)
        // and overlaps the proofs of the other streams (PCIe-inclusive rate 94 % of the resident-trace rate).  ZK_UPLOAD_PARTS=2..4
        // cuts a large upload into parts on side streams that start behind `on` and that `on` waits for afterwards -- for hosts
        // where one DMA engine does not fill the link; on the boxes above it LOWERS the overlapped rate (31 instead of 42 proofs/s).
        const int parts_env = getenv("ZK_UPLOAD_PARTS") ? atoi(getenv("ZK_UPLOAD_PARTS")) : 1;
        const int parts = (bytes >= ((size_t)64 << 20) && parts_env > 1) ? (parts_env > 4 ? 4 : parts_env) : 1;
        if (parts > 1) {
            for (int i = 0; i < parts - 1; i++)
                if (!part_stream[i] && hipStreamCreateWithFlags(&part_stream[i], hipStreamNonBlocking) != hipSuccess) { part_stream[i] = nullptr; return fail(ZK_ERR_HIP, "upload_large: side stream"); }
            auto get_event = [&]() { if (!event_pool.empty()) { hipEvent_t e = event_pool.back(); event_pool.pop_back(); return e; } hipEvent_t e = nullptr; hipEventCreate(&e); return e; };
            hipEvent_t start = get_event();
            if (!start || hipEventRecord(start, on) != hipSuccess) return fail(ZK_ERR_HIP, "upload_large: event");
            const size_t per = ((bytes / parts) + 4095) & ~(size_t)4095;
            for (int i = 0; i < parts; i++) {
                const size_t off = (size_t)i * per, len = i == parts - 1 ? bytes - off : per;
                hipStream_t st = i == 0 ? on : part_stream[i - 1];
                if (i && hipStreamWaitEvent(st, start, 0) != hipSuccess) return fail(ZK_ERR_HIP, "upload_large: ordering");
                if (hipMemcpyAsync((char *)d_dst + off, (const char *)src + off, len, hipMemcpyDefault, st) != hipSuccess) return fail(ZK_ERR_HIP, "upload_large: hipMemcpyAsync");
                if (i) {
                    hipEvent_t done = get_event();
                    if (!done || hipEventRecord(done, st) != hipSuccess || hipStreamWaitEvent(on, done, 0) != hipSuccess) return fail(ZK_ERR_HIP, "upload_large: join");
                    event_pool.push_back(done);
                }
            }
            event_pool.push_back(start);
            return ZK_OK;
        }
        if (hipMemcpyAsync(d_dst, src, bytes, hipMemcpyDefault, on) != hipSuccess) return fail(ZK_ERR_HIP, "upload_large: hipMemcpyAsync");
        return ZK_OK;
    }
    (void)hipGetLastError();                                   // an unregistered pointer is reported as an error: pageable memory
    for (int i = 0; i < 2; i++) {
        if (!h_chunk[i] && hipHostMalloc((void **)&h_chunk[i], ZK_CHUNK_BYTES, hipHostMallocDefault) != hipSuccess) { h_chunk[i] = nullptr; return fail(ZK_ERR_OOM, "upload_large: pinned chunk"); }
        if (!chunk_free[i] && hipEventCreateWithFlags(&chunk_free[i], hipEventDisableTiming) != hipSuccess) return fail(ZK_ERR_HIP, "upload_large: event");
    }
    size_t off = 0; int i = 0;
    while (off < bytes) {
        const size_t len = bytes - off < ZK_CHUNK_BYTES ? bytes - off : ZK_CHUNK_BYTES;
        if (hipEventSynchronize(chunk_free[i]) != hipSuccess) return fail(ZK_ERR_HIP, "upload_large: event wait");   // never recorded: returns at once
        memcpy(h_chunk[i], (const char *)src + off, len);
        if (hipMemcpyAsync((char *)d_dst + off, h_chunk[i], len, hipMemcpyHostToDevice, on) != hipSuccess || hipEventRecord(chunk_free[i], on) != hipSuccess)
            return fail(ZK_ERR_HIP, "upload_large: chunk copy");
        off += len; i ^= 1;
    }
    return ZK_OK;
}
void zk_ctx::wait_upload(const void *d_ptr) {
    // only the prefetch that fills THIS buffer: the next unit's upload, already under way on the copy stream, must not hold the
    // current unit's proof back
    for (size_t i = 0; i < pending_uploads.size();) {
        if (pending_uploads[i].first == d_ptr) {
            hipStreamWaitEvent(stream, pending_uploads[i].second, 0);
            event_pool.push_back(pending_uploads[i].second);
            pending_uploads.erase(pending_uploads.begin() + i);
        } else i++;
    }
}

hipError_t zk_ctx::wait_stream() {
    static const int blocking = getenv("ZK_BLOCKING_SYNC") ? atoi(getenv("ZK_BLOCKING_SYNC")) : 0;
    if (!blocking) return hipStreamSynchronize(stream);
    if (!sync_event && hipEventCreateWithFlags(&sync_event, hipEventBlockingSync | hipEventDisableTiming) != hipSuccess) { sync_event = nullptr; return hipStreamSynchronize(stream); }
    hipError_t e = hipEventRecord(sync_event, stream);
    if (e != hipSuccess) return e;
    return hipEventSynchronize(sync_event);
}

void zk_ctx::release(void *p) {
    if (!p) return;
    auto it = block_size.find(p);
    if (it == block_size.end()) return;
    free_blocks.emplace(it->second, p);
}

ProfScope::ProfScope(zk_ctx *c, const char *name) : ctx(c), idx(-1), e0(nullptr), e1(nullptr) {
    if (!c->prof) return;
    auto it = c->prof_index.find(name);
    if (it == c->prof_index.end()) {
        idx = (int)c->prof_recs.size();
        c->prof_index[name] = idx;
        c->prof_recs.emplace_back();
        c->prof_recs.back().name = name;
    } else idx = it->second;
    auto get = [&]() {
        if (!c->event_pool.empty()) { hipEvent_t e = c->event_pool.back(); c->event_pool.pop_back(); return e; }
        hipEvent_t e; hipEventCreate(&e); return e;
    };
    e0 = get(); e1 = get();
    hipEventRecord(e0, c->stream);
}
ProfScope::~ProfScope() {
    if (idx < 0) return;
    hipEventRecord(e1, ctx->stream);
    ctx->prof_recs[idx].pending.emplace_back(e0, e1);
    ctx->prof_recs[idx].launches++;
}
void zk_ctx::prof_flush() {
    for (auto &r : prof_recs) {
        for (auto &pr : r.pending) {
            float ms = 0.f;
            hipEventSynchronize(pr.second);
            hipEventElapsedTime(&ms, pr.first, pr.second);
            r.total_ms += ms;
            event_pool.push_back(pr.first); event_pool.push_back(pr.second);
        }
        r.pending.clear();
    }
}

// zk_trace_upload_packed: column j of the packed buffer (width 1 / 2 / 4 / 8 bytes per row, little-endian) -> column j of the usual u64 layout
__global__ void trace_widen_kernel(const unsigned char *__restrict__ packed, const u64 *__restrict__ off, size_t n, u64 *__restrict__ out) {
    const size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const u32 j = blockIdx.y;
    const u64 o = off[j];
    const u32 w = (u32)(o >> 56);
    const unsigned char *src = packed + (o & 0x00FFFFFFFFFFFFFFull);
    u64 v;
    if (w == 1) v = src[r];
    else if (w == 2) v = ((const uint16_t *)src)[r];
    else if (w == 4) v = ((const uint32_t *)src)[r];
    else v = gl_canon(((const u64 *)src)[r]);
    out[(size_t)j * n + r] = v;
}

extern "C" {

int32_t zk_ctx_create(int32_t device, void *stream, zk_ctx **out) {
    if (!out) return ZK_ERR_INVALID_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return ZK_ERR_NO_DEVICE;   // no CPU fallback
    if (device < 0 || device >= count) return ZK_ERR_INVALID_ARG;
    if (hipSetDevice(device) != hipSuccess) return ZK_ERR_HIP;
    zk_ctx *c = new zk_ctx();
    c->device = device;
    { const char *e = getenv("ZK_FORCE_EXACT_REDO"); c->force_redo = e ? (u64)atoi(e) : 0; }   // 1: every wave / slice; 2: as 1, but the compiled quotient kernel flags one slice in seven
    { const char *e = getenv("ZK_QUOTIENT_JIT"); if (e) c->quotient_jit_mode = atoi(e); }
    { const char *e = getenv("ZK_QUOTIENT_AOT"); if (e) c->quotient_aot = atoi(e); }
    { const char *e = getenv("ZK_POSEIDON_MFMA"); if (e) c->poseidon_mfma = atoi(e); }
    { const char *e = getenv("ZK_SYNC_LAUNCH"); if (e) c->sync_launch = atoi(e); }
    { const char *e = getenv("ZK_ALLOC_EFENCE"); if (e) c->efence = atoi(e); }
    { const char *e = getenv("ZK_QUOTIENT_JIT_MIN_LOG"); if (e) c->quotient_jit_min_log = atoi(e); }
    { const char *e = getenv("ZK_TRANSCRIPT_HOST"); if (e) c->host_transcript = atoi(e) ? 1 : 0; }
    if (stream) { c->stream = (hipStream_t)stream; c->own_stream = false; }
    else {
        if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return ZK_ERR_HIP; }
        c->own_stream = true;
    }
    if (hipMalloc((void **)&c->d_err, 256) != hipSuccess) { delete c; return ZK_ERR_OOM; }
    ZK_ALLOC_TRACE("dev-err-word", c->d_err, 256);
    hipMemsetAsync(c->d_err, 0, 256, c->stream);
    zk_fault_register_ctx(c);
    *out = c;
    return ZK_OK;
}

int32_t zk_ctx_trim(zk_ctx *c, uint64_t *bytes_freed) {
    if (!c) return ZK_ERR_INVALID_ARG;
    hipSetDevice(c->device);
    if (c->wait_stream() != hipSuccess) return c->fail(ZK_ERR_HIP, "trim: stream");
    uint64_t freed = 0;
    // the big NTT pass tables first (keys start with the transform's log size, ntt_fast.hip pass_table): back to the free list ...
    for (auto it = c->table_cache.begin(); it != c->table_cache.end();) {
        const int k = atoi(it->first.c_str() + (it->first.compare(0, 3, "cs.") == 0 ? 3 : 0));
        if (k > 24) { c->release(it->second); it = c->table_cache.erase(it); } else ++it;
    }
    // ... and the whole free list to the driver
    if (!c->efence) {
        for (auto &kv : c->free_blocks) { ZK_ALLOC_TRACE("free-dev-trim", kv.second, kv.first); hipFree(kv.second); c->block_size.erase(kv.second); c->bytes_allocated -= kv.first; freed += kv.first; }
        c->free_blocks.clear();
    }
    if (c->side) { uint64_t f2 = 0; zk_ctx_trim(c->side, &f2); freed += f2; }
    if (bytes_freed) *bytes_freed = freed;
    return ZK_OK;
}

int32_t zk_ctx_set_side_commit(zk_ctx *c, int32_t on) {
    if (!c) return ZK_ERR_INVALID_ARG;
    c->side_commit = on ? 1 : 0;
    return ZK_OK;
}

int32_t zk_ctx_set_host_transcript(zk_ctx *c, int32_t on) {
    if (!c) return ZK_ERR_INVALID_ARG;
    if (c->host_transcript == (on ? 1 : 0)) return ZK_OK;
    // between proofs only: the two challengers keep their sponge states in different places
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    zk_host_transcript_begin_proof(c);
    // challengers that outlive the switch (zk_challenger_new handles) take their sponge along; a proof's own challenger starts with
    // TOP_INIT on whichever side is current
    for (DevChallenger *d : c->live_challengers) {
        const int32_t rc = on ? zk_host_transcript_adopt(c, d) : zk_host_transcript_hand_back(c, d);
        if (rc != ZK_OK) return rc;
    }
    c->host_transcript = on ? 1 : 0;
    return ZK_OK;
}

int32_t zk_ctx_destroy(zk_ctx *c) {
    if (!c) return ZK_ERR_INVALID_ARG;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    zk_fault_unregister_ctx(c);
    zk_host_transcript_destroy(c);
    if (c->side) { zk_ctx_destroy(c->side); c->side = nullptr; }
    c->prof_flush();
    for (auto e : c->event_pool) hipEventDestroy(e);
    if (zk_alloc_trace_on()) fprintf(stderr, "[zk-alloc] ctx-destroy %p %zu blocks\n", (void *)c, c->block_size.size());     // stderr only: not a block of the flight recorder
    if (!c->efence) for (auto &kv : c->block_size) { ZK_ALLOC_TRACE("free-dev-block", kv.first, kv.second); hipFree(kv.first); }
    ZK_ALLOC_TRACE("free-dev-err-word", c->d_err, 256); hipFree(c->d_err);
    if (c->h_stage) { ZK_ALLOC_TRACE("free-host-stage", c->h_stage, 0); hipHostFree(c->h_stage); }
    if (c->h_out) { ZK_ALLOC_TRACE("free-host-out", c->h_out, c->h_out_cap); hipHostFree(c->h_out); }
    if (c->h_io) { ZK_ALLOC_TRACE("free-host-io", c->h_io, 0); hipHostFree(c->h_io); }
    if (c->sync_event) hipEventDestroy(c->sync_event);
    if (c->h_packed_off) hipHostFree(c->h_packed_off);
    if (c->packed_off_done) hipEventDestroy(c->packed_off_done);
    if (c->copy_stream) { hipStreamSynchronize(c->copy_stream); hipStreamDestroy(c->copy_stream); }
    for (int i = 0; i < 3; i++) if (c->part_stream[i]) { hipStreamSynchronize(c->part_stream[i]); hipStreamDestroy(c->part_stream[i]); }
    for (auto &pe : c->pending_uploads) hipEventDestroy(pe.second);
    for (int i = 0; i < 2; i++) { if (c->h_chunk[i]) hipHostFree(c->h_chunk[i]); if (c->chunk_free[i]) hipEventDestroy(c->chunk_free[i]); }
    if (c->own_stream) hipStreamDestroy(c->stream);
    delete c;
    return ZK_OK;
}

int32_t zk_ctx_synchronize(zk_ctx *c) {
    if (!c) return ZK_ERR_INVALID_ARG;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return ZK_OK;
}

const char *zk_last_error(zk_ctx *c) { return c ? c->last_error.c_str() : "null ctx"; }
void *zk_ctx_stream(zk_ctx *c) { return c ? (void *)c->stream : nullptr; }

int32_t zk_profile_enable(zk_ctx *c, int32_t on) { if (!c) return ZK_ERR_INVALID_ARG; c->prof = on != 0; if (c->side) c->side->prof = c->prof; return ZK_OK; }
int32_t zk_profile_reset(zk_ctx *c) {
    if (!c) return ZK_ERR_INVALID_ARG;
    hipStreamSynchronize(c->stream);
    c->prof_flush();
    c->prof_recs.clear(); c->prof_index.clear();
    if (c->side) zk_profile_reset(c->side);
    return ZK_OK;
}
int32_t zk_profile_count(zk_ctx *c) {
    if (!c) return ZK_ERR_INVALID_ARG;
    hipStreamSynchronize(c->stream);
    c->prof_flush();
    if (c->side) {
        // the side context's launches (zk_ctx_set_side_commit) belong to this context's proofs: fold its records into these
        hipStreamSynchronize(c->side->stream);
        c->side->prof_flush();
        for (auto &r : c->side->prof_recs) {
            auto it = c->prof_index.find(r.name);
            int idx;
            if (it == c->prof_index.end()) { idx = (int)c->prof_recs.size(); c->prof_index[r.name] = idx; c->prof_recs.emplace_back(); c->prof_recs.back().name = r.name; }
            else idx = it->second;
            c->prof_recs[idx].launches += r.launches; c->prof_recs[idx].total_ms += r.total_ms;
        }
        c->side->prof_recs.clear(); c->side->prof_index.clear();
    }
    return (int32_t)c->prof_recs.size();
}
int32_t zk_profile_get(zk_ctx *c, int32_t i, const char **name, uint64_t *launches, double *total_ms) {
    if (!c || i < 0 || i >= (int32_t)c->prof_recs.size()) return ZK_ERR_INVALID_ARG;
    if (name) *name = c->prof_recs[i].name.c_str();
    if (launches) *launches = c->prof_recs[i].launches;
    if (total_ms) *total_ms = c->prof_recs[i].total_ms;
    return ZK_OK;
}

int32_t zk_dev_alloc(zk_ctx *c, size_t bytes, void **d_ptr) {
    if (!c || !d_ptr) return ZK_ERR_INVALID_ARG;
    hipSetDevice(c->device);
    *d_ptr = c->alloc(bytes);
    return *d_ptr ? ZK_OK : ZK_ERR_OOM;
}
int32_t zk_dev_free(zk_ctx *c, void *d_ptr) {
    if (!c) return ZK_ERR_INVALID_ARG;
    c->wait_upload(d_ptr);                      // a prefetch into this buffer must land before the block can be reused
    c->release(d_ptr);
    return ZK_OK;
}
int32_t zk_host_alloc(zk_ctx *c, size_t bytes, void **h_ptr) {
    if (!c || !h_ptr) return ZK_ERR_INVALID_ARG;
    hipSetDevice(c->device);
    *h_ptr = nullptr;
    if (hipHostMalloc(h_ptr, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { *h_ptr = nullptr; return c->fail(ZK_ERR_OOM, "zk_host_alloc: hipHostMalloc"); }
    ZK_ALLOC_TRACE("host-caller", *h_ptr, bytes);
    return ZK_OK;
}
int32_t zk_host_free(zk_ctx *c, void *h_ptr) {
    if (!c) return ZK_ERR_INVALID_ARG;
    ZK_ALLOC_TRACE("free-host-caller", h_ptr, 0);
    if (h_ptr && hipHostFree(h_ptr) != hipSuccess) return c->fail(ZK_ERR_HIP, "zk_host_free");
    return ZK_OK;
}
int32_t zk_trace_upload_async(zk_ctx *c, const uint64_t *trace, size_t n_words, uint64_t **d_ptr) {
    if (!c || !trace || !d_ptr || !n_words) return ZK_ERR_INVALID_ARG;
    hipSetDevice(c->device);
    *d_ptr = nullptr;
    if (!c->copy_stream && hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking) != hipSuccess) { c->copy_stream = nullptr; return c->fail(ZK_ERR_HIP, "upload_async: copy stream"); }
    void *d = c->alloc(n_words * sizeof(uint64_t));
    if (!d) return c->fail(ZK_ERR_OOM, "upload_async: device memory");
    auto get_event = [&]() { if (!c->event_pool.empty()) { hipEvent_t e = c->event_pool.back(); c->event_pool.pop_back(); return e; } hipEvent_t e = nullptr; hipEventCreate(&e); return e; };
    // the block may have been released by work that is still enqueued on the main stream: the copy starts behind everything enqueued
    // there SO FAR (and therefore overlaps whatever the caller enqueues next, e.g. the current unit's proof)
    hipEvent_t before = get_event(), done = get_event();
    int32_t rc = ZK_OK;
    if (!before || !done || hipEventRecord(before, c->stream) != hipSuccess || hipStreamWaitEvent(c->copy_stream, before, 0) != hipSuccess) rc = c->fail(ZK_ERR_HIP, "upload_async: ordering");
    if (rc == ZK_OK) rc = c->upload_large(d, trace, n_words * sizeof(uint64_t), c->copy_stream);
    if (rc == ZK_OK && hipEventRecord(done, c->copy_stream) != hipSuccess) rc = c->fail(ZK_ERR_HIP, "upload_async: event");
    if (before) c->event_pool.push_back(before);
    if (rc != ZK_OK) { if (done) c->event_pool.push_back(done); hipStreamSynchronize(c->copy_stream); c->release(d); return rc; }
    c->pending_uploads.emplace_back(d, done);
    *d_ptr = (uint64_t *)d;
    return ZK_OK;
}
int32_t zk_trace_upload_packed(zk_ctx *c, const void *packed, uint32_t log_n, uint32_t n_cols, const uint8_t *col_bytes, uint64_t **d_ptr) {
    if (!c || !packed || !col_bytes || !d_ptr || n_cols == 0 || n_cols > ZK_PACKED_MAX_COLS || log_n > 28) return ZK_ERR_INVALID_ARG;
    hipSetDevice(c->device);
    *d_ptr = nullptr;
    const size_t n = (size_t)1 << log_n;
    std::vector<u64> off(n_cols + 1, 0);                     // byte offset of every column in the packed buffer | its width in the top byte
    size_t total = 0;
    for (u32 j = 0; j < n_cols; j++) {
        const u32 w = col_bytes[j];
        if (w != 1 && w != 2 && w != 4 && w != 8) return c->fail(ZK_ERR_INVALID_ARG, "upload_packed: column width must be 1, 2, 4 or 8 bytes");
        total = (total + w - 1) / w * w;                     // every column starts at a multiple of its element size
        off[j] = (u64)total | ((u64)w << 56);
        total += n * w;
    }
    u64 *d = (u64 *)c->alloc(n * n_cols * sizeof(u64));
    unsigned char *d_packed = (unsigned char *)c->alloc(total);
    u64 *d_off = (u64 *)c->alloc((size_t)n_cols * sizeof(u64));
    if (!d || !d_packed || !d_off) { c->release(d); c->release(d_packed); c->release(d_off); return c->fail(ZK_ERR_OOM, "upload_packed: device memory"); }
    int32_t rc = c->upload_large(d_packed, packed, total, c->stream);
    // The offset table does NOT go through the per-proof staging buffer (h_stage): this function returns before its copies have
    // run, and the next zk_prove_tables / zk_prove_plonk of the context resets h_stage_used and writes its own header words over
    // the same staging bytes while the copy of d_off is still queued behind the large upload.  It has a pinned buffer
    // of its own, reused only after the event behind the previous call's copy has completed.
    if (rc == ZK_OK) {
        const size_t ob = (size_t)n_cols * sizeof(u64);
        if (c->h_packed_off_cap < ob) {
            if (c->h_packed_off) { hipStreamSynchronize(c->stream); hipHostFree(c->h_packed_off); c->h_packed_off = nullptr; c->h_packed_off_cap = 0; }
            size_t cap = (ob + 4095) & ~(size_t)4095;
            if (hipHostMalloc((void **)&c->h_packed_off, cap, hipHostMallocDefault) != hipSuccess) { c->h_packed_off = nullptr; rc = c->fail(ZK_ERR_OOM, "upload_packed: pinned offset table"); }
            else c->h_packed_off_cap = cap;
        }
        if (rc == ZK_OK && !c->packed_off_done && hipEventCreateWithFlags(&c->packed_off_done, hipEventDisableTiming) != hipSuccess) { c->packed_off_done = nullptr; rc = c->fail(ZK_ERR_HIP, "upload_packed: event"); }
        if (rc == ZK_OK && hipEventSynchronize(c->packed_off_done) != hipSuccess) rc = c->fail(ZK_ERR_HIP, "upload_packed: event wait");   // never recorded: returns at once
        if (rc == ZK_OK) {
            memcpy(c->h_packed_off, off.data(), ob);
            if (hipMemcpyAsync(d_off, c->h_packed_off, ob, hipMemcpyHostToDevice, c->stream) != hipSuccess || hipEventRecord(c->packed_off_done, c->stream) != hipSuccess)
                rc = c->fail(ZK_ERR_HIP, "upload_packed: offset table copy");
        }
    }
    if (rc == ZK_OK) ZK_LAUNCH(c, "trace_widen", trace_widen_kernel, dim3((unsigned)((n + 255) / 256), n_cols), dim3(256), 0, (const unsigned char *)d_packed, (const u64 *)d_off, n, d);
    c->release(d_packed); c->release(d_off);                 // stream-ordered
    if (rc != ZK_OK) { c->release(d); return rc; }
    *d_ptr = d;
    return ZK_OK;
}
int32_t zk_dev_upload(zk_ctx *c, void *d_dst, const void *src, size_t bytes) {
    if (!c || (!d_dst && bytes) || (!src && bytes)) return ZK_ERR_INVALID_ARG;
    hipSetDevice(c->device);
    return c->copy_h2d_sync(d_dst, src, bytes);
}
int32_t zk_dev_download(zk_ctx *c, void *dst, const void *d_src, size_t bytes) {
    if (!c || (!dst && bytes) || (!d_src && bytes)) return ZK_ERR_INVALID_ARG;
    hipSetDevice(c->device);
    return c->copy_d2h_sync(dst, d_src, bytes);
}

void zk_stark_config_standard_fast(zk_stark_config *c) {
    if (!c) return;
    c->security_bits = 100; c->num_challenges = 2; c->rate_bits = 1; c->cap_height = 4;
    c->proof_of_work_bits = 16; c->fri_arity_bits = 4; c->fri_final_poly_bits = 5; c->num_query_rounds = 84;
}
void zk_stark_config_standard_recursion(zk_stark_config *c) {
    if (!c) return;
    c->security_bits = 100; c->num_challenges = 2; c->rate_bits = 3; c->cap_height = 4;
    c->proof_of_work_bits = 16; c->fri_arity_bits = 4; c->fri_final_poly_bits = 5; c->num_query_rounds = 28;
}

}  // extern "C"
