// zk_prove_tables: the whole STARK proof of one unit of work, enqueued on one HIP stream with a single host
// synchronisation at the end.
// Mirrors evm_arithmetization 0.1.3 prover::prove_with_traces (commit every trace, observe every cap, draw the CTL
// challenges, then prove the tables one after another on the shared challenger) driving starky 0.4.0
// prover::prove_with_commitment (crates pinned at /root/reference/Cargo.lock:1675-1677 and :4529-4531; transcript
// order: SURVEY.md A7).  This is the work behind the reference's single hot call
//   generate_txn_proof(p_state(), input, None)       /root/reference/ops/src/lib.rs:52
// up to (not including) the Plonk recursion wrappers.  Witness generation (generate_traces, a CPU EVM
// interpreter upstream) is out of scope: traces are an input.
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <mutex>
#include "zk_internal.h"
#include "prover.h"
#include <memory>

__global__ void zeta_kernel(u64 *zeta /* in: zeta; out: +2 = g_n * zeta */, u32 log_n, u64 g_n, u64 *err) {
    if (threadIdx.x || blockIdx.x) return;
    gl2 z = gl2_make(zeta[0], zeta[1]);
    gl2 p = z;
    for (u32 i = 0; i < log_n; i++) p = gl2_mul(p, p);
    if (p.a == 1 && p.b == 0) atomicCAS((unsigned long long *)err, 0ULL, (unsigned long long)(uint32_t)ZK_ERR_ZETA_IN_SUBGROUP);
    zeta[2] = gl_mul(z.a, g_n); zeta[3] = gl_mul(z.b, g_n);
    zeta[4] = 1; zeta[5] = 0;
}
__global__ void take_re_kernel(const u64 *ext, u32 n, u64 *out) { u32 i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) out[i] = ext[2 * i]; }
__global__ void fold_err_word_kernel(u64 *dst, u64 *src) { const u64 e = *src; if (e) { if (!*dst) *dst = e; *src = 0; } }
__global__ void copy_words2_kernel(const u64 *src, u64 *dst, u32 n) { u32 i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) dst[i] = src[i]; }

// Debugging aid (env ZK_DEBUG_SCRATCH_PROBE=1, tools/stress_faults.py): a kernel that deliberately lives in scratch (a dynamically
// indexed private array), launched between the real kernels of every proof on every stream -- the situation in which round 1 saw
// GPU memory faults with spilling kernels.  Writes a checksum nobody reads.
__global__ __launch_bounds__(256) void scratch_probe_kernel(u64 *out, u32 rounds) {
    u64 a[192];
    const u32 tid = blockIdx.x * blockDim.x + threadIdx.x;
    for (u32 i = 0; i < 192; i++) a[i] = (u64)tid * 0x9E3779B97F4A7C15ULL + i;
    u32 j = tid % 191;
    u64 acc = 0;
    for (u32 r = 0; r < rounds; r++) { acc += a[j]; a[(j * 7 + r) % 192] ^= acc; j = (u32)(acc % 191); }
    if (acc == 0x123456789ULL) out[0] = acc;         // never true in practice; keeps the loop alive
}
static void scratch_probe(zk_ctx *ctx, u64 *d_sink) {
    static const int on = getenv("ZK_DEBUG_SCRATCH_PROBE") ? atoi(getenv("ZK_DEBUG_SCRATCH_PROBE")) : 0;
    if (on) ZK_LAUNCH(ctx, "scratch_probe", scratch_probe_kernel, dim3(2048), dim3(256), 0, d_sink, 64u);
}

struct TableLayout { size_t base, o_trace_cap, o_aux_cap, o_quot_cap; FriLayout fri; size_t end; };

static u32 n_aux_cols(const zk_air *air, u32 nch) { return zk_air_n_aux(air, nch); }

static TableLayout table_layout(const zk_stark_config &cfg, const zk_table &t, size_t base) {
    TableLayout y = {};
    const u32 nch = cfg.num_challenges, capw = 4u << cfg.cap_height;
    const u32 C = t.air->n_trace_cols, A = n_aux_cols(t.air, nch), Q = nch << cfg.rate_bits;
    const u32 lgL = t.log_n + cfg.rate_bits;
    size_t o = base;
    y.base = o; o += 3;
    y.o_trace_cap = o; o += capw;
    if (A) { y.o_aux_cap = o; o += capw; }
    y.o_quot_cap = o; o += capw;
    y.fri.o_local = o; o += 2 * (size_t)C;
    y.fri.o_next = o; o += 2 * (size_t)C;
    y.fri.o_aux = o; o += 2 * (size_t)A;
    y.fri.o_auxn = o; o += 2 * (size_t)A;
    y.fri.o_zs = o; if (A) o += nch;
    y.fri.o_q = o; o += 2 * (size_t)Q;
    u32 ar[64], nl; zk_fri_arities(cfg, t.log_n, ar, &nl);
    y.fri.o_nl = o; o += 1;
    y.fri.o_layer_caps = o; o += (size_t)nl * capw;
    u32 d = t.log_n; for (u32 l = 0; l < nl; l++) d -= ar[l];
    y.fri.o_flen = o; o += 1;
    y.fri.o_final = o; o += 2 * ((size_t)1 << d);
    y.fri.o_witness = o; o += 1;
    size_t per_q = (size_t)(C + A + Q) + (size_t)(A ? 3 : 2) * 4 * (lgL - cfg.cap_height);
    u32 lg = lgL;
    for (u32 l = 0; l < nl; l++) { lg -= ar[l]; per_q += (2u << ar[l]) + 4 * (size_t)(lg - cfg.cap_height); }
    y.fri.per_query = per_q;
    y.fri.o_queries = o; o += per_q * cfg.num_query_rounds;
    y.end = o;
    return y;
}

extern "C" size_t zk_proof_size_words(const zk_stark_config *cfg, uint32_t n_tables, const zk_table *tables) {
    if (!cfg || !tables) return 0;
    for (u32 t = 0; t < n_tables; t++)
        if (!tables[t].air || tables[t].log_n + cfg->rate_bits < cfg->cap_height || !zk_fri_reduction_ok(*cfg, tables[t].log_n)) return 0;
    size_t o = 2;
    for (u32 t = 0; t < n_tables; t++) o = table_layout(*cfg, tables[t], o).end;
    return o;
}

static inline dim3 g1(size_t n) { return dim3((unsigned)((n + 255) / 256)); }

static int32_t prove_tables_impl(zk_ctx *ctx, const zk_stark_config *cfgp, uint32_t nt, const zk_table *tables,
                                 uint64_t *proof_out, size_t cap_words, size_t *len_words, zk_table_ready_fn on_table, void *user);

// Round 1 serialised the first proof of a process behind a mutex ("cold start") after sporadic GPU memory faults in bench.py.  The
// fault hunt of round 2 (tools/stress_faults.py: 8 streams x 100 full-size proofs x {build-time, hiprtc, interpreter} kernels x
// {plain, fence allocator, a scratch-resident kernel between the real ones, mutex off}, each in a fresh process:
// profiles/r02g_stress_faults.jsonl) ran 9600 proofs without a fault or a differing proof word, with the mutex off as well, so
// the serialisation is off by default; ZK_COLD_MUTEX=1 brings it back.  That hunt prepared the kernels on one thread first: the one
// scenario it did not cover -- several threads compiling and loading code objects at their FIRST proofs (ZK_QUOTIENT_JIT=1 / 2: hiprtc at
// first use) -- keeps the mutex unless ZK_COLD_MUTEX=0 says otherwise.
extern "C" int32_t zk_prove_tables_notify(zk_ctx *ctx, const zk_stark_config *cfgp, uint32_t nt, const zk_table *tables,
                                          uint64_t *proof_out, size_t cap_words, size_t *len_words, zk_table_ready_fn on_table, void *user) {
    static std::atomic<bool> warm{false};
    static std::mutex cold_mu;
    static const bool cold_env = getenv("ZK_COLD_MUTEX") != nullptr, cold_on = cold_env && atoi(getenv("ZK_COLD_MUTEX"));
    const bool cold = cold_env ? cold_on : (ctx && ctx->quotient_jit_mode >= 1);
    if (!cold || warm.load(std::memory_order_acquire)) return prove_tables_impl(ctx, cfgp, nt, tables, proof_out, cap_words, len_words, on_table, user);
    std::lock_guard<std::mutex> lk(cold_mu);
    int32_t rc = prove_tables_impl(ctx, cfgp, nt, tables, proof_out, cap_words, len_words, on_table, user);
    if (rc == ZK_OK) warm.store(true, std::memory_order_release);
    return rc;
}
extern "C" int32_t zk_prove_tables(zk_ctx *ctx, const zk_stark_config *cfgp, uint32_t nt, const zk_table *tables,
                                   uint64_t *proof_out, size_t cap_words, size_t *len_words) {
    return zk_prove_tables_notify(ctx, cfgp, nt, tables, proof_out, cap_words, len_words, nullptr, nullptr);
}

static int32_t prove_tables_impl(zk_ctx *ctx, const zk_stark_config *cfgp, uint32_t nt, const zk_table *tables,
                                 uint64_t *proof_out, size_t cap_words, size_t *len_words, zk_table_ready_fn on_table, void *user) {
    if (!ctx || !cfgp || !tables || nt == 0 || !proof_out) return ZK_ERR_INVALID_ARG;
    const zk_stark_config &cfg = *cfgp;
    if (cfg.num_challenges == 0 || cfg.num_challenges > 4 || cfg.rate_bits != 1 || cfg.cap_height > 8 || cfg.fri_arity_bits < 2 || cfg.fri_arity_bits > 6)
        return ctx->fail(ZK_ERR_INVALID_ARG, "prove: unsupported StarkConfig");
    for (u32 t = 0; t < nt; t++) {
        if (!tables[t].air || !tables[t].trace || tables[t].log_n > 28 || tables[t].log_n + cfg.rate_bits < cfg.cap_height)
            return ctx->fail(ZK_ERR_INVALID_ARG, "prove: bad table");
        if (!zk_fri_reduction_ok(cfg, tables[t].log_n)) return ctx->fail(ZK_ERR_INVALID_ARG, "prove: the FRI reduction strategy does not fit this table size (upstream panics)");
        if (zk_air_validate(tables[t].air, cfg.num_challenges) != ZK_OK) return ctx->fail(ZK_ERR_INVALID_ARG, "prove: malformed AIR (zk_air_validate)");
        if (tables[t].air->n_public_inputs && !tables[t].public_inputs) return ctx->fail(ZK_ERR_INVALID_ARG, "prove: public inputs missing");
    }
    hipSetDevice(ctx->device);
    for (u32 t = 0; t < nt; t++) if (tables[t].trace_on_device) ctx->wait_upload(tables[t].trace);   // zk_trace_upload_async prefetches (device-side wait)
    const u32 nch = cfg.num_challenges, capw = 4u << cfg.cap_height;
    std::vector<TableLayout> lay(nt);
    size_t total = 2;
    for (u32 t = 0; t < nt; t++) { lay[t] = table_layout(cfg, tables[t], total); total = lay[t].end; }
    if (len_words) *len_words = total;
    if (cap_words < total) return ctx->fail(ZK_ERR_BUFFER_TOO_SMALL, "prove: proof buffer too small");

    int32_t rc = ZK_OK;
    std::vector<hipEvent_t> table_done;
    // Everything this proof allocates is handed back on EVERY exit path, and only after the stream has drained: an early return
    // must not let the next proof of this context reuse pinned staging or scratch that an enqueued copy or kernel still reads.
    struct Guard {
        zk_ctx *ctx; std::vector<void *> owned; std::vector<zk_batch *> trace_b, aux_b, quot_b;
        ~Guard() {
            hipStreamSynchronize(ctx->stream);
            for (auto *b : trace_b) if (b) zk_batch_free(b);
            for (auto *b : aux_b) if (b) zk_batch_free(b);
            for (auto *b : quot_b) if (b) zk_batch_free(b);
            for (void *q : owned) ctx->release(q);
        }
    } guard{ctx, {}, std::vector<zk_batch *>(nt, nullptr), std::vector<zk_batch *>(nt, nullptr), std::vector<zk_batch *>(nt, nullptr)};
    std::vector<void *> &owned = guard.owned;
    auto A_ = [&](size_t bytes) { void *p = ctx->alloc(bytes); if (p) owned.push_back(p); return p; };
    u64 *d_proof = (u64 *)A_(total * sizeof(u64));
    DevChallenger *ch = (DevChallenger *)A_(sizeof(DevChallenger));
    u64 *d_chal = (u64 *)A_(16 * sizeof(u64)), *d_alphas = (u64 *)A_(8 * sizeof(u64)), *d_zeta = (u64 *)A_(8 * sizeof(u64));
    if (!d_proof || !ch || !d_chal || !d_alphas || !d_zeta) return ctx->fail(ZK_ERR_OOM, "prove: out of device memory");

    // small host objects go up through the context's pinned staging (zk_ctx::upload_small); the proof comes back through
    // its pinned download buffer.  The staging is reused from the start: the previous proof of this context has been waited for.
    ctx->h_stage_used = 0;
    zk_host_transcript_begin_proof(ctx);
    u64 *h_out = (u64 *)ctx->pinned_out((total + 1) * sizeof(u64));
    if (!h_out) return ctx->fail(ZK_ERR_OOM, "prove: pinned download buffer");

    // header words of the proof (magic, n_tables, per table log_n / C / A)
    {
        u64 h2[2] = {0x31465250474C4B5AULL, nt};
        if ((rc = ctx->upload_small(d_proof, h2, sizeof h2)) != ZK_OK) return rc;
        for (u32 t = 0; t < nt; t++) {
            u64 h3[3] = {tables[t].log_n, tables[t].air->n_trace_cols, n_aux_cols(tables[t].air, nch)};
            if ((rc = ctx->upload_small(d_proof + lay[t].base, h3, sizeof h3)) != ZK_OK) return rc;
        }
    }
    // AIRs and public inputs to the device
    std::vector<DevAir> dair(nt);
    std::vector<u64 *> d_pub(nt, nullptr);
    for (u32 t = 0; t < nt; t++) {
        const zk_air *a = tables[t].air;
        dair[t].host = a;
        dair[t].d_instrs = (zk_instr *)A_((size_t)(a->n_instrs ? a->n_instrs : 1) * sizeof(zk_instr));
        dair[t].d_constants = (u64 *)A_((size_t)(a->n_constants ? a->n_constants : 1) * sizeof(u64));
        dair[t].d_helpers = (zk_helper_desc *)A_((size_t)(a->n_helpers ? a->n_helpers : 1) * sizeof(zk_helper_desc));
        d_pub[t] = (u64 *)A_((size_t)(a->n_public_inputs ? a->n_public_inputs : 1) * sizeof(u64));
        if (!dair[t].d_instrs || !dair[t].d_constants || !dair[t].d_helpers || !d_pub[t]) return ZK_ERR_OOM;
        if (a->n_instrs && (rc = ctx->upload_small(dair[t].d_instrs, a->instrs, a->n_instrs * sizeof(zk_instr))) != ZK_OK) return rc;
        if (a->n_constants) {
            std::vector<u64> c(a->constants, a->constants + a->n_constants);
            for (auto &v : c) v = gl_canon(v);
            if ((rc = ctx->upload_small(dair[t].d_constants, c.data(), c.size() * sizeof(u64))) != ZK_OK) return rc;
        }
        if (a->n_helpers && (rc = ctx->upload_small(dair[t].d_helpers, a->helpers, a->n_helpers * sizeof(zk_helper_desc))) != ZK_OK) return rc;
        dair[t].d_helpers_ext = nullptr;
        if (a->n_helpers && a->helpers_ext) {
            dair[t].d_helpers_ext = (zk_helper_ext *)A_((size_t)a->n_helpers * sizeof(zk_helper_ext));
            if (!dair[t].d_helpers_ext) return ZK_ERR_OOM;
            if ((rc = ctx->upload_small(dair[t].d_helpers_ext, a->helpers_ext, a->n_helpers * sizeof(zk_helper_ext))) != ZK_OK) return rc;
        }
        if (a->n_public_inputs) {
            std::vector<u64> c(tables[t].public_inputs, tables[t].public_inputs + a->n_public_inputs);
            for (auto &v : c) v = gl_canon(v);
            if ((rc = ctx->upload_small(d_pub[t], c.data(), c.size() * sizeof(u64))) != ZK_OK) return rc;
        }
    }

    std::vector<zk_batch *> &trace_b = guard.trace_b, &aux_b = guard.aux_b, &quot_b = guard.quot_b;
    std::vector<const u64 *> d_trace(nt, nullptr);

    // ---- commit every trace; observe every trace cap, then the public values; draw the lookup/CTL challenges
    // (phase scopes carry upstream's `timed!` names -- the TimingTree the reference creates at ops/src/lib.rs:52 -- next to the
    // per-kernel labels of the profiler: "phase: ..." entries of zk_profile_get)
    { TOpList a = {}; a.n = 1; a.op[0].kind = TOP_INIT; zk_transcript_run(ctx, ch, a); }
    std::unique_ptr<ProfScope> phase(new ProfScope(ctx, "phase: compute all trace commitments"));
    // zk_ctx_set_side_commit: which table (the one with the most cells) commits on the side context
    u32 side_t = 0xFFFFFFFFu;
    hipEvent_t side_ev[2] = {nullptr, nullptr};
    if (ctx->side_commit && nt >= 2) {
        if (!ctx->side && zk_ctx_create(ctx->device, nullptr, &ctx->side) != ZK_OK) ctx->side = nullptr;
        if (ctx->side && hipEventCreateWithFlags(&side_ev[0], hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&side_ev[1], hipEventDisableTiming) == hipSuccess) {
            size_t best = 0;
            for (u32 t = 0; t < nt; t++) {
                const size_t cells = (size_t)tables[t].air->n_trace_cols << tables[t].log_n;
                if (cells > best) { best = cells; side_t = t; }
            }
        }
    }
    for (u32 t = 0; t < nt && rc == ZK_OK; t++) {
        const size_t n = (size_t)1 << tables[t].log_n;
        const u32 C = tables[t].air->n_trace_cols;
        if (tables[t].trace_on_device) d_trace[t] = (const u64 *)tables[t].trace;
        else {
            u64 *d = (u64 *)A_(n * C * sizeof(u64));
            if (!d) { rc = ZK_ERR_OOM; break; }
            // host-resident trace: pinned memory (zk_host_alloc) goes up with one asynchronous copy on this stream, pageable
            // memory through the context's two pinned chunks; no host wait either way (zk_ctx::upload_large)
            if ((rc = ctx->upload_large(d, tables[t].trace, n * C * sizeof(u64), ctx->stream)) != ZK_OK) break;
            d_trace[t] = d;
        }
        if (t == side_t) {
            // the widest table on the side context: its stream starts behind what this stream has enqueued so far (the trace itself),
            // its commitment joins this stream again where its cap is observed
            if (hipEventRecord(side_ev[0], ctx->stream) != hipSuccess || hipStreamWaitEvent(ctx->side->stream, side_ev[0], 0) != hipSuccess) { rc = ctx->fail(ZK_ERR_HIP, "prove: side stream"); break; }
            ctx->side->prof = ctx->prof;          // its launches are part of this proof's profile (zk_profile_count folds them in)
            rc = zk_batch_commit_dev(ctx->side, d_trace[t], tables[t].log_n, C, cfg.rate_bits, cfg.cap_height, 1, &trace_b[t]);
            if (rc == ZK_OK && hipEventRecord(side_ev[1], ctx->side->stream) != hipSuccess) rc = ctx->fail(ZK_ERR_HIP, "prove: side stream");
        } else rc = zk_batch_commit_dev(ctx, d_trace[t], tables[t].log_n, C, cfg.rate_bits, cfg.cap_height, 1, &trace_b[t]);
        if (rc) break;
    }
    for (u32 t = 0; t < nt && rc == ZK_OK; t++) {
        if (t == side_t) {
            if (hipStreamWaitEvent(ctx->stream, side_ev[1], 0) != hipSuccess) { rc = ctx->fail(ZK_ERR_HIP, "prove: side stream"); break; }
            // device-side checks raised by the side context's kernels land in ITS error word: fold it into this proof's (first error
            // wins) and clear it, behind the join
            ctx->note_launch("fold_err_word", 1, 1);
            hipLaunchKernelGGL(fold_err_word_kernel, dim3(1), dim3(1), 0, ctx->stream, ctx->d_err, ctx->side->d_err);
        }
        scratch_probe(ctx, d_zeta + 7);
        ZK_LAUNCH(ctx, "copy_words", copy_words2_kernel, g1(capw), dim3(256), 0, trace_b[t]->cap(), d_proof + lay[t].o_trace_cap, capw);
        TOpList a = {}; a.n = 1; a.op[0].kind = TOP_OBSERVE; a.op[0].src = d_proof + lay[t].o_trace_cap; a.op[0].n = capw;
        zk_transcript_run(ctx, ch, a);
    }
    phase.reset();
    for (int i = 0; i < 2; i++) if (side_ev[i]) hipEventDestroy(side_ev[i]);
    if (rc) { if (side_t != 0xFFFFFFFFu) hipStreamSynchronize(ctx->side->stream); return rc; }
    for (u32 t = 0; t < nt; t++) if (tables[t].air->n_public_inputs) {
        TOpList a = {}; a.n = 1; a.op[0].kind = TOP_OBSERVE; a.op[0].src = d_pub[t]; a.op[0].n = tables[t].air->n_public_inputs;
        zk_transcript_run(ctx, ch, a);
    }
    { TOpList a = {}; a.n = 1; a.op[0].kind = TOP_GET; a.op[0].dst = d_chal; a.op[0].n = 2 * nch; zk_transcript_run(ctx, ch, a); }

    // ---- the tables, one after another on the shared challenger
    for (u32 t = 0; t < nt && rc == ZK_OK; t++) {
        const zk_air *air = tables[t].air;
        const u32 log_n = tables[t].log_n, C = air->n_trace_cols, A = n_aux_cols(air, nch), Q = nch << cfg.rate_bits;
        const size_t n = (size_t)1 << log_n, L = n << cfg.rate_bits;
        const TableLayout &y = lay[t];
        { TOpList a = {}; a.n = 1; a.op[0].kind = TOP_COMPACT; zk_transcript_run(ctx, ch, a); }     // init_challenger_state
        if (A) {
            ProfScope ph(ctx, "phase: compute auxiliary polynomials commitment");
            u64 *d_aux = (u64 *)ctx->alloc((size_t)A * n * sizeof(u64));
            if (!d_aux) { rc = ZK_ERR_OOM; break; }
            rc = zk_aux_columns_dev(ctx, air, dair[t].d_helpers, dair[t].d_helpers_ext, nch, d_chal, d_trace[t], log_n, d_aux);
            if (rc == ZK_OK) rc = zk_batch_commit_dev(ctx, d_aux, log_n, A, cfg.rate_bits, cfg.cap_height, 1, &aux_b[t]);
            ctx->release(d_aux);
            if (rc) break;
            ZK_LAUNCH(ctx, "copy_words", copy_words2_kernel, g1(capw), dim3(256), 0, aux_b[t]->cap(), d_proof + y.o_aux_cap, capw);
            TOpList a = {}; a.n = 1; a.op[0].kind = TOP_OBSERVE; a.op[0].src = d_proof + y.o_aux_cap; a.op[0].n = capw;
            zk_transcript_run(ctx, ch, a);
        }
        { TOpList a = {}; a.n = 1; a.op[0].kind = TOP_GET; a.op[0].dst = d_alphas; a.op[0].n = nch; zk_transcript_run(ctx, ch, a); }
        u64 *d_q = (u64 *)ctx->alloc((size_t)nch * L * sizeof(u64));
        if (!d_q) { rc = ZK_ERR_OOM; break; }
        {
            ProfScope ph(ctx, "phase: compute quotient polys");
            rc = zk_quotient_dev(ctx, dair[t], cfg, trace_b[t], aux_b[t], d_pub[t], d_chal, d_alphas, d_q);
        }
        // the nch*L natural-order coefficients are Q = nch << rate_bits polynomials of n coefficients, back to back
        if (rc == ZK_OK) { ProfScope ph(ctx, "phase: compute quotient commitment"); rc = zk_batch_commit_dev(ctx, d_q, log_n, Q, cfg.rate_bits, cfg.cap_height, 0, &quot_b[t]); }
        ctx->release(d_q);
        if (rc) break;
        scratch_probe(ctx, d_zeta + 7);
        ZK_LAUNCH(ctx, "copy_words", copy_words2_kernel, g1(capw), dim3(256), 0, quot_b[t]->cap(), d_proof + y.o_quot_cap, capw);
        { TOpList a = {}; a.n = 2; a.op[0].kind = TOP_OBSERVE; a.op[0].src = d_proof + y.o_quot_cap; a.op[0].n = capw;
          a.op[1].kind = TOP_GET; a.op[1].dst = d_zeta; a.op[1].n = 2; zk_transcript_run(ctx, ch, a); }
        ZK_LAUNCH(ctx, "zeta", zeta_kernel, dim3(1), dim3(1), 0, d_zeta, log_n, gl_root(log_n), ctx->d_err);

        // ---- StarkOpeningSet::new: every coefficient polynomial at zeta, trace and aux also at g*zeta, Z columns at 1
        std::unique_ptr<ProfScope> ph_open(new ProfScope(ctx, "phase: construct the opening set"));
        u64 *zs = nullptr;
        {
            zk_eval_job jobs[6]; u32 nj = 0;
            jobs[nj++] = {trace_b[t]->coeffs, n, C, d_zeta, d_proof + y.fri.o_local};
            jobs[nj++] = {trace_b[t]->coeffs, n, C, d_zeta + 2, d_proof + y.fri.o_next};
            if (A) {
                jobs[nj++] = {aux_b[t]->coeffs, n, A, d_zeta, d_proof + y.fri.o_aux};
                jobs[nj++] = {aux_b[t]->coeffs, n, A, d_zeta + 2, d_proof + y.fri.o_auxn};
            }
            jobs[nj++] = {quot_b[t]->coeffs, n, Q, d_zeta, d_proof + y.fri.o_q};
            if (A) {
                zs = (u64 *)ctx->alloc(2 * nch * sizeof(u64));
                if (!zs) { rc = ZK_ERR_OOM; break; }
                jobs[nj++] = {aux_b[t]->coeffs + (size_t)(A - nch) * n, n, nch, d_zeta + 4, zs};
            }
            rc = zk_eval_cols_multi(ctx, (int)log_n, nj, jobs);
        }
        if (zs) {
            if (rc == ZK_OK) ZK_LAUNCH(ctx, "take_re", take_re_kernel, dim3(1), dim3(64), 0, (const u64 *)zs, nch, d_proof + y.fri.o_zs);
            ctx->release(zs);
        }
        ph_open.reset();
        if (rc) break;
        // challenger.observe_openings(&openings.to_fri_openings()): zeta batch (local | aux | quotient), g*zeta batch, then Z(1)
        {
            TOpList a = {}; int k = 0;
            a.op[k].kind = TOP_OBSERVE; a.op[k].src = d_proof + y.fri.o_local; a.op[k].n = 2 * C; k++;
            if (A) { a.op[k].kind = TOP_OBSERVE; a.op[k].src = d_proof + y.fri.o_aux; a.op[k].n = 2 * A; k++; }
            a.op[k].kind = TOP_OBSERVE; a.op[k].src = d_proof + y.fri.o_q; a.op[k].n = 2 * Q; k++;
            a.op[k].kind = TOP_OBSERVE; a.op[k].src = d_proof + y.fri.o_next; a.op[k].n = 2 * C; k++;
            if (A) { a.op[k].kind = TOP_OBSERVE; a.op[k].src = d_proof + y.fri.o_auxn; a.op[k].n = 2 * A; k++; }
            if (A) { a.op[k].kind = TOP_OBSERVE_BASE_AS_EXT; a.op[k].src = d_proof + y.fri.o_zs; a.op[k].n = nch; k++; }
            a.n = k;
            zk_transcript_run(ctx, ch, a);
        }
        { ProfScope ph(ctx, "phase: compute openings proof"); rc = zk_fri_prove_dev(ctx, cfg, ch, trace_b[t], aux_b[t], quot_b[t], d_zeta, d_proof, y.fri); }
        // this table's StarkProof is final: with a listener, its words go to the host now (the later tables are proved meanwhile)
        if (on_table && rc == ZK_OK) {
            hipEvent_t e = nullptr;
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess ||
                hipMemcpyAsync(h_out + lay[t].base, d_proof + lay[t].base, (lay[t].end - lay[t].base) * sizeof(u64), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                hipEventRecord(e, ctx->stream) != hipSuccess) { if (e) hipEventDestroy(e); rc = ctx->fail(ZK_ERR_HIP, "prove: per-table hand-off"); break; }
            table_done.push_back(e);
        }
    }
    // everything is enqueued (the host runs tens of milliseconds ahead of the GPU here): hand the tables over as they finish
    for (u32 t = 0; t < (u32)table_done.size(); t++) {
        if (rc == ZK_OK && hipEventSynchronize(table_done[t]) != hipSuccess) rc = ctx->fail(ZK_ERR_HIP, "prove: per-table hand-off");
        if (rc == ZK_OK) on_table(user, t, h_out + lay[t].base, lay[t].end - lay[t].base);
        hipEventDestroy(table_done[t]);
    }

    // ---- the only host synchronisation of the proof: fetch the proof bytes and the device error word
    u64 err_word = 0;
    if (rc == ZK_OK) {
        if (hipMemcpyAsync(h_out, d_proof, total * sizeof(u64), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) rc = ZK_ERR_HIP;
        if (hipMemcpyAsync(h_out + total, ctx->d_err, sizeof(u64), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) rc = ZK_ERR_HIP;
    }
    hipError_t se = ctx->wait_stream();
    if (se != hipSuccess && rc == ZK_OK) rc = ctx->fail(ZK_ERR_HIP, std::string("stream sync: ") + hipGetErrorString(se));
    if (rc == ZK_OK) { memcpy(proof_out, h_out, total * sizeof(u64)); err_word = h_out[total]; }
    if (rc == ZK_OK && err_word) {
        rc = (int32_t)(uint32_t)err_word;
        ctx->fail(rc, "prove: device-side check failed");
        hipMemsetAsync(ctx->d_err, 0, sizeof(u64), ctx->stream);
    }
    if (const int32_t he = zk_host_transcript_take_error(ctx)) { if (rc == ZK_OK) { rc = he; ctx->fail(rc, "prove: host-side transcript check failed"); } }
    return rc;
}
