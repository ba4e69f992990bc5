// Constraint programs compiled to gfx950 code.
// Upstream, a table's constraints are Rust code (`Stark::eval_packed_generic`, starky 0.4.0 / evm_arithmetization 0.1.3,
// pinned at /root/reference/Cargo.lock:4529-4531, 1675-1677) monomorphised when the prover is built; the reference builds
// its prover state once at start-up (/root/reference/common/src/prover_state/mod.rs:59-95).  Here the constraints arrive
// as data (zk_air), and the counterpart of that build step is: turn the straight-line program into HIP source, compile it
// with hiprtc for gfx950, keep the module.  The generated kernel does exactly what the bytecode interpreter in
// prover_quotient.hip does -- same field operations in the same order, so the same bits -- without the per-instruction
// fetch / decode / LDS round trip: temporaries live in VGPRs, constants are immediates, every column is loaded once.
// Small tables (and ZK_QUOTIENT_JIT=0) stay on the interpreter.
#include <hip/hiprtc.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>
#include "zk_internal.h"
#include "prover.h"

static const char GL64_SRC[] =
#include "gl64_src.inc"
    ;

namespace {
// `image` = the code object handed to hipModuleLoadData.  It is kept for the life of the process: the runtime may keep reading
// the caller's buffer after the call returns (deferred / asynchronous code-object loading); when the vector was freed at the
// end of the compile block -- in a worker thread's heap arena, which glibc gives back to the OS -- first proofs ended in sporadic
// "memory access fault by GPU" at host heap addresses (bench.py, one thread per stream).
struct JitKernel { hipModule_t mod = nullptr; hipFunction_t fn = nullptr; std::shared_ptr<std::vector<char>> image; };
std::mutex g_mu;
std::map<std::string, JitKernel> g_cache;

void appendf(std::string &s, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
void appendf(std::string &s, const char *fmt, ...) {
    char buf[512];
    va_list ap; va_start(ap, fmt); int n = vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    s.append(buf, (size_t)(n < (int)sizeof buf ? n : (int)sizeof buf - 1));
}

// C++ expression for an operand; emits the (single) load of a column / public input / challenge at its first use
struct Gen {
    const zk_air *air; u32 C, A;
    std::string body;
    std::vector<char> have_l, have_n, have_pub, have_ch;
    std::string opnd(u32 o) {
        u32 kind = o >> 24, idx = o & 0xFFFFFF;
        char name[64];
        switch (kind) {
        case ZK_OPND_REG: snprintf(name, sizeof name, "r%u", idx); return name;
        case ZK_OPND_LOCAL:
        case ZK_OPND_NEXT: {
            const bool nx = kind == ZK_OPND_NEXT;
            std::vector<char> &have = nx ? have_n : have_l;
            snprintf(name, sizeof name, "%c%u", nx ? 'n' : 'l', idx);
            if (!have[idx]) {
                have[idx] = 1;
                if (idx < C) appendf(body, "    const u64 %s = p.trace_lde[%uull * p.L + %s];\n", name, idx, nx ? "sn" : "s");
                else appendf(body, "    const u64 %s = p.aux_lde[%uull * p.L + %s];\n", name, idx - C, nx ? "sn" : "s");
            }
            return name;
        }
        case ZK_OPND_CONST: snprintf(name, sizeof name, "0x%llxull", (unsigned long long)gl_canon(air->constants[idx])); return name;
        case ZK_OPND_PUBLIC:
            snprintf(name, sizeof name, "pub%u", idx);
            if (!have_pub[idx]) { have_pub[idx] = 1; appendf(body, "    const u64 %s = p.pub[%u];\n", name, idx); }
            return name;
        default:
            snprintf(name, sizeof name, "ch%u", idx);
            if (idx >= have_ch.size()) have_ch.resize(idx + 1, 0);
            if (!have_ch[idx]) { have_ch[idx] = 1; appendf(body, "    const u64 %s = p.chal[%u];\n", name, idx); }
            return name;
        }
    }
};

// Wide AIRs (thousands of instructions): ONE straight-line function makes hipcc's time grow with the square of its length (430 columns /
// 2 191 instructions: 79 s; 2 431 columns / 10 195: more than half an hour).  Above JIT_CHUNK_MIN instructions the program is cut into
// chunks of JIT_CHUNK instructions, each a separate non-inlined device function that takes the few values alive at a cut -- the
// program's registers (at most 24 by the bytecode's own rule), the accumulators, the row's selectors -- through memory and loads the
// columns it touches itself.  Compile time becomes linear (2 431 columns: about a minute), the run-time cost is a few dozen scratch
// accesses per chunk.
#define JIT_CHUNK_MIN 1536u
#define JIT_CHUNK 256u
static std::string generate_chunked(const zk_air *air, u32 C, u32 A, u32 nch, u32 rate_bits, const char *kname, bool standalone) {
    std::string s;
    if (standalone) {
        s += "#include \"gl64.cuh\"\n";
        s += "struct QJitParams { const u64 *trace_lde, *aux_lde, *pub, *chal, *alphas, *lag, *xA, *xB; u64 *out; u64 L; u64 last_inv; u64 zh_inv[8]; u32 lgL; u32 force; u32 *redo; u32 redo_cap; u32 pad; };\n";
    }
    const u32 NR = air->n_regs ? air->n_regs : 1;
    u32 n_chunks = 0;
    for (u32 pc0 = 0; pc0 < air->n_instrs; pc0 += JIT_CHUNK, n_chunks++) {
        Gen g; g.air = air; g.C = C; g.A = A;
        g.have_l.assign(C + A, 0); g.have_n.assign(C + A, 0); g.have_pub.assign(air->n_public_inputs + 1, 0);
        const u32 pc1 = pc0 + JIT_CHUNK < air->n_instrs ? pc0 + JIT_CHUNK : air->n_instrs;
        for (u32 pc = pc0; pc < pc1; pc++) {
            const zk_instr &in = air->instrs[pc];
            std::string a = g.opnd(in.a);
            if (in.op <= ZK_OP_MUL) {
                std::string b = g.opnd(in.b);
                appendf(g.body, "    r%u = %s(%s, %s);\n", in.dst & 0xFFFFFF, in.op == ZK_OP_ADD ? "gl_add" : in.op == ZK_OP_SUB ? "gl_sub" : "g.mul", a.c_str(), b.c_str());
            } else {
                const char *f = in.op == ZK_OP_EMIT_TRANSITION ? "z_last" : in.op == ZK_OP_EMIT_FIRST ? "l_first" : in.op == ZK_OP_EMIT_LAST ? "l_last" : nullptr;
                if (f) appendf(g.body, "    { const u64 c = g.mul(%s, %s);", a.c_str(), f);
                else appendf(g.body, "    { const u64 c = %s;", a.c_str());
                for (u32 j = 0; j < nch; j++) appendf(g.body, " acc%u = g.mul_add(acc%u, al%u, c);", j, j, j);
                g.body += " }\n";
            }
        }
        appendf(s, "template <bool EX> __device__ __noinline__ u64 %s_c%u(const u64 *__restrict__ trace_lde, const u64 *__restrict__ aux_lde, const u64 *__restrict__ pubv, "
                   "const u64 *__restrict__ chalv, u64 L_, u64 s, u64 sn, u64 z_last, u64 l_first, u64 l_last, const u64 *al_, u64 *r_, u64 *acc_) {\n", kname, n_chunks);
        s += "    glx<EX> g; g.fl = 0;\n";
        s += "    const struct { const u64 *trace_lde, *aux_lde, *pub, *chal; u64 L; } p = {trace_lde, aux_lde, pubv, chalv, L_};\n";
        for (u32 r = 0; r < NR; r++) appendf(s, "    u64 r%u = r_[%u];\n", r, r);
        for (u32 j = 0; j < nch; j++) appendf(s, "    u64 acc%u = acc_[%u]; const u64 al%u = al_[%u];\n", j, j, j, j);
        s += g.body;
        for (u32 r = 0; r < NR; r++) appendf(s, "    r_[%u] = r%u;\n", r, r);
        for (u32 j = 0; j < nch; j++) appendf(s, "    acc_[%u] = acc%u;\n", j, j);
        s += "    return g.fl;\n}\n";
    }
    appendf(s, "template <bool EX> __device__ __forceinline__ u64 %s_run(const QJitParams &p, u64 s, u32 i, u64 sn) {\n", kname);
    s += "    glx<EX> g; g.fl = 0;\n";
    s += "    const u64 x = g.mul(p.xA[i & 2047], p.xB[i >> 11]);\n    const u64 z_last = gl_sub(x, p.last_inv);\n";
    s += "    const u64 l_first = p.lag[s], l_last = p.lag[p.L + s];\n";
    appendf(s, "    u64 rr[%u], ac[%u], alv[%u];\n", NR, nch, nch);
    for (u32 r = 0; r < NR; r++) appendf(s, "    rr[%u] = 0;\n", r);
    for (u32 j = 0; j < nch; j++) appendf(s, "    ac[%u] = 0; alv[%u] = p.alphas[%u];\n", j, j, j);
    s += "    u64 fl = 0;\n";
    for (u32 k = 0; k < n_chunks; k++)
        appendf(s, "    fl |= %s_c%u<EX>(p.trace_lde, p.aux_lde, p.pub, p.chal, p.L, s, sn, z_last, l_first, l_last, alv, rr, ac);\n", kname, k);
    s += "    u64 zi = p.zh_inv[0];\n";
    appendf(s, "    { const u32 zsel = i & %uu;\n", (1u << rate_bits) - 1);
    for (u32 j = 1; j < (1u << rate_bits); j++) appendf(s, "      zi = zsel == %uu ? p.zh_inv[%u] : zi;\n", j, j);
    s += "    }\n";
    for (u32 j = 0; j < nch; j++) appendf(s, "    p.out[%uull * p.L + s] = g.mul(ac[%u], zi);\n", j, j);
    s += "    return g.fl | fl;\n}\n";
    appendf(s, "extern \"C\" __global__ __launch_bounds__(256) void %s(QJitParams p) {\n", kname);
    s += "    const u64 s = (u64)blockIdx.x * 256 + threadIdx.x;\n    if (s >= p.L) return;\n";
    s += "    const u32 i = bitrev_u32((u32)s, p.lgL);\n";
    appendf(s, "    const u64 sn = bitrev_u32((i + %uu) & (u32)(p.L - 1), p.lgL);\n", 1u << rate_bits);
    appendf(s, "    const u64 f = %s_run<false>(p, s, i, sn);\n", kname);
    appendf(s, "    if ((f != 0) | p.force) %s_run<true>(p, s, i, sn);\n}\n", kname);
    return s;
}

std::string generate(const zk_air *air, u32 C, u32 A, u32 nch, u32 rate_bits, const char *kname = "quotient_jit", bool standalone = true) {
    if (air->n_instrs >= JIT_CHUNK_MIN) return generate_chunked(air, C, A, nch, rate_bits, kname, standalone);
    Gen g; g.air = air; g.C = C; g.A = A;
    g.have_l.assign(C + A, 0); g.have_n.assign(C + A, 0); g.have_pub.assign(air->n_public_inputs + 1, 0);
    for (u32 pc = 0; pc < air->n_instrs; pc++) {
        const zk_instr &in = air->instrs[pc];
        std::string a = g.opnd(in.a);
        if (in.op <= ZK_OP_MUL) {
            std::string b = g.opnd(in.b);
            appendf(g.body, "    r%u = %s(%s, %s);\n", in.dst & 0xFFFFFF, in.op == ZK_OP_ADD ? "gl_add" : in.op == ZK_OP_SUB ? "gl_sub" : "g.mul",
                    a.c_str(), b.c_str());
        } else {
            const char *f = in.op == ZK_OP_EMIT_TRANSITION ? "z_last" : in.op == ZK_OP_EMIT_FIRST ? "l_first" : in.op == ZK_OP_EMIT_LAST ? "l_last" : nullptr;
            if (f) appendf(g.body, "    { const u64 c = g.mul(%s, %s);", a.c_str(), f);
            else appendf(g.body, "    { const u64 c = %s;", a.c_str());
            for (u32 j = 0; j < nch; j++) appendf(g.body, " acc%u = g.mul_add(acc%u, al%u, c);", j, j, j);
            g.body += " }\n";
        }
    }
    std::string s;
    if (standalone) {
        s += "#include \"gl64.cuh\"\n";
        s += "struct QJitParams { const u64 *trace_lde, *aux_lde, *pub, *chal, *alphas, *lag, *xA, *xB; u64 *out; u64 L; u64 last_inv; u64 zh_inv[8]; u32 lgL; u32 force; u32 *redo; u32 redo_cap; u32 pad; };\n";
    }
    appendf(s, "extern \"C\" __global__ __launch_bounds__(256) void %s(QJitParams p) {\n", kname);
    s += "    const u64 s = (u64)blockIdx.x * 256 + threadIdx.x;\n    if (s >= p.L) return;\n";
    s += "    const u32 i = bitrev_u32((u32)s, p.lgL);\n";
    appendf(s, "    const u64 sn = bitrev_u32((i + %uu) & (u32)(p.L - 1), p.lgL);\n", 1u << rate_bits);
    // The whole evaluation is one generic lambda instantiated twice: with glx<false> (products without the two 1-in-2^32 repairs,
    // a wave-uniform sticky word instead) and, only in a wave whose word came back non-zero, again with glx<true>.  Same protocol as
    // the hashing and NTT kernels; no list of flagged slices, no second kernel.  (The first version pushed flagged slice ids with
    // an atomic at the very end of the kernel and had a fix-up kernel redo them: that variant ended in sporadic GPU memory faults
    // that were never understood)
    s += "    auto eval = [&](auto &g) {\n";
    s += "    const u64 x = g.mul(p.xA[i & 2047], p.xB[i >> 11]);\n    const u64 z_last = gl_sub(x, p.last_inv);\n";
    s += "    const u64 l_first = p.lag[s], l_last = p.lag[p.L + s];\n";
    for (u32 j = 0; j < nch; j++) appendf(s, "    u64 acc%u = 0; const u64 al%u = p.alphas[%u];\n", j, j, j);
    for (u32 r = 0; r < air->n_regs; r++) appendf(s, "    u64 r%u = 0;\n", r);
    s += g.body;
    appendf(s, "    const u64 zi = p.zh_inv[i & %uu];\n", (1u << rate_bits) - 1);
    for (u32 j = 0; j < nch; j++) appendf(s, "    p.out[%uull * p.L + s] = g.mul(acc%u, zi);\n", j, j);
    s += "    };\n";
    s += "    glx<false> gf; gf.fl = 0;\n    eval(gf);\n";
    s += "    if ((gf.fl != 0) | p.force) { glx<true> ge; ge.fl = 0; eval(ge); }\n";
    s += "}\n";
    return s;
}
}  // namespace

// everything the generated source depends on, verbatim (the cache key of compiled kernels; its FNV-1a hash names ahead-of-time ones)
static std::string air_key(const zk_air *air, u32 C, u32 A, u32 nch, u32 rate_bits) {
    std::string key;
    u32 hdr[7] = {C, A, nch, rate_bits, air->n_regs, air->n_instrs, air->n_constants};
    key.append((const char *)hdr, sizeof hdr);
    key.append((const char *)air->instrs, (size_t)air->n_instrs * sizeof(zk_instr));
    key.append((const char *)air->constants, (size_t)air->n_constants * sizeof(u64));
    return key;
}
static u64 fnv1a(const std::string &s) {
    u64 h = 0xcbf29ce484222325ULL;
    for (unsigned char c : s) { h ^= c; h *= 0x100000001b3ULL; }
    return h;
}
// prover_quotient_aot.hip (generated by tools/gen_aot_quotient.py): kernels compiled with the library for the AIRs known at build time
const void *zk_quotient_aot_lookup(u64 key_hash);

// The source of the constraint kernel of one AIR, for the build-time generator (host only, no GPU needed).
extern "C" int32_t zk_quotient_kernel_source(const zk_air *air, uint32_t num_challenges, uint32_t rate_bits, char *buf, size_t cap,
                                             size_t *len, uint64_t *key_hash) {
    if (!air || !len || !key_hash) return ZK_ERR_INVALID_ARG;
    if (zk_air_validate(air, num_challenges) != ZK_OK) return ZK_ERR_INVALID_ARG;      // the generator indexes by operand
    const u32 A = zk_air_n_aux(air, num_challenges);
    const u64 h = fnv1a(air_key(air, air->n_trace_cols, A, num_challenges, rate_bits));
    char name[64]; snprintf(name, sizeof name, "quotient_aot_%016llx", (unsigned long long)h);
    std::string src = generate(air, air->n_trace_cols, A, num_challenges, rate_bits, name, false);
    *len = src.size(); *key_hash = h;
    if (buf) { if (cap < src.size() + 1) return ZK_ERR_BUFFER_TOO_SMALL; memcpy(buf, src.c_str(), src.size() + 1); }
    return ZK_OK;
}

// mode -1: never; 0 (default): only kernels compiled ahead of time (with the library, or by zk_prover_state_prepare / loaded from the
// prover-state directory); 1: also compile at first use for tables with >= 2^min_log coset points; 2: compile at first use, always.
// A kernel that EXISTS is used at any table size: only compiling at first use is gated by the size.
bool zk_quotient_jit_enabled(const zk_ctx *ctx, size_t L) {
    (void)L;
    return ctx->quotient_jit_mode >= 0 || ctx->quotient_aot;
}

static int32_t jit_get(zk_ctx *ctx, const zk_air *air, u32 C, u32 A, u32 nch, u32 rate_bits, bool may_compile, JitKernel &k);

int32_t zk_quotient_jit_launch(zk_ctx *ctx, const zk_air *air, u32 C, u32 A, u32 nch, u32 rate_bits, QJitParams &p) {
    if (ctx->quotient_aot) {
        // a kernel for this very program was compiled with the library: an ordinary launch, no hiprtc, no module loading
        if (const void *fn = zk_quotient_aot_lookup(fnv1a(air_key(air, C, A, nch, rate_bits)))) {
            void *args[] = {&p};
            hipError_t e;
            // ZK_DEBUG_QSYNC (debugging aid, default 0): bit 0 = wait for the stream before the launch, bit 1 = after it
            static const int qsync = getenv("ZK_DEBUG_QSYNC") ? atoi(getenv("ZK_DEBUG_QSYNC")) : 0;
            if (qsync & 1) hipStreamSynchronize(ctx->stream);
            { ProfScope ps(ctx, "quotient_eval"); e = hipLaunchKernel(fn, dim3((unsigned)((p.L + 255) / 256)), dim3(256), args, 0, ctx->stream); }
            if (qsync & 2) hipStreamSynchronize(ctx->stream);
            if (e != hipSuccess) return ctx->fail(ZK_ERR_HIP, "quotient aot: launch failed");
            return ZK_OK;
        }
    }
    if (ctx->quotient_jit_mode < 0) return ZK_JIT_UNAVAILABLE;
    const bool worth_compiling = ctx->quotient_jit_mode == 2 || (ctx->quotient_jit_mode == 1 && p.L >= ((size_t)1 << ctx->quotient_jit_min_log));
    JitKernel k;
    int32_t rc = jit_get(ctx, air, C, A, nch, rate_bits, worth_compiling, k);
    if (rc != ZK_OK) return rc;
    if (!k.fn) return ZK_JIT_UNAVAILABLE;
    void *args[] = {&p};
    const unsigned blocks = (unsigned)((p.L + 255) / 256);
    hipError_t e;
    { ProfScope ps(ctx, "quotient_eval"); e = hipModuleLaunchKernel(k.fn, blocks, 1, 1, 256, 1, 1, 0, ctx->stream, args, nullptr); }
    if (e != hipSuccess) return ctx->fail(ZK_ERR_HIP, "quotient jit: launch failed");
    return ZK_OK;
}

// The counterpart of the reference building its prover state once at start-up (/root/reference/common/src/prover_state/mod.rs:59-95,
// reached from ops::p_state, /root/reference/ops/src/lib.rs:16-18): compile the constraint kernels of the given tables now, in
// the calling thread, so that proofs never compile.  Call it from the thread that created the context, before units are handed
// to worker threads.
extern "C" int32_t zk_prover_state_prepare(zk_ctx *ctx, const zk_stark_config *cfg, uint32_t n_tables, const zk_table *tables) {
    if (!ctx || !cfg || (!tables && n_tables)) return ZK_ERR_INVALID_ARG;
    if (ctx->quotient_jit_mode < 0) return ZK_OK;
    hipSetDevice(ctx->device);
    for (uint32_t t = 0; t < n_tables; t++) {
        const zk_air *air = tables[t].air;
        if (!air) return ZK_ERR_INVALID_ARG;
        if (zk_air_validate(air, cfg->num_challenges) != ZK_OK) return ctx->fail(ZK_ERR_INVALID_ARG, "prover_state_prepare: malformed AIR");
        const u32 nch = cfg->num_challenges, A = zk_air_n_aux(air, nch);
        if (ctx->quotient_aot && zk_quotient_aot_lookup(fnv1a(air_key(air, air->n_trace_cols, A, nch, cfg->rate_bits)))) continue;   // built with the library
        JitKernel k;
        int32_t rc = jit_get(ctx, air, air->n_trace_cols, A, nch, cfg->rate_bits, true, k);
        if (rc != ZK_OK) return rc;
    }
    return ZK_OK;
}

// ---- on-disk cache of compiled kernels (see zk_prover_state_cache_dir in include/zkgl.h)
namespace {
std::string g_cache_dir; bool g_cache_dir_set = false;
uint64_t g_n_compiled = 0, g_n_loaded = 0;
struct CoHeader { char magic[8]; uint32_t version, key_len; uint64_t code_len, checksum; };     // then key bytes, then code bytes
const char CO_MAGIC[8] = {'Z', 'K', 'G', 'L', 'C', 'O', '0', '1'};
const std::string &cache_dir() {                       // g_mu held
    if (!g_cache_dir_set) { const char *e = getenv("ZK_PROVER_STATE_DIR"); g_cache_dir = e ? e : ""; g_cache_dir_set = true; }
    return g_cache_dir;
}
std::string co_path(const std::string &key) {
    char name[96]; snprintf(name, sizeof name, "/prover_state_%016llx.gfx950.co", (unsigned long long)fnv1a(key));
    return cache_dir() + name;
}
bool co_load(const std::string &key, std::vector<char> &code) {
    if (cache_dir().empty()) return false;
    FILE *f = fopen(co_path(key).c_str(), "rb");
    if (!f) return false;
    CoHeader h; bool ok = fread(&h, sizeof h, 1, f) == 1 && !memcmp(h.magic, CO_MAGIC, 8) && h.version == 1 && h.key_len == key.size() &&
                          h.code_len > 0 && h.code_len < ((uint64_t)1 << 30);
    std::string k2;
    if (ok) { k2.resize(h.key_len); ok = fread(&k2[0], 1, h.key_len, f) == h.key_len && k2 == key; }
    if (ok) { code.resize(h.code_len); ok = fread(code.data(), 1, h.code_len, f) == h.code_len && fgetc(f) == EOF; }
    fclose(f);
    if (ok) ok = fnv1a(std::string(code.data(), code.size())) == h.checksum;
    if (!ok) fprintf(stderr, "[zkgl] prover state file %s is unusable, rebuilding\n", co_path(key).c_str());
    return ok;
}
void co_store(const std::string &key, const std::vector<char> &code) {
    if (cache_dir().empty()) return;
    const std::string path = co_path(key), tmp = path + ".tmp." + std::to_string((long)getpid());
    FILE *f = fopen(tmp.c_str(), "wb");
    if (!f) { fprintf(stderr, "[zkgl] cannot create prover state file %s\n", tmp.c_str()); return; }
    CoHeader h; memcpy(h.magic, CO_MAGIC, 8); h.version = 1; h.key_len = (uint32_t)key.size(); h.code_len = code.size();
    h.checksum = fnv1a(std::string(code.data(), code.size()));
    bool ok = fwrite(&h, sizeof h, 1, f) == 1 && fwrite(key.data(), 1, key.size(), f) == key.size() && fwrite(code.data(), 1, code.size(), f) == code.size();
    ok = (fclose(f) == 0) && ok;
    if (!ok || rename(tmp.c_str(), path.c_str()) != 0) { remove(tmp.c_str()); fprintf(stderr, "[zkgl] failed to write prover state file %s\n", path.c_str()); }
}
}  // namespace
extern "C" int32_t zk_prover_state_cache_dir(const char *dir) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_cache_dir = dir ? dir : ""; g_cache_dir_set = true;
    return ZK_OK;
}
extern "C" int32_t zk_prover_state_stats(uint64_t *compiled, uint64_t *loaded_from_disk) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (compiled) *compiled = g_n_compiled;
    if (loaded_from_disk) *loaded_from_disk = g_n_loaded;
    return ZK_OK;
}

static int32_t jit_get(zk_ctx *ctx, const zk_air *air, u32 C, u32 A, u32 nch, u32 rate_bits, bool may_compile, JitKernel &k) {
    std::string key = air_key(air, C, A, nch, rate_bits);
    { u32 dev = (u32)ctx->device; key.append((const char *)&dev, sizeof dev); }
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_cache.find(key);
        if (it == g_cache.end()) {
            if (!may_compile) { k = JitKernel(); return ZK_OK; }
            // device-independent (the code object is per architecture); tied to the generator and to the field header it compiles
            // against, so a library upgrade never runs a stale kernel
            std::string disk_key = air_key(air, C, A, nch, rate_bits);
            { const u64 ver[2] = {3 /* generator version */, fnv1a(GL64_SRC)}; disk_key.append((const char *)ver, sizeof ver); }
            k.image = std::make_shared<std::vector<char>>();
            const bool from_disk = co_load(disk_key, *k.image);
            std::string src = from_disk ? std::string() : generate(air, C, A, nch, rate_bits);
            if (!from_disk) if (const char *dump = getenv("ZK_QUOTIENT_JIT_DUMP")) {       // debugging aid: the generated source, for offline inspection
                if (FILE *f = fopen(dump, "w")) { fputs(src.c_str(), f); fclose(f); }
            }
            if (!from_disk) {
            hiprtcProgram prog;
            const char *hs[] = {GL64_SRC}, *hn[] = {"gl64.cuh"};
            if (hiprtcCreateProgram(&prog, src.c_str(), "quotient_jit.hip", 1, hs, hn) != HIPRTC_SUCCESS)
                return ctx->fail(ZK_ERR_HIP, "quotient jit: hiprtcCreateProgram failed");
            const char *opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17"};
            hiprtcResult r = hiprtcCompileProgram(prog, 3, opts);
            if (r != HIPRTC_SUCCESS) {
                size_t ls = 0; hiprtcGetProgramLogSize(prog, &ls);
                std::string log(ls + 1, 0); hiprtcGetProgramLog(prog, &log[0]);
                hiprtcDestroyProgram(&prog);
                return ctx->fail(ZK_ERR_HIP, ("quotient jit: compile failed: " + log.substr(0, 400)).c_str());
            }
            size_t cs = 0; hiprtcGetCodeSize(prog, &cs);
            k.image->resize(cs);
            hiprtcGetCode(prog, k.image->data());
            hiprtcDestroyProgram(&prog);
            co_store(disk_key, *k.image);
            g_n_compiled++;
            } else g_n_loaded++;
            std::vector<char> &code = *k.image;
            if (hipModuleLoadData(&k.mod, code.data()) != hipSuccess) return ctx->fail(ZK_ERR_HIP, "quotient jit: hipModuleLoadData failed");
            if (hipModuleGetFunction(&k.fn, k.mod, "quotient_jit") != hipSuccess) return ctx->fail(ZK_ERR_HIP, "quotient jit: kernel not found in module");
            // Round 1 refused generated kernels that spill to scratch (faults were seen with such kernels on several streams).  Round 2's
            // fault hunt ran a scratch-resident kernel between the real kernels of 2400 proofs on 8 streams without a fault
            // (profiles/r02g_stress_faults.jsonl), so spilling kernels are used; ZK_QUOTIENT_JIT_NO_SCRATCH=1 refuses them again.
            static const bool no_scratch = getenv("ZK_QUOTIENT_JIT_NO_SCRATCH") && atoi(getenv("ZK_QUOTIENT_JIT_NO_SCRATCH"));
            int scratch = 0;
            if (hipFuncGetAttribute(&scratch, HIP_FUNC_ATTRIBUTE_LOCAL_SIZE_BYTES, k.fn) != hipSuccess || (no_scratch && scratch > 0)) k.fn = nullptr;
            if (k.fn) {
                // settle the freshly loaded code object before it is used in earnest: drain the stream, run the kernel once
                // with L = 0 (every thread returns at its first instruction) and wait for it
                QJitParams z = {};
                void *zargs[] = {&z};
                if (hipStreamSynchronize(ctx->stream) != hipSuccess || hipModuleLaunchKernel(k.fn, 1, 1, 1, 256, 1, 1, 0, ctx->stream, zargs, nullptr) != hipSuccess ||
                    hipStreamSynchronize(ctx->stream) != hipSuccess)
                    k.fn = nullptr;
            }
            g_cache.emplace(key, k);
        } else k = it->second;
    }
    return ZK_OK;
}
