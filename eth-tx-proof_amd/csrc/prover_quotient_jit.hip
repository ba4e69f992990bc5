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
#include <functional>
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
    bool fenced = false;       // loads through the eval lambda's local base pointers tp / tn / ap_ / an (generate(): fences every QJIT_FENCE instructions)
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
                if (fenced) {
                    if (idx < C) appendf(body, "    const u64 %s = %s[%uull * Lq];\n", name, nx ? "tn" : "tp", idx);
                    else appendf(body, "    const u64 %s = %s[%uull * Lq];\n", name, nx ? "an" : "ap_", idx - C);
                } else if (idx < C) appendf(body, "    const u64 %s = p.trace_lde[%uull * p.L + %s];\n", name, idx, nx ? "sn" : "s");
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
// 2 191 instructions: 79 s; 2 431 columns / 10 195: more than half an hour), and with the basic-block fences that keep its loads in place
// hipcc gives up on the sticky word of glx<false> ("illegal VGPR to SGPR copy").  Rounds 2-3 cut such programs into non-inlined device
// functions that took the live values through memory: compile time became linear, but calls mean a stack -- scratch on every launch,
// the one thing every GPU memory fault of three rounds had in common.  Round 4: above JIT_SEG_MIN instructions the
// program becomes a SEQUENCE of kernels of at most JIT_SEG instructions, exactly like the gate programs of the Plonk half below: kernel k
// hands the accumulators and the program registers still alive at its cut (liveness analysis; a cut is moved, within JIT_SEG_SLACK
// instructions, to where the fewest are) to kernel k + 1 through a per-launch state buffer [slot][point] in HBM.  Ping-pong halves: a
// kernel reads one half and writes the other, so its exact twin (run by a wave whose sticky word came up) finds its inputs untouched.
// No calls, no stack, no scratch; every column a kernel touches is loaded by that kernel.
#define JIT_SEG_MIN 1536u
#define JIT_SEG 768u
#define JIT_SEG_SLACK 192u       // = JIT_SEG / 4
#define QJIT_FENCE 48u
struct SegPlan { std::vector<u32> cut; std::vector<std::vector<u32>> live; u32 max_live = 0; };   // segment k = [cut[k], cut[k+1]); live[k]: registers alive at cut[k+1]
// Which programs become a kernel SEQUENCE, and how long a piece is (round 5): a program of JIT_SEG_MIN instructions or more is cut into
// pieces of at most JIT_SEG (the wide zkEVM tables: compile time, round 4); a MID-SIZE program -- JIT_SEG_MID_MIN .. JIT_SEG_MIN: the
// 93-column bench AIR (844), the 71- and 165-column tables -- into pieces of at most JIT_SEG_MID.  Whole, those kernels take 217-249 vector
// registers (two waves per SIMD) and are bound by memory latency: 1.69 ms alone, 3.6-4.5 ms beside the other streams of a bench step; as
// three pieces of 96 / 150 / 108 registers the same work took 2.3-2.9 ms there (profiles/r05j_quotient_split.txt; 250, 300 and 350 tried,
// 448 leaves a 258-register piece).  ZK_QJIT_SEG_MIN / ZK_QJIT_SEG: one threshold / one length for everything (experiments, tests).
#define JIT_SEG_MID_MIN 640u
#define JIT_SEG_MID 300u
static u32 seg_min() {
    static const u32 v = getenv("ZK_QJIT_SEG_MIN") ? (u32)atoi(getenv("ZK_QJIT_SEG_MIN")) : JIT_SEG_MID_MIN;      // test / experiment switch
    return v;
}
static u32 seg_len(const zk_air *air) {
    static const u32 v = getenv("ZK_QJIT_SEG") ? (u32)atoi(getenv("ZK_QJIT_SEG")) : 0;                            // experiment switch
    if (v) return v < 64 ? 64 : v;
    return air->n_instrs >= JIT_SEG_MIN ? JIT_SEG : JIT_SEG_MID;
}
static SegPlan seg_plan(const zk_air *air, u32 seg, u32 slack) {
    const u32 n = air->n_instrs, NR = air->n_regs ? air->n_regs : 1;
    // live_before[pc]: registers read at or after pc before being written (backward scan)
    std::vector<std::vector<char>> lb(n + 1, std::vector<char>(NR, 0));
    for (u32 pc = n; pc-- > 0;) {
        lb[pc] = lb[pc + 1];
        const zk_instr &in = air->instrs[pc];
        if (in.op <= ZK_OP_MUL) lb[pc][in.dst & 0xFFFFFF] = 0;
        const u32 ops[2] = {in.a, in.b};
        for (int q = 0; q < (in.op <= ZK_OP_MUL ? 2 : 1); q++) if ((ops[q] >> 24) == ZK_OPND_REG) lb[pc][ops[q] & 0xFFFFFF] = 1;
    }
    auto count = [&](u32 pc) { u32 c = 0; for (char b : lb[pc]) c += b; return c; };
    SegPlan pl;
    pl.cut.push_back(0);
    while (n - pl.cut.back() > seg) {
        const u32 hi = pl.cut.back() + seg, lo = hi - slack;
        u32 best = hi;
        for (u32 pc = hi; pc >= lo; pc--) if (count(pc) < count(best)) best = pc;       // the latest cut with the fewest live registers
        pl.cut.push_back(best);
    }
    pl.cut.push_back(n);
    for (size_t k = 0; k + 1 < pl.cut.size(); k++) {
        std::vector<u32> lv;
        if (k + 2 < pl.cut.size()) for (u32 r = 0; r < NR; r++) if (lb[pl.cut[k + 1]][r]) lv.push_back(r);
        if (lv.size() > pl.max_live) pl.max_live = (u32)lv.size();
        pl.live.push_back(lv);
    }
    return pl;
}
static const char QJIT_PARAMS_SRC[] =
    "struct QJitParams { const u64 *trace_lde, *aux_lde, *pub, *chal, *alphas, *lag, *xA, *xB; u64 *out; u64 L; u64 last_inv; u64 zh_inv[8]; u32 lgL; u32 force; u64 *state; u32 redo_cap; u32 pad; };\n";

// one instruction of the program as source text (shared by the single-kernel and the segmented form)
static void emit_instr(Gen &g, const zk_instr &in, u32 nch) {
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

// n_kernels / n_state: how many kernels the program becomes (names kname_0 .. kname_{n-1}) and how many state words per point ONE half
// of the state buffer holds (accumulators + the most registers alive at a cut)
static std::string generate_segmented(const zk_air *air, u32 C, u32 A, u32 nch, u32 rate_bits, const char *kname, bool standalone, u32 *n_kernels, u32 *n_state) {
    const SegPlan pl = seg_plan(air, seg_len(air), seg_len(air) / 4);
    const u32 K = (u32)pl.cut.size() - 1, NS = nch + pl.max_live, NR = air->n_regs ? air->n_regs : 1;
    if (n_kernels) *n_kernels = K;
    if (n_state) *n_state = NS;
    std::string s;
    if (standalone) { s += "#include \"gl64.cuh\"\n"; s += QJIT_PARAMS_SRC; }
    for (u32 k = 0; k < K; k++) {
        Gen g; g.air = air; g.C = C; g.A = A; g.fenced = true;
        g.have_l.assign(C + A, 0); g.have_n.assign(C + A, 0); g.have_pub.assign(air->n_public_inputs + 1, 0);
        for (u32 pc = pl.cut[k]; pc < pl.cut[k + 1]; pc++) {
            if (pc > pl.cut[k] && (pc - pl.cut[k]) % QJIT_FENCE == 0)
                appendf(g.body, "    asm volatile(\"\" : \"+v\"(tp), \"+v\"(tn), \"+v\"(ap_), \"+v\"(an));\n    if (p.redo_cap == 0x7FFF0000u + %uu) return;\n", pc);
            emit_instr(g, air->instrs[pc], nch);
        }
        appendf(s, "// instructions [%u, %u)\nextern \"C\" __global__ __launch_bounds__(256) void %s_%u(QJitParams p) {\n", pl.cut[k], pl.cut[k + 1], kname, k);
        appendf(s, "    const u64 s = (u64)(p.pad ? zk_qblock(blockIdx.x, p.lgL, %uu) : blockIdx.x) * 256 + threadIdx.x;\n    if (s >= p.L) return;\n", rate_bits);
        s += "    const u32 i = bitrev_u32((u32)s, p.lgL);\n";
        appendf(s, "    const u64 sn = bitrev_u32((i + %uu) & (u32)(p.L - 1), p.lgL);\n", 1u << rate_bits);
        appendf(s, "    const u64 *sin = p.state + %s + s; u64 *sout = p.state + %s + s;\n", (k & 1) ? std::string("(u64)" + std::to_string(NS) + " * p.L").c_str() : "0",
                (k & 1) ? "0" : std::string("(u64)" + std::to_string(NS) + " * p.L").c_str());
        s += "    auto eval = [&](auto &g) {\n";
        s += "    const u64 *tp = p.trace_lde + s, *tn = p.trace_lde + sn, *ap_ = p.aux_lde + s, *an = p.aux_lde + sn;\n    const u64 Lq = p.L;\n";
        s += "    const u64 x = g.mul(p.xA[i & 2047], p.xB[i >> 11]);\n    const u64 z_last = gl_sub(x, p.last_inv);\n";
        s += "    const u64 l_first = p.lag[s], l_last = p.lag[p.L + s];\n";
        for (u32 j = 0; j < nch; j++) {
            if (k == 0) appendf(s, "    u64 acc%u = 0; const u64 al%u = p.alphas[%u];\n", j, j, j);
            else appendf(s, "    u64 acc%u = sin[%uull * Lq]; const u64 al%u = p.alphas[%u];\n", j, j, j, j);
        }
        for (u32 r = 0; r < NR; r++) appendf(s, "    u64 r%u = 0;\n", r);
        if (k > 0) for (size_t q = 0; q < pl.live[k - 1].size(); q++) appendf(s, "    r%u = sin[%lluull * Lq];\n", pl.live[k - 1][q], (unsigned long long)(nch + q));
        s += g.body;
        if (k + 1 == K) {
            appendf(s, "    const u64 zi = p.zh_inv[i & %uu];\n", (1u << rate_bits) - 1);
            for (u32 j = 0; j < nch; j++) appendf(s, "    p.out[%uull * p.L + s] = g.mul(acc%u, zi);\n", j, j);
        } else {
            for (u32 j = 0; j < nch; j++) appendf(s, "    sout[%uull * Lq] = acc%u;\n", j, j);
            for (size_t q = 0; q < pl.live[k].size(); q++) appendf(s, "    sout[%lluull * Lq] = r%u;\n", (unsigned long long)(nch + q), pl.live[k][q]);
        }
        s += "    };\n";
        s += "    glx<false> gf; gf.fl = 0;\n    eval(gf);\n";
        s += "    if ((gf.fl != 0) | p.force) { glx<true> ge; ge.fl = 0; eval(ge); }\n";
        s += "}\n";
    }
    return s;
}

std::string generate(const zk_air *air, u32 C, u32 A, u32 nch, u32 rate_bits, const char *kname = "quotient_jit", bool standalone = true,
                     u32 *n_kernels = nullptr, u32 *n_state = nullptr) {
    if (air->n_instrs >= seg_min()) return generate_segmented(air, C, A, nch, rate_bits, kname, standalone, n_kernels, n_state);
    if (n_kernels) *n_kernels = 1;
    if (n_state) *n_state = 0;
    Gen g; g.air = air; g.C = C; g.A = A; g.fenced = true;
    g.have_l.assign(C + A, 0); g.have_n.assign(C + A, 0); g.have_pub.assign(air->n_public_inputs + 1, 0);
    for (u32 pc = 0; pc < air->n_instrs; pc++) {
        // as in the gate kernels of the Plonk half: every QJIT_FENCE instructions the column base pointers pass through an empty asm and
        // an undecidable never-taken branch ends the basic block, so that hipcc cannot pull every column load to the top of the kernel
        // (93 + 40 columns, local and next rows: 163 SGPRs spilled to VGPR lanes for the hoisted offsets before, 22 now)
        if (pc && pc % QJIT_FENCE == 0)
            appendf(g.body, "    asm volatile(\"\" : \"+v\"(tp), \"+v\"(tn), \"+v\"(ap_), \"+v\"(an));\n    if (p.redo_cap == 0x7FFF0000u + %uu) return;\n", pc);
        emit_instr(g, air->instrs[pc], nch);
    }
    std::string s;
    if (standalone) { s += "#include \"gl64.cuh\"\n"; s += QJIT_PARAMS_SRC; }
    appendf(s, "extern \"C\" __global__ __launch_bounds__(256) void %s(QJitParams p) {\n", kname);
    appendf(s, "    const u64 s = (u64)(p.pad ? zk_qblock(blockIdx.x, p.lgL, %uu) : blockIdx.x) * 256 + threadIdx.x;\n    if (s >= p.L) return;\n", rate_bits);
    s += "    const u32 i = bitrev_u32((u32)s, p.lgL);\n";
    appendf(s, "    const u64 sn = bitrev_u32((i + %uu) & (u32)(p.L - 1), p.lgL);\n", 1u << rate_bits);
    // The whole evaluation is one generic lambda instantiated twice: with glx<false> (products without the two 1-in-2^32 repairs,
    // a wave-uniform sticky word instead) and, only in a wave whose word came back non-zero, again with glx<true>.  Same protocol as
    // the hashing and NTT kernels; no list of flagged slices, no second kernel.  (The first version pushed flagged slice ids with
    // an atomic at the very end of the kernel and had a fix-up kernel redo them: that variant ended in sporadic GPU memory faults
    // that were never understood)
    s += "    auto eval = [&](auto &g) {\n";
    s += "    const u64 *tp = p.trace_lde + s, *tn = p.trace_lde + sn, *ap_ = p.aux_lde + s, *an = p.aux_lde + sn;\n    const u64 Lq = p.L;\n";
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

// ------------------------------------------------------------------------------------------------ Plonk gate programs
// The vanishing-polynomial kernel of zk_prove_plonk (prover_plonk.hip: plonk_quotient_kernel) with the circuit's gate program as
// straight-line code: program registers in VGPRs, constants as immediates, every wire / constant column loaded once per kernel, every
// emit one multiply-add with alpha^t from the table.  Same field operations on the same canonical values as the interpreter, so the
// same bits.  hipcc's time grows with the square of a straight-line function's length, so a long program (the recursion gate set of
// gates.py: 5 816 instructions) becomes a SEQUENCE of kernels of at most PLK_SEG instructions, launched back to back: kernel k hands
// the accumulators and the program registers that are still alive at its cut (liveness analysis; cuts are moved to where few are --
// between two gates none) to kernel k+1 through a per-proof state buffer [slot][point] in HBM.  No calls, no stack.
// (Two single-kernel forms were tried first.  Chunks as non-inlined device functions taking registers and accumulators through
// pointers, as the tables' chunked constraint kernels do: the callees hoisted every column load, ran to ~250 VGPRs and ~100 SGPRs, and
// the kernel ended in a GPU memory fault on its first launch although the same source, run on the host, is exact and in bounds.  Chunks
// as cases of a switch inside a loop: eight minutes of hipcc and then "illegal VGPR to SGPR copy" for the sticky word of glx<false>.)
static const char PLK_PARAMS_SRC[] =
    "struct zk_instr_ { u32 op, dst, a, b; };\n"
    "struct PlkQuotParams { const u64 *w_lde, *z_lde, *cs_lde; const u64 *l0; const u64 *xA, *xB; const u64 *kis, *bg; const u64 *apow; const zk_instr_ *instrs; const u64 *gconst; "
    "const u64 *pih; u64 zh_inv[8]; u64 L; u32 lgL, W, NC, R, NZ, qdf, nchunks, nch, n_instrs, n_regs, NT; u64 *out; u64 *state; };\n";
#define PLK_SEG 768u
#define PLK_SEG_SLACK 192u
#define PLK_FENCE 64u
#define PLK_GEN_VERSION 8u

// where the program is cut and which registers cross each cut
using PlkPlan = SegPlan;
static PlkPlan plonk_plan(const zk_air *air) { return seg_plan(air, PLK_SEG, PLK_SEG_SLACK); }

struct PGen {
    const zk_air *air; u32 W;
    std::string body;
    std::vector<char> have_l, have_pub;
    std::string opnd(u32 o) {
        u32 kind = o >> 24, idx = o & 0xFFFFFF;
        char name[64];
        switch (kind) {
        case ZK_OPND_REG: snprintf(name, sizeof name, "r%u", idx); return name;
        case ZK_OPND_LOCAL:
            snprintf(name, sizeof name, "l%u", idx);
            if (!have_l[idx]) {
                have_l[idx] = 1;
                if (idx < W) appendf(body, "    const u64 %s = wp[%uull * Lq];\n", name, idx);
                else appendf(body, "    const u64 %s = cp[%uull * Lq];\n", name, idx - W);
            }
            return name;
        case ZK_OPND_CONST: {
            // small constants as immediates; full-width ones from the program's constant table in HBM (scalar loads, canonical since
            // zk_plonk_setup): as immediates, hipcc's constant hoisting parks hundreds of 64-bit literals in registers from the kernel's start
            const u64 v = gl_canon(air->constants[idx]);
            if (v < (1ull << 16)) snprintf(name, sizeof name, "0x%llxull", (unsigned long long)v);
            else snprintf(name, sizeof name, "gc[%u]", idx);
            return name;
        }
        default:
            snprintf(name, sizeof name, "pub%u", idx);
            if (!have_pub[idx]) { have_pub[idx] = 1; appendf(body, "    const u64 %s = p.pih[%u];\n", name, idx); }
            return name;
        }
    }
};

static bool plonk_small_const(const zk_air *air, u32 o, u64 *v) {
    if ((o >> 24) != ZK_OPND_CONST) return false;
    *v = gl_canon(air->constants[o & 0xFFFFFF]);
    return *v < (1ull << 16);
}
// [pc, end of run) if instructions pc.. are  MUL d, x0, c0;  (MUL t, xi, ci;  ADD d, d, t) x (n - 1)  with n >= 3 small constants whose sum
// stays below 2^31, no xi reading d or a temporary, and every temporary t dead behind the run (overwritten before it is read); else 0
static u32 plonk_dot_run(const zk_air *air, u32 pc, u32 end) {
    static const int on = getenv("ZK_PLONK_DOT_FUSE") ? atoi(getenv("ZK_PLONK_DOT_FUSE")) : 1;
    if (!on) return 0;
    const zk_instr &h = air->instrs[pc];
    if (h.op != ZK_OP_MUL) return 0;
    u64 c, sum = 0;
    const bool hb = plonk_small_const(air, h.b, &c);
    if (!hb && !plonk_small_const(air, h.a, &c)) return 0;
    sum = c;
    const u32 d = h.dst & 0xFFFFFF, dreg = ((u32)ZK_OPND_REG << 24) | d;
    std::vector<u32> temps;
    u32 q = pc + 1, n = 1;
    while (q + 1 < end) {
        const zk_instr &m = air->instrs[q], &a = air->instrs[q + 1];
        if (m.op != ZK_OP_MUL || a.op != ZK_OP_ADD) break;
        const bool mb = plonk_small_const(air, m.b, &c);
        if (!mb && !plonk_small_const(air, m.a, &c)) break;
        const u32 t = m.dst & 0xFFFFFF, treg = ((u32)ZK_OPND_REG << 24) | t, x = mb ? m.a : m.b;
        if (t == d || x == dreg || x == treg) break;
        bool x_is_temp = false;
        for (u32 tt : temps) x_is_temp |= x == (((u32)ZK_OPND_REG << 24) | tt);
        if (x_is_temp) break;
        if ((a.dst & 0xFFFFFF) != d || !((a.a == dreg && a.b == treg) || (a.a == treg && a.b == dreg))) break;
        if (sum + c >= (1ull << 31)) break;
        sum += c; temps.push_back(t); n++; q += 2;
    }
    if (n < 3) return 0;
    for (u32 t : temps) {
        const u32 treg = ((u32)ZK_OPND_REG << 24) | t;
        if ((hb ? h.a : h.b) == treg) return 0;
        bool dead = true;                                     // dead unless read before the next write
        for (u32 r = q; r < air->n_instrs; r++) {
            const zk_instr &in = air->instrs[r];
            if (in.a == treg || (in.op <= ZK_OP_MUL && in.b == treg)) { dead = false; break; }
            if (in.op <= ZK_OP_MUL && (in.dst & 0xFFFFFF) == t) break;
        }
        if (!dead) return 0;
    }
    return q;
}

// kernels kname_0 .. kname_{K-1}; *n_kernels, *n_state (state slots per point: nch accumulators + the most registers alive at a cut)
static std::string generate_plonk(const zk_air *air, u32 W, u32 NC, u32 nch, u32 rate_bits, const char *kname, bool standalone, u32 *n_kernels, u32 *n_state) {
    std::string s;
    if (standalone) { s += "#include \"gl64.cuh\"\n"; s += PLK_PARAMS_SRC; }
    const u32 NR = air->n_regs ? air->n_regs : 1;
    const PlkPlan pl = plonk_plan(air);
    const u32 K = (u32)pl.live.size();
    if (n_kernels) *n_kernels = K + 1;
    if (n_state) *n_state = nch + pl.max_live;
    // kernel 0: the permutation part [L_0 (Z - 1)] [partial-product checks] -- the loop code of the interpreter kernel, with its
    // arithmetic (canonical gl_mul: no sticky word carried around loops), accumulators to the state buffer.  32 points x 8 slots per
    // workgroup: a slot takes every eighth (challenge, chunk) pair of its point (at most three of the twenty: 24 wires instead of 160
    // dependent column loads per thread; slot 0 also the L_0 terms), the partial sums meet in LDS.  Launched with ceil(L / 32) workgroups.
    appendf(s, "extern \"C\" __global__ __launch_bounds__(256) void %s_0(PlkQuotParams p) {\n", kname);
    appendf(s, "    __shared__ u64 part[%u][8][32];\n", nch);
    s += "    const u32 pl = threadIdx.x & 31, slot = threadIdx.x >> 5;\n    const u64 s = (u64)blockIdx.x * 32 + pl;\n    const bool live = s < p.L;\n";
    for (u32 j = 0; j < nch; j++) appendf(s, "    u64 acc%u = 0;\n", j);
    s += "    if (live) {\n";
    s += "    const u32 i = bitrev_u32((u32)s, p.lgL);\n    const u64 sn = bitrev_u32((i + p.qdf) & (u32)(p.L - 1), p.lgL);\n";
    s += "    const u64 x = gl_mul(p.xA[i & 2047], p.xB[i >> 11]);\n";
    s += "    const u32 npp = p.nchunks - 1, G = p.nch * p.nchunks;\n";
    s += "    if (slot == 0) { const u64 l0v = p.l0[s];\n      for (u32 ci = 0; ci < p.nch; ci++) { const u64 t = gl_mul(l0v, gl_sub(p.z_lde[(u64)ci * p.L + s], 1));\n";
    for (u32 j = 0; j < nch; j++) appendf(s, "        acc%u = gl_mul_add(t, p.apow[%uull * p.NT + ci], acc%u);\n", j, j, j);
    s += "      } }\n";
    s += "    for (u32 g = slot; g < G; g += 8) {\n        const u32 ci = g / p.nchunks, c = g % p.nchunks, ti = p.nch + g;\n        const u64 beta = p.bg[ci], gamma = p.bg[p.nch + ci];\n";
    s += "        u64 num = 1, den = 1;\n        for (u32 u = 0; u < p.qdf; u++) {\n            const u32 j = c * p.qdf + u;\n            const u64 w = p.w_lde[(u64)j * p.L + s];\n";
    s += "            num = gl_mul(num, gl_add(gl_add(w, gl_mul(beta, gl_mul(p.kis[j], x))), gamma));\n";
    s += "            den = gl_mul(den, gl_add(gl_add(w, gl_mul(beta, p.cs_lde[(u64)(p.NC + j) * p.L + s])), gamma));\n        }\n";
    s += "        const u64 prev = c == 0 ? p.z_lde[(u64)ci * p.L + s] : p.z_lde[((u64)p.nch + (u64)ci * npp + c - 1) * p.L + s];\n";
    s += "        const u64 next = c == p.nchunks - 1 ? p.z_lde[(u64)ci * p.L + sn] : p.z_lde[((u64)p.nch + (u64)ci * npp + c) * p.L + s];\n";
    s += "        const u64 t = gl_sub(gl_mul(prev, num), gl_mul(next, den));\n";
    for (u32 j = 0; j < nch; j++) appendf(s, "        acc%u = gl_mul_add(t, p.apow[%uull * p.NT + ti], acc%u);\n", j, j, j);
    s += "    }\n    }\n";
    for (u32 j = 0; j < nch; j++) appendf(s, "    part[%u][slot][pl] = acc%u;\n", j, j);
    s += "    __syncthreads();\n";
    appendf(s, "    if (slot < %uu && live) { u64 a = part[slot][0][pl];\n", nch);
    s += "        for (int q = 1; q < 8; q++) a = gl_add(a, part[slot][q][pl]);\n        p.state[(u64)slot * p.L + s] = a; }\n";
    s += "}\n";
    u32 seq = 0;
    for (u32 k = 0; k < K; k++) {
        PGen g; g.air = air; g.W = W;
        g.have_l.assign(W + NC, 0); g.have_pub.assign(4, 0);
        u32 next_fence = PLK_FENCE;
        for (u32 pc = pl.cut[k]; pc < pl.cut[k + 1]; pc++) {
            const zk_instr &in = air->instrs[pc];
            // every PLK_FENCE instructions the column base pointers pass through an empty asm: loads behind it cannot be hoisted above it.
            // Left alone, hipcc hoists every column load of a kernel to its top, runs to 512 registers (VGPRs + AGPRs) and still spills to
            // scratch; a gate kernel with scratch on several streams at once ended in a GPU memory fault.
            // ... and a branch the compiler cannot decide (never taken: n_regs is a small number) ends the basic block there, so that the
            // scheduler does not pull the materialisation of hundreds of 64-bit constants (PoseidonGate's partial rounds) to the top either.
            if (pc - pl.cut[k] >= next_fence && (next_fence += PLK_FENCE, true))      // (a fused run below may have stepped over the multiple itself)
                appendf(g.body, "    asm volatile(\"\" : \"+v\"(wp), \"+v\"(cp), \"+s\"(gc));\n    if (p.n_regs == 0x7FFF0000u + %uu) return;\n", pc);
            // A run  d = x0 c0;  t = x1 c1;  d = d + t;  t = x2 c2;  d = d + t; ...  with small constants (an MDS row of PoseidonGate: 96 rows of
            // 12 terms, a third of the gate set's instructions) becomes ONE dot product with deferred reduction (glx::dot_acc / dot_fold).
            {
                const u32 run_end = plonk_dot_run(air, pc, pl.cut[k + 1]);
                if (run_end) {
                    const u32 d = air->instrs[pc].dst & 0xFFFFFF;
                    std::vector<std::pair<std::string, u64>> terms;
                    for (u32 q = pc; q < run_end; q += (q == pc ? 1 : 2)) {
                        const zk_instr &m = air->instrs[q];
                        u64 cv; const bool cb = plonk_small_const(air, m.b, &cv);
                        if (!cb) plonk_small_const(air, m.a, &cv);
                        terms.emplace_back(g.opnd(cb ? m.a : m.b), cv);
                    }
                    g.body += "    { u64 al = 0, ah = 0;";
                    for (auto &t : terms) appendf(g.body, " g.dot_acc(al, ah, %s, %lluu);", t.first.c_str(), (unsigned long long)t.second);
                    appendf(g.body, " r%u = g.dot_fold(al, ah); }\n", d);
                    pc = run_end - 1;
                    continue;
                }
            }
            std::string a = g.opnd(in.a);
            if (in.op <= ZK_OP_MUL) {
                std::string b = g.opnd(in.b);
                appendf(g.body, "    r%u = %s(%s, %s);\n", in.dst & 0xFFFFFF, in.op == ZK_OP_ADD ? "gl_add" : in.op == ZK_OP_SUB ? "gl_sub" : "g.mul", a.c_str(), b.c_str());
            } else {
                const u32 slot = in.op == ZK_OP_EMIT ? seq++ : in.dst;
                g.body += "    {";
                for (u32 j = 0; j < nch; j++) appendf(g.body, " acc%u = g.mul_add(%s, ap%u[%u], acc%u);", j, a.c_str(), j, slot, j);
                g.body += " }\n";
            }
        }
        appendf(s, "extern \"C\" __global__ __launch_bounds__(256) void %s_%u(PlkQuotParams p) {\n", kname, k + 1);
        s += "    const u64 s = (u64)blockIdx.x * 256 + threadIdx.x;\n    if (s >= p.L) return;\n";
        s += "    const u32 gbase = p.nch + p.nch * p.nchunks;\n";
        for (u32 j = 0; j < nch; j++) appendf(s, "    const u64 *ap%u = p.apow + %uull * p.NT + gbase;\n", j, j);
        // what the previous kernel left: read before anything is written (the exact redo starts from the same values)
        for (u32 j = 0; j < nch; j++) appendf(s, "    const u64 in_acc%u = p.state[%uull * p.L + s];\n", j, j);
        if (k > 0) for (size_t q = 0; q < pl.live[k - 1].size(); q++) appendf(s, "    const u64 in_r%u = p.state[%uull * p.L + s];\n", pl.live[k - 1][q], (u32)(nch + q));
        s += "    auto eval = [&](auto &g) {\n";
        s += "    const u64 *wp = p.w_lde + s, *cp = p.cs_lde + s, *gc = p.gconst;\n    const u64 Lq = p.L;\n";
        {
            for (u32 j = 0; j < nch; j++) appendf(s, "    u64 acc%u = in_acc%u;\n", j, j);
            for (u32 r = 0; r < NR; r++) {
                bool lv = false;
                if (k > 0) for (u32 q : pl.live[k - 1]) lv |= q == r;
                if (lv) appendf(s, "    u64 r%u = in_r%u;\n", r, r); else appendf(s, "    u64 r%u = 0;\n", r);
            }
        }
        s += g.body;
        if (k + 1 < K) {
            for (u32 j = 0; j < nch; j++) appendf(s, "    p.state[%uull * p.L + s] = acc%u;\n", j, j);
            for (size_t q = 0; q < pl.live[k].size(); q++) appendf(s, "    p.state[%uull * p.L + s] = r%u;\n", (u32)(nch + q), pl.live[k][q]);
        } else {
            s += "    const u32 i2 = bitrev_u32((u32)s, p.lgL);\n    u64 zi = p.zh_inv[0];\n";
            appendf(s, "    { const u32 zsel = i2 & %uu;\n", (1u << rate_bits) - 1);
            for (u32 j = 1; j < (1u << rate_bits); j++) appendf(s, "      zi = zsel == %uu ? p.zh_inv[%u] : zi;\n", j, j);
            s += "    }\n";
            for (u32 j = 0; j < nch; j++) appendf(s, "    p.out[%uull * p.L + s] = g.mul(acc%u, zi);\n", j, j);
        }
        s += "    };\n";
        s += "    glx<false> gf; gf.fl = 0;\n    eval(gf);\n";
        s += "    if (gf.fl != 0) { glx<true> ge; ge.fl = 0; eval(ge); }\n";
        s += "}\n";
    }
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
struct QuotAotEntry { const void *const *fns; u32 n_kernels, n_state; };
const QuotAotEntry *zk_quotient_aot_lookup(u64 key_hash);

// The source of the constraint kernel of one AIR, for the build-time generator (host only, no GPU needed).
static thread_local u32 g_last_source_kernels = 1, g_last_source_state = 0;
// what the last zk_quotient_kernel_source of this thread generated: kernels of the sequence (names <name>_0 .. when more than one) and
// state words per point and buffer half -- the build-time generator writes them into the registry
extern "C" int32_t zk_quotient_kernel_source_plan(uint32_t *n_kernels, uint32_t *n_state) {
    if (n_kernels) *n_kernels = g_last_source_kernels;
    if (n_state) *n_state = g_last_source_state;
    return ZK_OK;
}
extern "C" int32_t zk_quotient_kernel_source(const zk_air *air, uint32_t num_challenges, uint32_t rate_bits, char *buf, size_t cap,
                                             size_t *len, uint64_t *key_hash) {
    if (!air || !len || !key_hash) return ZK_ERR_INVALID_ARG;
    if (zk_air_validate(air, num_challenges) != ZK_OK) return ZK_ERR_INVALID_ARG;      // the generator indexes by operand
    const u32 A = zk_air_n_aux(air, num_challenges);
    const u64 h = fnv1a(air_key(air, air->n_trace_cols, A, num_challenges, rate_bits));
    char name[64]; snprintf(name, sizeof name, "quotient_aot_%016llx", (unsigned long long)h);
    std::string src = generate(air, air->n_trace_cols, A, num_challenges, rate_bits, name, false, &g_last_source_kernels, &g_last_source_state);
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

// the kernel(s) of one constraint program: one function, or the sequence of a segmented program with its state slots per point
struct QuotKernels { std::vector<hipFunction_t> fns; u32 n_state = 0; };
static int32_t jit_get(zk_ctx *ctx, const zk_air *air, u32 C, u32 A, u32 nch, u32 rate_bits, bool may_compile, QuotKernels &k);

int32_t zk_quotient_jit_launch(zk_ctx *ctx, const zk_air *air, u32 C, u32 A, u32 nch, u32 rate_bits, QJitParams &p) {
    if (ctx->quotient_aot) {
        // a kernel for this very program was compiled with the library: an ordinary launch, no hiprtc, no module loading
        if (const QuotAotEntry *en = zk_quotient_aot_lookup(fnv1a(air_key(air, C, A, nch, rate_bits)))) {
            // a sequence hands accumulators and live registers from kernel to kernel through a state buffer [half][slot][point], as below
            u64 *state = nullptr;
            if (en->n_kernels > 1) {
                state = (u64 *)ctx->alloc((size_t)2 * en->n_state * p.L * sizeof(u64));
                if (!state) return ZK_JIT_UNAVAILABLE;                  // the interpreter needs no such buffer
            }
            p.state = state;
            void *args[] = {&p};
            hipError_t e = hipSuccess;
            // ZK_DEBUG_QSYNC (debugging aid, default 0): bit 0 = wait for the stream before the launch, bit 1 = after it
            static const int qsync = getenv("ZK_DEBUG_QSYNC") ? atoi(getenv("ZK_DEBUG_QSYNC")) : 0;
            if (qsync & 1) hipStreamSynchronize(ctx->stream);
            {
                ProfScope ps(ctx, "quotient_eval");
                for (u32 j = 0; j < en->n_kernels && e == hipSuccess; j++) {
                    ctx->note_launch(j ? "quotient_eval(aot,k)" : "quotient_eval(aot,0)", (unsigned)((p.L + 255) / 256), 256);
                    e = hipLaunchKernel(en->fns[j], dim3((unsigned)((p.L + 255) / 256)), dim3(256), args, 0, ctx->stream);
                }
            }
            if (qsync & 2) hipStreamSynchronize(ctx->stream);
            ctx->release(state);                    // stream-ordered
            if (e != hipSuccess) return ctx->fail(ZK_ERR_HIP, "quotient aot: launch failed");
            return ZK_OK;
        }
    }
    if (ctx->quotient_jit_mode < 0) return ZK_JIT_UNAVAILABLE;
    const bool worth_compiling = ctx->quotient_jit_mode == 2 || (ctx->quotient_jit_mode == 1 && p.L >= ((size_t)1 << ctx->quotient_jit_min_log));
    QuotKernels k;
    int32_t rc = jit_get(ctx, air, C, A, nch, rate_bits, worth_compiling, k);
    if (rc != ZK_OK) return rc;
    if (k.fns.empty()) return ZK_JIT_UNAVAILABLE;
    // a segmented program hands accumulators and live registers from kernel to kernel through a state buffer [half][slot][point]
    u64 *state = nullptr;
    if (k.fns.size() > 1) {
        // 2 * n_state * L words: about 110 MB for the widest zkEVM table at L = 2^18, several GB for a wide AIR at L >= 2^23.  When the
        // device cannot give that, the proof must not fail for it: the bytecode interpreter needs no such buffer
        state = (u64 *)ctx->alloc((size_t)2 * k.n_state * p.L * sizeof(u64));
        if (!state) {
            static bool told = false;
            if (!told) { told = true; fprintf(stderr, "[zkgl] no memory for the %.1f GB state buffer of a constraint-kernel sequence: this table is evaluated by the interpreter\n", 2.0 * k.n_state * p.L * 8e-9); }
            return ZK_JIT_UNAVAILABLE;
        }
    }
    p.state = state;
    void *args[] = {&p};
    const unsigned blocks = (unsigned)((p.L + 255) / 256);
    hipError_t e = hipSuccess;
    {
        ProfScope ps(ctx, "quotient_eval");
        for (size_t j = 0; j < k.fns.size() && e == hipSuccess; j++) { ctx->note_launch(j ? "quotient_eval(compiled,k)" : "quotient_eval(compiled,0)", blocks, 256); e = hipModuleLaunchKernel(k.fns[j], blocks, 1, 1, 256, 1, 1, 0, ctx->stream, args, nullptr); }
    }
    ctx->release(state);                    // stream-ordered
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
        QuotKernels k;
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

// The code object of one generated kernel: process cache -> prover-state file -> hiprtc.  `key`: everything the source depends on;
// the file's key adds the generator version, the field header, the GPU architecture and the hiprtc version, so a library, driver or
// hardware change never loads a stale image; an image that fails to load is deleted and rebuilt.  settle_args: a zeroed parameter block
// (L = 0: every thread returns at once) for the first launch of the freshly loaded code object.
// the on-disk key of a kernel: what its source depends on + generator version, field header, GPU architecture name, hiprtc version
static std::string jit_disk_key(const std::string &base_key, const char *gcn_arch_name) {
    std::string disk_key = base_key;
    const u64 ver[2] = {7 /* generator version */, fnv1a(GL64_SRC)};
    disk_key.append((const char *)ver, sizeof ver);
    if (gcn_arch_name) disk_key.append(gcn_arch_name);
    int rmaj = 0, rmin = 0;
    if (hiprtcVersion(&rmaj, &rmin) == HIPRTC_SUCCESS) { const int rv[2] = {rmaj, rmin}; disk_key.append((const char *)rv, sizeof rv); }
    return disk_key;
}
// hiprtc for gfx950 (needs no device); on failure `log` holds the compiler's words
static bool jit_hiprtc(const std::string &src, std::vector<char> &image, std::string &log) {
    hiprtcProgram prog;
    const char *hs[] = {GL64_SRC}, *hn[] = {"gl64.cuh"};
    if (hiprtcCreateProgram(&prog, src.c_str(), "quotient_jit.hip", 1, hs, hn) != HIPRTC_SUCCESS) { log = "hiprtcCreateProgram failed"; return false; }
    const char *opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17"};
    hiprtcResult r = hiprtcCompileProgram(prog, 3, opts);
    if (r != HIPRTC_SUCCESS) {
        size_t ls = 0; hiprtcGetProgramLogSize(prog, &ls);
        log.assign(ls + 1, 0); hiprtcGetProgramLog(prog, &log[0]);
        hiprtcDestroyProgram(&prog);
        return false;
    }
    size_t cs = 0; hiprtcGetCodeSize(prog, &cs);
    image.resize(cs);
    hiprtcGetCode(prog, image.data());
    hiprtcDestroyProgram(&prog);
    return true;
}

static int32_t jit_compile_cached(zk_ctx *ctx, const std::string &base_key, bool may_compile, const std::function<std::string()> &gen, const char *kname,
                                  void *settle_args, JitKernel &k) {
    std::string key = base_key;
    { u32 dev = (u32)ctx->device; key.append((const char *)&dev, sizeof dev); }
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_cache.find(key);
    if (it != g_cache.end()) { k = it->second; return ZK_OK; }
    if (!may_compile) { k = JitKernel(); return ZK_OK; }
    std::string disk_key;
    {
        hipDeviceProp_t pr;
        disk_key = jit_disk_key(base_key, hipGetDeviceProperties(&pr, ctx->device) == hipSuccess ? pr.gcnArchName : nullptr);
    }
    for (int attempt = 0; attempt < 2; attempt++) {
        k = JitKernel();
        k.image = std::make_shared<std::vector<char>>();
        const bool from_disk = attempt == 0 && co_load(disk_key, *k.image);
        if (!from_disk) {
            std::string src = gen();
            if (const char *dump = getenv("ZK_QUOTIENT_JIT_DUMP")) {       // debugging aid: the generated source, for offline inspection
                if (FILE *f = fopen(dump, "w")) { fputs(src.c_str(), f); fclose(f); }
            }
            std::string log;
            if (!jit_hiprtc(src, *k.image, log)) return ctx->fail(ZK_ERR_HIP, ("quotient jit: compile failed: " + log.substr(0, 400)).c_str());
            co_store(disk_key, *k.image);
            g_n_compiled++;
        }
        std::vector<char> &code = *k.image;
        if (hipModuleLoadData(&k.mod, code.data()) != hipSuccess || hipModuleGetFunction(&k.fn, k.mod, kname) != hipSuccess) {
            if (from_disk) {        // a stale or foreign image behind a matching key: drop the file, build from source
                fprintf(stderr, "[zkgl] prover state file %s does not load on this device, rebuilding\n", co_path(disk_key).c_str());
                remove(co_path(disk_key).c_str());
                continue;
            }
            return ctx->fail(ZK_ERR_HIP, "quotient jit: the compiled code object does not load");
        }
        if (from_disk) g_n_loaded++;
        break;
    }
    // Round 1 refused generated kernels that spill to scratch (faults were seen with such kernels on several streams).  Round 2's
    // fault hunt ran a scratch-resident kernel between the real kernels of 2400 proofs on 8 streams without a fault
    // (profiles/r02g_stress_faults.jsonl), so spilling kernels are used; ZK_QUOTIENT_JIT_NO_SCRATCH=1 refuses them again.
    static const bool no_scratch = getenv("ZK_QUOTIENT_JIT_NO_SCRATCH") && atoi(getenv("ZK_QUOTIENT_JIT_NO_SCRATCH"));
    int scratch = 0;
    if (hipFuncGetAttribute(&scratch, HIP_FUNC_ATTRIBUTE_LOCAL_SIZE_BYTES, k.fn) != hipSuccess || (no_scratch && scratch > 0)) k.fn = nullptr;
    if (k.fn) {
        // settle the freshly loaded code object before it is used in earnest: drain the stream, run the kernel once
        // with L = 0 (every thread returns at its first instruction) and wait for it
        void *zargs[] = {settle_args};
        if (hipStreamSynchronize(ctx->stream) != hipSuccess || hipModuleLaunchKernel(k.fn, 1, 1, 1, 256, 1, 1, 0, ctx->stream, zargs, nullptr) != hipSuccess ||
            hipStreamSynchronize(ctx->stream) != hipSuccess)
            k.fn = nullptr;
    }
    g_cache.emplace(key, k);
    return ZK_OK;
}

static std::mutex g_qmu;
static std::map<std::string, QuotKernels> g_qcache;        // (program, device) -> its kernels, once they exist
static int32_t jit_get(zk_ctx *ctx, const zk_air *air, u32 C, u32 A, u32 nch, u32 rate_bits, bool may_compile, QuotKernels &out) {
    out = QuotKernels();
    std::string qkey = air_key(air, C, A, nch, rate_bits);
    { u32 dev = (u32)ctx->device; qkey.append((const char *)&dev, sizeof dev); }
    {
        std::lock_guard<std::mutex> lk(g_qmu);
        auto it = g_qcache.find(qkey);
        if (it != g_qcache.end()) { out = it->second; return ZK_OK; }
    }
    QJitParams z = {};
    JitKernel k;
    const bool seg = air->n_instrs >= seg_min();
    u32 K = 1, ns = 0;
    std::string key = air_key(air, C, A, nch, rate_bits);
    if (seg) {
        const SegPlan pl = seg_plan(air, seg_len(air), seg_len(air) / 4);   // the plan alone: cheap next to a compile
        K = (u32)pl.cut.size() - 1; ns = nch + pl.max_live;
        const u32 hdr[4] = {0x53454751u /* "SEGQ" */, seg_len(air), seg_len(air) / 4, QJIT_FENCE};
        key.append((const char *)hdr, sizeof hdr);
    }
    int32_t rc = jit_compile_cached(ctx, key, may_compile, [&]() { return generate(air, C, A, nch, rate_bits); }, seg ? "quotient_jit_0" : "quotient_jit", &z, k);
    if (rc != ZK_OK || !k.fn) return rc;
    out.fns.push_back(k.fn); out.n_state = ns;
    for (u32 j = 1; j < K; j++) {
        char nm[64]; snprintf(nm, sizeof nm, "quotient_jit_%u", j);
        hipFunction_t f = nullptr;
        if (hipModuleGetFunction(&f, k.mod, nm) != hipSuccess) { out = QuotKernels(); return ctx->fail(ZK_ERR_HIP, "quotient jit: a kernel of the sequence is missing from the module"); }
        // as for the first kernel (jit_compile_cached): no scratch if so required.  (No settling launch of its own: the module was loaded
        // and its first kernel launched and waited for in jit_compile_cached -- what the settling launch is about is the freshly loaded
        // code object, not the individual entry point.)
        static const bool no_scratch = getenv("ZK_QUOTIENT_JIT_NO_SCRATCH") && atoi(getenv("ZK_QUOTIENT_JIT_NO_SCRATCH"));
        int scratch = 0;
        if (hipFuncGetAttribute(&scratch, HIP_FUNC_ATTRIBUTE_LOCAL_SIZE_BYTES, f) != hipSuccess || (no_scratch && scratch > 0)) { out = QuotKernels(); return ZK_OK; }
        out.fns.push_back(f);
    }
    std::lock_guard<std::mutex> lk(g_qmu);
    g_qcache[qkey] = out;
    return ZK_OK;
}

// The prover-state files WITHOUT a GPU (VERDICT r4 #5a): __graft_entry__.build() -- and whoever packages the library -- writes the compiled
// constraint kernels of the given AIRs into the prover-state directory, so that a clean clone proves its first transaction with compiled
// kernels instead of interpreting (or spending 100 s in hiprtc inside a proof).  The same source, key and file format as
// zk_prover_state_prepare; gcn_arch_name is what hipGetDeviceProperties reports on the target ("gfx950:sramecc+:xnack-" on MI355X) and is
// part of the key: a file written for another target is simply not found.  Tables whose kernel was built with the library are skipped.
extern "C" int32_t zk_prover_state_build_offline(const zk_stark_config *cfg, uint32_t n_tables, const zk_table *tables, const char *gcn_arch_name,
                                                 uint64_t *n_built, uint64_t *n_present) {
    if (!cfg || (!tables && n_tables) || !gcn_arch_name) return ZK_ERR_INVALID_ARG;
    uint64_t built = 0, present = 0;
    for (uint32_t t = 0; t < n_tables; t++) {
        const zk_air *air = tables[t].air;
        if (!air || zk_air_validate(air, cfg->num_challenges) != ZK_OK) return ZK_ERR_INVALID_ARG;
        const u32 nch = cfg->num_challenges, A = zk_air_n_aux(air, nch), C = air->n_trace_cols;
        if (zk_quotient_aot_lookup(fnv1a(air_key(air, C, A, nch, cfg->rate_bits)))) continue;
        std::string key = air_key(air, C, A, nch, cfg->rate_bits);
        if (air->n_instrs >= seg_min()) {                       // exactly jit_get's key of a kernel sequence
            const u32 hdr[4] = {0x53454751u /* "SEGQ" */, seg_len(air), seg_len(air) / 4, QJIT_FENCE};
            key.append((const char *)hdr, sizeof hdr);
        }
        const std::string disk_key = jit_disk_key(key, gcn_arch_name);
        std::vector<char> image;
        {
            std::lock_guard<std::mutex> lk(g_mu);
            if (cache_dir().empty()) return ZK_ERR_INVALID_ARG;
            if (co_load(disk_key, image)) { present++; continue; }
        }
        std::string log;
        if (!jit_hiprtc(generate(air, C, A, nch, cfg->rate_bits), image, log)) { fprintf(stderr, "[zkgl] offline build: %s\n", log.substr(0, 600).c_str()); return ZK_ERR_HIP; }
        std::lock_guard<std::mutex> lk(g_mu);
        co_store(disk_key, image);
        built++;
    }
    if (n_built) *n_built = built;
    if (n_present) *n_present = present;
    return ZK_OK;
}

// ---- Plonk gate programs: build-time registry (prover_plonk_aot.hip, tools/gen_aot_quotient.py), else hiprtc at zk_plonk_setup
struct PlkAotEntry { const void *const *fns; u32 n_kernels, n_state; };
const PlkAotEntry *zk_plonk_gates_aot_lookup(u64 key_hash);
static std::string plonk_key(const zk_air *g, u32 W, u32 NC, u32 nch, u32 rate_bits) {
    std::string key = "PLONK";
    u32 hdr[11] = {W, NC, nch, rate_bits, g->n_regs, g->n_instrs, g->n_constants, PLK_SEG, PLK_SEG_SLACK, PLK_FENCE, PLK_GEN_VERSION};
    key.append((const char *)hdr, sizeof hdr);
    key.append((const char *)g->instrs, (size_t)g->n_instrs * sizeof(zk_instr));
    key.append((const char *)g->constants, (size_t)g->n_constants * sizeof(u64));
    return key;
}
// ZK_PLONK_JIT: 0 = interpreter always; 1 (default) = build-time kernels, and hiprtc for programs of at least 256 instructions;
// 2 = hiprtc for every program that has no build-time kernel
int32_t zk_plonk_gates_kernel(zk_ctx *ctx, const zk_air *gates, u32 W, u32 NC, u32 nch, u32 rate_bits, std::vector<const void *> &fns, int *is_module, u32 *n_state) {
    fns.clear(); *is_module = 0; *n_state = 0;
    static const int mode = getenv("ZK_PLONK_JIT") ? atoi(getenv("ZK_PLONK_JIT")) : 1;
    if (mode <= 0 || gates->n_instrs == 0) return ZK_OK;
    const std::string key = plonk_key(gates, W, NC, nch, rate_bits);
    if (ctx->quotient_aot)
        if (const PlkAotEntry *e = zk_plonk_gates_aot_lookup(fnv1a(key))) { fns.assign(e->fns, e->fns + e->n_kernels); *n_state = e->n_state; return ZK_OK; }
    if (mode == 1 && gates->n_instrs < 256) return ZK_OK;
    u32 K = 0, ns = 0;
    (void)generate_plonk(gates, W, NC, nch, rate_bits, "plonk_gates_jit", true, &K, &ns);       // the plan (cheap next to the compile)
    JitKernel k0;
    PlkQuotParams z = {};
    int32_t rc = jit_compile_cached(ctx, key, true, [&]() { return generate_plonk(gates, W, NC, nch, rate_bits, "plonk_gates_jit", true, nullptr, nullptr); },
                                    "plonk_gates_jit_0", &z, k0);
    if (rc != ZK_OK) return rc;
    if (!k0.fn) return ZK_OK;
    fns.push_back((const void *)k0.fn);
    for (u32 k = 1; k < K; k++) {
        char nm[64]; snprintf(nm, sizeof nm, "plonk_gates_jit_%u", k);
        hipFunction_t f = nullptr;
        if (hipModuleGetFunction(&f, k0.mod, nm) != hipSuccess) { fns.clear(); return ctx->fail(ZK_ERR_HIP, "plonk gate kernels: a kernel of the sequence is missing from the module"); }
        fns.push_back((const void *)f);
    }
    *is_module = 1; *n_state = ns;
    return ZK_OK;
}
// The source of the gate kernels of one program, for the build-time generator (host only).  W / NC: wires / constant columns of the circuit.
extern "C" int32_t zk_plonk_gates_kernel_source(const zk_air *gates, uint32_t num_wires, uint32_t num_constants, uint32_t num_challenges, uint32_t rate_bits,
                                                char *buf, size_t cap, size_t *len, uint64_t *key_hash, uint32_t *n_kernels, uint32_t *n_state) {
    if (!gates || !len || !key_hash || (gates->n_instrs && !gates->instrs) || (gates->n_constants && !gates->constants)) return ZK_ERR_INVALID_ARG;
    for (u32 pc = 0; pc < gates->n_instrs; pc++) {       // the generator indexes by operand
        const zk_instr &in = gates->instrs[pc];
        if (in.op > ZK_OP_EMIT && in.op != ZK_OP_EMIT_SLOT) return ZK_ERR_INVALID_ARG;
        const u32 ops[2] = {in.a, in.b};
        for (int q = 0; q < (in.op <= ZK_OP_MUL ? 2 : 1); q++) {
            const u32 kd = ops[q] >> 24, ix = ops[q] & 0xFFFFFFu;
            if (!(kd == ZK_OPND_REG ? ix < gates->n_regs : kd == ZK_OPND_LOCAL ? ix < num_wires + num_constants : kd == ZK_OPND_CONST ? ix < gates->n_constants : kd == ZK_OPND_PUBLIC ? ix < 4 : false))
                return ZK_ERR_INVALID_ARG;
        }
        if (in.op <= ZK_OP_MUL && (in.dst & 0xFFFFFFu) >= gates->n_regs) return ZK_ERR_INVALID_ARG;
    }
    const u64 h = fnv1a(plonk_key(gates, num_wires, num_constants, num_challenges, rate_bits));
    char name[64]; snprintf(name, sizeof name, "plonk_gates_aot_%016llx", (unsigned long long)h);
    u32 K = 0, ns = 0;
    std::string src = generate_plonk(gates, num_wires, num_constants, num_challenges, rate_bits, name, false, &K, &ns);
    *len = src.size(); *key_hash = h;
    if (n_kernels) *n_kernels = K;
    if (n_state) *n_state = ns;
    if (buf) { if (cap < src.size() + 1) return ZK_ERR_BUFFER_TOO_SMALL; memcpy(buf, src.c_str(), src.size() + 1); }
    return ZK_OK;
}
