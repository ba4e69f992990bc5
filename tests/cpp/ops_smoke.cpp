// Compiles and (on a GPU box) runs the C++ operator mirror: ops::set_prover_state_from_config + ops::TxProof::execute,
// reading like the reference's call site (/root/reference/ops/src/lib.rs:29-61), then the caller's shape
// (TxProof = prove_root: STARK proofs -> shrink chain -> root proof), then
// ProverInput::prove = map(&TxProof).fold(&AggProof) -> BlockProof (/root/reference/leader/src/prover.rs:20-43) with a tiny
// stand-in recursion circuit (identity permutation, no gates: any witness satisfies it).  Exit code 0 = proofs produced.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../../eth-tx-proof_amd/csrc/ops.hpp"

typedef unsigned __int128 u128;
static const uint64_t GLP = 0xFFFFFFFF00000001ULL;
static uint64_t fmul(uint64_t a, uint64_t b) { return (uint64_t)((u128)a * b % GLP); }
static uint64_t fpow(uint64_t b, uint64_t e) { uint64_t r = 1; while (e) { if (e & 1) r = fmul(r, b); b = fmul(b, b); e >>= 1; } return r; }

// stand-in recursion circuit as plain data: 8 routed wires, 2^4 rows, sigma = identity (sigma_j(w^r) = k_j w^r, k_j = g^j)
struct TinyCircuit {
    static constexpr uint32_t K = 4, W = 8, N = 1u << K;
    zk_stark_config cfg;
    zk_air gates = {};
    zk_plonk_circuit circ = {};
    std::vector<uint64_t> constants, sigmas, wires;
    zk_plonk_state *state = nullptr;
    TinyCircuit() {
        zk_stark_config_standard_recursion(&cfg);
        gates.n_trace_cols = W + 1; gates.n_regs = 1; gates.constraint_degree = 1;
        constants.assign(N, 1);
        const uint64_t w = fpow(7277203076849721926ULL, 1ULL << (32 - K)), g = 14293326489335486720ULL;
        sigmas.resize((size_t)W * N); wires.resize((size_t)W * N);
        for (uint32_t j = 0; j < W; j++) {
            uint64_t x = fpow(g, j);
            for (uint32_t r = 0; r < N; r++) { sigmas[(size_t)j * N + r] = x; x = fmul(x, w); wires[(size_t)j * N + r] = 1000 * j + r; }
        }
        circ.degree_bits = K; circ.num_wires = W; circ.num_routed_wires = W; circ.num_constants = 1; circ.quotient_degree_factor = 8;
        circ.gates = &gates; circ.constants = constants.data(); circ.sigmas = sigmas.data();
    }
    ops::PlonkJob job(zk_ctx *ctx, std::vector<uint64_t> pubs) {
        if (!state && zk_plonk_setup(ctx, &cfg, &circ, &state) != ZK_OK) throw ops::FatalError(std::string("zk_plonk_setup: ") + zk_last_error(ctx));
        ops::PlonkJob j;
        j.cfg = &cfg; j.state = state; j.circuit = &circ; j.wires = wires.data(); j.public_inputs = std::move(pubs);
        return j;
    }
};

int main() {
    // tiny AIR: one column counting up by one (transition constraint next - local - 1 == 0)
    uint64_t constants[1] = {1};
    zk_instr prog[3] = {{ZK_OP_ADD, 0, ZK_OPND(ZK_OPND_LOCAL, 0), ZK_OPND(ZK_OPND_CONST, 0)},
                        {ZK_OP_SUB, 0, ZK_OPND(ZK_OPND_NEXT, 0), ZK_OPND(ZK_OPND_REG, 0)},
                        {ZK_OP_EMIT_TRANSITION, 0, ZK_OPND(ZK_OPND_REG, 0), 0}};
    zk_air air = {};
    air.n_trace_cols = 1; air.n_constants = 1; air.n_regs = 1; air.n_instrs = 3; air.constraint_degree = 3;
    air.constants = constants; air.instrs = prog;
    const uint32_t log_n = 8;
    std::vector<uint64_t> trace(1u << log_n);
    for (size_t i = 0; i < trace.size(); i++) trace[i] = 1000 + i;
    zk_table t = {&air, log_n, 0, trace.data(), nullptr};

    bool threw = false;
    try { ops::p_state(); } catch (const std::logic_error &) { threw = true; }     // "Prover state is not initialized"
    if (!threw) return 2;
    static TinyCircuit tiny;
    ops::RecursionCircuits rec;
    // children are bound through the public inputs only (first proof word of each child + lengths): a stand-in, not recursion
    rec.aggregation = [](zk_ctx *ctx, const ops::AggregatableProof &a, const ops::AggregatableProof &b) {
        return tiny.job(ctx, {a.proof.size(), b.proof.size(), a.proof[1], b.proof[1]});
    };
    // prove_root: the table's StarkProof goes through a chain of two wrappers (initial + one shrinking), then the root circuit
    rec.shrink_chain_len = [](uint32_t, const zk_table &) { return 2u; };
    rec.shrink_step = [](zk_ctx *ctx, uint32_t table, const zk_table &t, uint32_t step, const std::vector<uint64_t> &inner) {
        return tiny.job(ctx, {table, t.log_n, step, inner.size(), inner[inner.size() / 2]});
    };
    rec.root = [](zk_ctx *ctx, const std::vector<std::vector<uint64_t>> &shrunk, const std::vector<zk_table> &tables) {
        return tiny.job(ctx, {tables.size(), shrunk.size(), shrunk[0].size(), shrunk[0][7]});
    };
    rec.block = [](zk_ctx *ctx, const std::vector<uint64_t> *prev, const ops::AggregatableProof &agg) {
        return tiny.job(ctx, {7 /* height */, agg.proof.size(), prev ? prev->size() : 0});
    };
    ops::set_prover_state_from_config(0, rec);
    if (ops::set_prover_state_from_config(0)) return 3;                              // second init keeps the first
    ops::GenerationInputs in;
    in.tables.push_back(t);
    try {
        ops::AggregatableProof p = ops::TxProof{}.execute(in);
        std::printf("proof words: %zu, magic %016llx\n", p.proof.size(), (unsigned long long)p.proof[0]);
        if (p.proof[0] != 0x314B4C504C474B5AULL) return 4;                           // "ZKGLPLK1": the ROOT proof, a Plonk proof
        if (ops::prove(in)[0] != 0x31465250474C4B5AULL) return 8;                    // "ZKGLPRF1": the STARK half alone
        ops::ProverInput job;
        for (int u = 0; u < 3; u++) job.proof_gen_ir.push_back(in);
        ops::GeneratedBlockProof blk = job.prove();
        std::printf("block proof words: %zu, height %llu\n", blk.intern.size(), (unsigned long long)blk.b_height);
        if (blk.b_height != 7 || blk.intern.size() != zk_plonk_proof_size_words(&tiny.cfg, &tiny.circ)) return 5;
        ops::ProverInput one;
        one.proof_gen_ir.push_back(in);
        try { one.prove(); return 6; } catch (const std::runtime_error &e) { if (!std::strstr(e.what(), "not GeneratedAggProof")) return 7; }
    } catch (const ops::FatalError &e) {
        std::printf("FatalError: %s\n", e.what());
        return 1;                                                                    // expected on a box without a GPU
    }
    return 0;
}
