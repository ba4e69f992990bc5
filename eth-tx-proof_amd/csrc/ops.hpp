// C++ host-side mirror of the reference's operator surface over the C ABI (include/zkgl.h).
//   /root/reference/ops/src/lib.rs:16-18   fn p_state() -> &'static ProverState   (P_STATE.get().expect(..))
//   /root/reference/ops/src/lib.rs:22-62   TxProof: Operation { execute(GenerationInputs) -> Result<AggregatableProof> }
//                                          = prove_root: zk_prove_tables, per table initial wrapper + shrinking wrappers, root proof
//   /root/reference/ops/src/lib.rs:64-82   AggProof: Monoid   { combine, empty }            (zk_prove_plonk)
//   /root/reference/ops/src/lib.rs:84-101  BlockProof: Operation                              (zk_prove_plonk)
//   /root/reference/leader/src/prover.rs:14-43  ProverInput::prove = map(&TxProof).fold(&AggProof) -> BlockProof
// Same names, by-value input, errors as paladin::operation::FatalError.  Header only; link libzkgl_hip.so.
// The recursion CIRCUITS (AllRecursiveCircuits: gate sets, verifier gadgets, witness generators) live in crates absent from the
// reference tree: they are an input here (ProverState::recursion); without them combine / execute throw FatalError.
#pragma once
#include <chrono>
#include <cstdint>
#include <functional>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>
#include "../../include/zkgl.h"

namespace ops {

struct FatalError : std::runtime_error { using std::runtime_error::runtime_error; };

// Upstream GenerationInputs is the witness of one transaction; trace generation (CPU EVM interpreter) is out of scope
// (SURVEY.md 8a row a4), so the input carries the generated traces of the unit's tables.
struct GenerationInputs {
    std::vector<zk_table> tables;
    uint64_t txn_number_before = 0, block_number = 0;
    std::vector<uint8_t> signed_txn;   // empty for a dummy (padding) transaction
};
struct AggregatableProof {
    enum Kind { Txn, Agg } kind = Txn;
    std::vector<uint64_t> proof;       // words as documented in zkgl.h
};

struct GeneratedBlockProof { uint64_t b_height = 0; std::vector<uint64_t> intern; };     // leader/src/prover.rs:9,36-40

// one Plonk proof to produce: circuit prover data (zk_plonk_setup), its config, the witness, the public inputs
struct PlonkJob {
    const zk_stark_config *cfg = nullptr;
    zk_plonk_state *state = nullptr;
    const zk_plonk_circuit *circuit = nullptr;
    const uint64_t *wires = nullptr;
    uint32_t wires_on_device = 0;
    std::vector<uint64_t> public_inputs;
};
struct RecursionCircuits {             // supplied by the integrator; called on the worker thread with its context
    // by_table[table].by_stark_size[log_n]: number of circuits in the chain (1 initial wrapper + shrinking wrappers); 0 = "Missing
    // preprocessed circuits for table with size" (FatalError).  Step 0 gets the table's StarkProof words, step i > 0 the proof of step i-1.
    std::function<uint32_t(uint32_t table, const zk_table &)> shrink_chain_len;
    std::function<PlonkJob(zk_ctx *, uint32_t table, const zk_table &, uint32_t step, const std::vector<uint64_t> &inner)> shrink_step;
    std::function<PlonkJob(zk_ctx *, const std::vector<std::vector<uint64_t>> &shrunk, const std::vector<zk_table> &)> root;
    std::function<PlonkJob(zk_ctx *, const AggregatableProof &, const AggregatableProof &)> aggregation;
    std::function<PlonkJob(zk_ctx *, const std::vector<uint64_t> *prev, const AggregatableProof &)> block;
};

struct ProverState {                   // common::prover_state::ProverState: built once, immutable afterwards
    int device = 0;
    zk_stark_config config;
    RecursionCircuits recursion;
    bool stark_only = false;           // TxProof stops after the STARK proofs (BASELINE's headline unit), on explicit request only
};
inline ProverState *&p_state_slot() { static ProverState *s = nullptr; return s; }
inline std::mutex &p_state_mutex() { static std::mutex m; return m; }
// common/src/prover_state/mod.rs:59-95; a second call leaves the first state (worker/src/main.rs:25-27 only warns)
inline bool set_prover_state_from_config(int device, RecursionCircuits recursion = {}, bool stark_only = false) {
    std::lock_guard<std::mutex> g(p_state_mutex());
    if (p_state_slot()) return false;
    auto *s = new ProverState();
    s->device = device;
    s->recursion = std::move(recursion);
    s->stark_only = stark_only;
    zk_stark_config_standard_fast(&s->config);     // circuit.rs:200-206
    p_state_slot() = s;
    return true;
}
inline const ProverState &p_state() {
    if (!p_state_slot()) throw std::logic_error("Prover state is not initialized");   // lib.rs:17
    return *p_state_slot();
}
// one zk_ctx per worker thread (paladin `--num-workers N` runs N executions concurrently over one P_STATE)
inline zk_ctx *thread_ctx() {
    thread_local zk_ctx *ctx = nullptr;
    if (!ctx) {
        int32_t rc = zk_ctx_create(p_state().device, nullptr, &ctx);
        if (rc != ZK_OK) throw FatalError("zk_ctx_create failed: " + std::to_string(rc));
    }
    return ctx;
}

inline std::vector<uint64_t> prove_plonk_job(const char *what, const PlonkJob &job) {
    zk_ctx *ctx = thread_ctx();
    std::vector<uint64_t> proof(zk_plonk_proof_size_words(job.cfg, job.circuit));
    size_t len = 0;
    int32_t rc = zk_prove_plonk(ctx, job.cfg, job.state, job.wires, job.wires_on_device, job.public_inputs.data(),
                                (uint32_t)job.public_inputs.size(), proof.data(), proof.size(), &len);
    if (rc != ZK_OK) throw FatalError(std::string(what) + ": " + zk_last_error(ctx));               // lib.rs:52,72,96
    proof.resize(len);
    return proof;
}
// evm_arithmetization::prover::prove: the STARK proofs of all tables on one challenger
inline std::vector<uint64_t> prove(const GenerationInputs &input) {
    const ProverState &st = p_state();
    zk_ctx *ctx = thread_ctx();
    std::vector<uint64_t> out(zk_proof_size_words(&st.config, (uint32_t)input.tables.size(), input.tables.data()));
    size_t len = 0;
    int32_t rc = zk_prove_tables(ctx, &st.config, (uint32_t)input.tables.size(), input.tables.data(), out.data(), out.size(), &len);
    if (rc != ZK_OK) throw FatalError(std::string("generate_txn_proof: ") + zk_last_error(ctx));   // lib.rs:52
    out.resize(len);
    return out;
}
// AllRecursiveCircuits::prove_root: prove(), then per table the shrink chain, then the root circuit
inline std::vector<uint64_t> prove_root(const GenerationInputs &input) {
    const ProverState &st = p_state();
    const RecursionCircuits &rc = st.recursion;
    if (!rc.shrink_chain_len || !rc.shrink_step || !rc.root)
        throw FatalError("generate_txn_proof: no recursion circuits in the prover state (SURVEY.md 8f-2, A10)");
    zk_ctx *ctx = thread_ctx();
    const std::vector<uint64_t> all_proof = prove(input);
    std::vector<std::vector<uint64_t>> shrunk;
    size_t at = 2;                                         // "ZKGLPRF1", n_tables, then the tables' StarkProof words back to back
    for (uint32_t t = 0; t < input.tables.size(); t++) {
        const size_t n = zk_proof_size_words(&st.config, 1, &input.tables[t]) - 2;
        std::vector<uint64_t> proof(all_proof.begin() + at, all_proof.begin() + at + n);
        at += n;
        const uint32_t steps = rc.shrink_chain_len(t, input.tables[t]);
        if (!steps) throw FatalError("Missing preprocessed circuits for table " + std::to_string(t) + " with size " + std::to_string(input.tables[t].log_n) + ".");
        for (uint32_t i = 0; i < steps; i++) proof = prove_plonk_job("shrink", rc.shrink_step(ctx, t, input.tables[t], i, proof));
        shrunk.push_back(std::move(proof));
    }
    return prove_plonk_job("root", rc.root(ctx, shrunk, input.tables));
}

struct TxProof {
    double last_seconds = 0.0;
    AggregatableProof execute(GenerationInputs input) {
        const ProverState &st = p_state();
        auto t0 = std::chrono::steady_clock::now();
        AggregatableProof out;
        out.proof = st.stark_only ? prove(input) : prove_root(input);                                // lib.rs:52
        last_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();    // lib.rs:53-58
        return out;
    }
};
struct AggProof {
    AggregatableProof combine(const AggregatableProof &a, const AggregatableProof &b) {
        const ProverState &st = p_state();
        if (!st.recursion.aggregation)
            throw FatalError("generate_agg_proof: no recursion circuits in the prover state (SURVEY.md 8f-2, A10); zk_prove_plonk proves circuits given as data");
        AggregatableProof out;
        out.kind = AggregatableProof::Agg;
        out.proof = prove_plonk_job("generate_agg_proof", st.recursion.aggregation(thread_ctx(), a, b));
        return out;
    }
    AggregatableProof empty() { throw std::logic_error("empty agg proof"); }   // lib.rs:78-81 unimplemented!
};
struct BlockProof {
    const std::vector<uint64_t> *prev = nullptr;
    GeneratedBlockProof execute(const AggregatableProof &agg) {
        const ProverState &st = p_state();
        if (!st.recursion.block)
            throw FatalError("generate_block_proof: no recursion circuits in the prover state (SURVEY.md 8f-2, A10); zk_prove_plonk proves circuits given as data");
        PlonkJob job = st.recursion.block(thread_ctx(), prev, agg);
        GeneratedBlockProof out;
        out.b_height = job.public_inputs.empty() ? 0 : job.public_inputs[0];
        out.intern = prove_plonk_job("generate_block_proof", job);
        return out;
    }
};
// leader/src/prover.rs:20-43 on the calling thread: map, balanced pairwise fold (combine is associative), block proof
struct ProverInput {
    std::vector<GenerationInputs> proof_gen_ir;
    GeneratedBlockProof prove(const std::vector<uint64_t> *previous = nullptr) {
        std::vector<AggregatableProof> level;
        for (auto &in : proof_gen_ir) level.push_back(TxProof{}.execute(in));                      // .map(&TxProof)
        if (level.empty()) AggProof{}.empty();
        while (level.size() > 1) {                                                                  // .fold(&AggProof)
            std::vector<AggregatableProof> next;
            for (size_t j = 0; j + 1 < level.size(); j += 2) next.push_back(AggProof{}.combine(level[j], level[j + 1]));
            if (level.size() & 1) next.push_back(level.back());
            level.swap(next);
        }
        if (level[0].kind != AggregatableProof::Agg) throw std::runtime_error("AggProof is is not GeneratedAggProof");   // prover.rs:41
        BlockProof bp; bp.prev = previous;
        return bp.execute(level[0]);                                                                // Literal(proof).map(&BlockProof)
    }
};

}  // namespace ops
