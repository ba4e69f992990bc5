// C++ host-side mirror of the reference's operator surface over the C ABI (include/zkgl.h).
//   /root/reference/ops/src/lib.rs:16-18   fn p_state() -> &'static ProverState   (P_STATE.get().expect(..))
//   /root/reference/ops/src/lib.rs:22-62   TxProof: Operation { execute(GenerationInputs) -> Result<AggregatableProof> }
//   /root/reference/ops/src/lib.rs:64-82   AggProof: Monoid   { combine, empty }            (recursion: SURVEY.md 8f-2)
//   /root/reference/ops/src/lib.rs:84-101  BlockProof: Operation                              (recursion: SURVEY.md 8f-2)
// Same names, by-value input, errors as paladin::operation::FatalError.  Header only; link libzkgl_hip.so.
#pragma once
#include <chrono>
#include <cstdint>
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

struct ProverState {                   // common::prover_state::ProverState: built once, immutable afterwards
    int device = 0;
    zk_stark_config config;
};
inline ProverState *&p_state_slot() { static ProverState *s = nullptr; return s; }
inline std::mutex &p_state_mutex() { static std::mutex m; return m; }
// common/src/prover_state/mod.rs:59-95; a second call leaves the first state (worker/src/main.rs:25-27 only warns)
inline bool set_prover_state_from_config(int device) {
    std::lock_guard<std::mutex> g(p_state_mutex());
    if (p_state_slot()) return false;
    auto *s = new ProverState();
    s->device = device;
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

struct TxProof {
    double last_seconds = 0.0;
    AggregatableProof execute(GenerationInputs input) {
        const ProverState &st = p_state();
        zk_ctx *ctx = thread_ctx();
        auto t0 = std::chrono::steady_clock::now();
        AggregatableProof out;
        out.proof.resize(zk_proof_size_words(&st.config, (uint32_t)input.tables.size(), input.tables.data()));
        size_t len = 0;
        int32_t rc = zk_prove_tables(ctx, &st.config, (uint32_t)input.tables.size(), input.tables.data(), out.proof.data(),
                                     out.proof.size(), &len);
        if (rc != ZK_OK) throw FatalError(std::string("generate_txn_proof: ") + zk_last_error(ctx));   // lib.rs:52
        out.proof.resize(len);
        last_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();    // lib.rs:53-58
        return out;
    }
};
struct AggProof {
    AggregatableProof combine(const AggregatableProof &, const AggregatableProof &) {
        throw FatalError("generate_agg_proof: the aggregation circuit is not available (SURVEY.md 8f-2, A10); zk_prove_plonk proves circuits given as data");
    }
    AggregatableProof empty() { throw std::logic_error("empty agg proof"); }   // lib.rs:78-81 unimplemented!
};
struct BlockProof {
    AggregatableProof execute(const AggregatableProof &) {
        throw FatalError("generate_block_proof: the block circuit is not available (SURVEY.md 8f-2, A10); zk_prove_plonk proves circuits given as data");
    }
};

}  // namespace ops
