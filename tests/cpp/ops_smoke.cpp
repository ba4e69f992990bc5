// Compiles and (on a GPU box) runs the C++ operator mirror: ops::set_prover_state_from_config + ops::TxProof::execute,
// reading like the reference's call site (/root/reference/ops/src/lib.rs:29-61).  Exit code 0 = proof produced.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../eth-tx-proof_amd/csrc/ops.hpp"

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
    ops::set_prover_state_from_config(0);
    if (ops::set_prover_state_from_config(0)) return 3;                              // second init keeps the first
    ops::GenerationInputs in;
    in.tables.push_back(t);
    try {
        ops::AggregatableProof p = ops::TxProof{}.execute(in);
        std::printf("proof words: %zu, magic %016llx\n", p.proof.size(), (unsigned long long)p.proof[0]);
        if (p.proof[0] != 0x31465250474C4B5AULL) return 4;
        try { ops::AggProof{}.combine(p, p); return 5; } catch (const ops::FatalError &) {}
    } catch (const ops::FatalError &e) {
        std::printf("FatalError: %s\n", e.what());
        return 1;                                                                    // expected on a box without a GPU
    }
    return 0;
}
