"""ctypes mirror of the plain-data structs in include/zkgl.h (zk_instr, zk_helper_desc, zk_air, zk_stark_config)."""
import ctypes as C

u32, u64, i32 = C.c_uint32, C.c_uint64, C.c_int32
P = 0xFFFFFFFF00000001
COSET_SHIFT = 14293326489335486720

ZK_OK = 0
ERRORS = {0: "ZK_OK", -1: "ZK_ERR_INVALID_ARG", -2: "ZK_ERR_OOM", -3: "ZK_ERR_HIP", -4: "ZK_ERR_QUOTIENT_DEGREE",
          -5: "ZK_ERR_ZERO_INVERSE", -6: "ZK_ERR_POW", -7: "ZK_ERR_NO_DEVICE", -8: "ZK_ERR_ZETA_IN_SUBGROUP",
          -9: "ZK_ERR_BUFFER_TOO_SMALL"}

OPND_REG, OPND_LOCAL, OPND_NEXT, OPND_CONST, OPND_PUBLIC, OPND_CHALLENGE = range(6)
OP_ADD, OP_SUB, OP_MUL, OP_EMIT, OP_EMIT_TRANSITION, OP_EMIT_FIRST, OP_EMIT_LAST, OP_EMIT_SLOT = range(8)


def opnd(kind, idx):
    return (kind << 24) | idx


class Instr(C.Structure):
    _fields_ = [("op", u32), ("dst", u32), ("a", u32), ("b", u32)]


class HelperDesc(C.Structure):
    _fields_ = [("a0", u32), ("b0", u32), ("a1", u32), ("b1", u32)]


CTL_MAX_EXTRA = 6


class HelperExt(C.Structure):
    """zk_helper_ext"""
    _fields_ = [("filter", u32 * 2), ("n_extra", u32 * 2), ("extra", (u32 * CTL_MAX_EXTRA) * 2)]


class LookupCol(C.Structure):
    _fields_ = [("col", u32), ("filter", u32)]


class Lookup(C.Structure):
    _fields_ = [("n_cols", u32), ("table_col", u32), ("freq_col", u32), ("reserved", u32), ("cols", C.POINTER(LookupCol))]


NO_FILTER = 0xFFFFFFFF


class Air(C.Structure):
    _fields_ = [("n_trace_cols", u32), ("n_public_inputs", u32), ("n_helpers", u32), ("n_constants", u32),
                ("n_regs", u32), ("n_instrs", u32), ("constraint_degree", u32), ("reserved", u32),
                ("helpers", C.POINTER(HelperDesc)), ("constants", C.POINTER(u64)), ("instrs", C.POINTER(Instr)),
                ("lookup", C.POINTER(Lookup)), ("helpers_ext", C.POINTER(HelperExt))]


class StarkConfig(C.Structure):
    _fields_ = [("security_bits", u32), ("num_challenges", u32), ("rate_bits", u32), ("cap_height", u32),
                ("proof_of_work_bits", u32), ("fri_arity_bits", u32), ("fri_final_poly_bits", u32),
                ("num_query_rounds", u32)]


def standard_fast_config():
    """StarkConfig::standard_fast_config() -- fixed for the hot path at
    /root/reference/common/src/prover_state/circuit.rs:200-206."""
    return StarkConfig(100, 2, 1, 4, 16, 4, 5, 84)


def standard_recursion_config():
    """FRI / commitment parameters of plonky2's CircuitConfig::standard_recursion_config() (rate_bits 3, cap_height 4, PoW 16,
    ConstantArityBits(4, 5), 28 query rounds, 2 challenges) -- what the shrink / root / aggregation / block circuits behind
    /root/reference/ops/src/lib.rs:52,72,95 are proved with; used with zk_fri_prove_instance."""
    return StarkConfig(100, 2, 3, 4, 16, 4, 5, 28)


class FriPoly(C.Structure):
    _fields_ = [("oracle_index", u32), ("polynomial_index", u32)]


class FriBatch(C.Structure):
    _fields_ = [("point", u64 * 2), ("n_polys", u32), ("reserved", u32), ("polys", C.POINTER(FriPoly))]


def fri_instance(batches):
    """[(point_ext(2), [(oracle_index, polynomial_index), ...]), ...] -> (FriBatch array, keep-alive list)"""
    arr = (FriBatch * len(batches))()
    keep = []
    for i, (pt, polys) in enumerate(batches):
        pa = (FriPoly * max(1, len(polys)))(*[FriPoly(int(o), int(k)) for o, k in polys])
        keep.append(pa)
        arr[i].point[0], arr[i].point[1] = int(pt[0]), int(pt[1])
        arr[i].n_polys = len(polys)
        arr[i].polys = C.cast(pa, C.POINTER(FriPoly))
    return arr, keep
