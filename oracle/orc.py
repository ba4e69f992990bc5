"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE (tests/, smoke(), bench.py cpu_baseline only)."""
import ctypes as C
import os
import subprocess

import numpy as np

_here = os.path.dirname(os.path.abspath(__file__))
_lib = None
u64p = C.POINTER(C.c_uint64)


def build(force=False):
    if os.environ.get("ORC_LIB"):                     # the sanitizer build (make -C oracle asan), chosen by the test that preloads libasan
        return os.environ["ORC_LIB"]
    so = os.path.join(_here, "liboracle.so")
    srcs = [os.path.join(_here, f) for f in ("prims.c", "stark.c", "verify.c", "plonk.c", "gl.h", "oracle.h", "poseidon_constants.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _here, "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        L = _lib
        L.orc_gl_mul.restype = C.c_uint64
        L.orc_gl_mul.argtypes = [C.c_uint64, C.c_uint64]
        L.orc_gl_inv.restype = C.c_uint64
        L.orc_gl_inv.argtypes = [C.c_uint64]
        L.orc_gl_root.restype = C.c_uint64
        L.orc_gl_root.argtypes = [C.c_uint32]
        L.orc_batch_from_values.restype = C.c_void_p
        L.orc_batch_from_coeffs.restype = C.c_void_p
        for f in ("orc_batch_cap", "orc_batch_leaves", "orc_batch_coeffs", "orc_batch_digests"):
            getattr(L, f).restype = u64p
            getattr(L, f).argtypes = [C.c_void_p]
        L.orc_batch_free.argtypes = [C.c_void_p]
        L.orc_batch_prove.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        L.orc_batch_eval_at.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_proof_size_words.restype = C.c_size_t
        L.orc_challenger_get.restype = C.c_uint64
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


def poseidon(state):
    s = u64(state).copy()
    lib().orc_poseidon(_p(s))
    return s


def poseidon_batch(states):
    s = u64(states).copy()
    lib().orc_poseidon_batch(_p(s), C.c_size_t(s.size // 12))
    return s


def hash_rows(rows):
    rows = u64(rows)
    out = np.empty((rows.shape[0], 4), dtype=np.uint64)
    lib().orc_hash_rows(_p(rows), C.c_size_t(rows.shape[0]), C.c_uint32(rows.shape[1]), _p(out))
    return out


def hash_no_pad(xs):
    """PoseidonHash::hash_no_pad of one vector (always the sponge: no hash_or_noop copy branch)"""
    xs = u64(xs)
    out = np.empty(4, dtype=np.uint64)
    lib().orc_hash_no_pad(_p(xs), C.c_size_t(xs.size), _p(out))
    return out


def two_to_one(l, r):
    l, r = u64(l), u64(r)
    out = np.empty_like(l)
    for i in range(l.reshape(-1, 4).shape[0]):
        lib().orc_two_to_one(_p(l.reshape(-1, 4)[i]), _p(r.reshape(-1, 4)[i]), C.c_void_p(out.ctypes.data + 32 * i))
    return out


def fft(cols, inverse=False, shift=0):
    """cols: (n_cols, n) -- natural order in/out."""
    a = u64(cols).copy()
    a2 = a.reshape(-1, a.shape[-1])
    log_n = a2.shape[1].bit_length() - 1
    lib().orc_fft(_p(a2), C.c_uint32(log_n), C.c_uint32(a2.shape[0]), C.c_int32(1 if inverse else 0), C.c_uint64(shift))
    return a


def merkle_build(leaves, cap_height):
    leaves = u64(leaves)
    L, ll = leaves.shape
    cap = np.empty((1 << cap_height, 4), dtype=np.uint64)
    dig = np.empty((max(1, 2 * (L - (1 << cap_height))), 4), dtype=np.uint64)
    rc = lib().orc_merkle_build(_p(leaves), C.c_size_t(L), C.c_uint32(ll), C.c_uint32(cap_height), _p(cap), _p(dig))
    assert rc == 0
    return cap, dig[:2 * (L - (1 << cap_height))]


def merkle_prove(leaves, cap_height, idx):
    leaves = u64(leaves)
    L, ll = leaves.shape
    ns = (L.bit_length() - 1) - cap_height
    sib = np.empty((max(ns, 1), 4), dtype=np.uint64)
    rc = lib().orc_merkle_prove(_p(leaves), C.c_size_t(L), C.c_uint32(ll), C.c_uint32(cap_height), C.c_uint32(idx), _p(sib))
    assert rc == 0
    return sib[:ns]


def merkle_verify(leaf, idx, sib, cap):
    leaf, sib, cap = u64(leaf), u64(sib), u64(cap)
    return lib().orc_merkle_verify(_p(leaf), C.c_uint32(leaf.size), C.c_uint32(idx), _p(sib), C.c_uint32(sib.size // 4), _p(cap)) == 0


class Challenger:
    class _S(C.Structure):
        _fields_ = [("state", C.c_uint64 * 12), ("inbuf", C.c_uint64 * 8), ("outbuf", C.c_uint64 * 8),
                    ("n_in", C.c_uint32), ("n_out", C.c_uint32)]

    def __init__(self):
        self.s = self._S()
        lib().orc_challenger_init(C.byref(self.s))

    def observe(self, xs):
        xs = u64(np.atleast_1d(xs)).ravel()
        lib().orc_challenger_observe(C.byref(self.s), _p(xs), C.c_uint32(xs.size))

    def get(self, n=1):
        out = np.empty(n, dtype=np.uint64)
        lib().orc_challenger_get_n(C.byref(self.s), C.c_uint32(n), _p(out))
        return out

    def compact(self):
        out = np.empty(12, dtype=np.uint64)
        lib().orc_challenger_compact(C.byref(self.s), _p(out))
        return out


class Batch:
    """orc_batch wrapper: PolynomialBatch::from_values / from_coeffs."""

    def __init__(self, cols, rate_bits=1, cap_height=4, from_coeffs=False):
        cols = u64(cols)
        self.n_cols, n = cols.shape
        self.log_n = n.bit_length() - 1
        self.rate_bits, self.cap_height = rate_bits, cap_height
        f = lib().orc_batch_from_coeffs if from_coeffs else lib().orc_batch_from_values
        self.h = C.c_void_p(f(_p(cols), C.c_uint32(self.log_n), C.c_uint32(self.n_cols), C.c_uint32(rate_bits), C.c_uint32(cap_height)))
        self.L = n << rate_bits

    def _arr(self, ptr, shape):
        return np.ctypeslib.as_array(ptr, shape=shape).copy()

    @property
    def cap(self):
        return self._arr(lib().orc_batch_cap(self.h), (1 << self.cap_height, 4))

    @property
    def leaves(self):
        return self._arr(lib().orc_batch_leaves(self.h), (self.L, self.n_cols))

    @property
    def coeffs(self):
        return self._arr(lib().orc_batch_coeffs(self.h), (self.n_cols, 1 << self.log_n))

    @property
    def digests(self):
        nd = 2 * (self.L - (1 << self.cap_height))
        return self._arr(lib().orc_batch_digests(self.h), (nd, 4)) if nd else np.empty((0, 4), np.uint64)

    def prove(self, idx):
        ns = self.log_n + self.rate_bits - self.cap_height
        sib = np.empty((max(ns, 1), 4), dtype=np.uint64)
        lib().orc_batch_prove(self.h, C.c_uint32(idx), _p(sib))
        return sib[:ns]

    def eval_at(self, point):
        pt = u64(point)
        out = np.empty((self.n_cols, 2), dtype=np.uint64)
        lib().orc_batch_eval_at(self.h, _p(pt), _p(out))
        return out

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_batch_free(self.h)
            self.h = None


class Table(C.Structure):
    _fields_ = [("air", C.c_void_p), ("log_n", C.c_uint32), ("reserved", C.c_uint32), ("trace", C.c_void_p),
                ("public_inputs", C.c_void_p)]


def synth_trace(n_cols, log_n, seed):
    t = np.empty((n_cols, 1 << log_n), dtype=np.uint64)
    pub = np.empty(3, dtype=np.uint64)
    lib().orc_synth_trace(C.c_uint32(n_cols), C.c_uint32(log_n), C.c_uint64(seed), _p(t), _p(pub))
    return t, pub


def aux_columns(air_prog, challenges, trace, num_challenges=2):
    trace = u64(trace)
    log_n = trace.shape[1].bit_length() - 1
    aux = np.empty((air_prog.n_aux_cols, trace.shape[1]), dtype=np.uint64)
    ch = u64(challenges)
    rc = lib().orc_aux_columns(air_prog.ref, C.c_uint32(num_challenges), _p(ch), _p(trace), C.c_uint32(log_n), _p(aux))
    return rc, aux


def _tables(items):
    """items: list of (AirProgram, log_n, trace or None, public_inputs)."""
    arr = (Table * len(items))()
    keep = []
    for i, (prog, log_n, trace, pub) in enumerate(items):
        pub = u64(pub)
        keep.append(pub)
        arr[i].air = C.cast(prog.ref, C.c_void_p)
        arr[i].log_n = log_n
        if trace is not None:
            trace = u64(trace)
            keep.append(trace)
            arr[i].trace = trace.ctypes.data
        arr[i].public_inputs = pub.ctypes.data
    return arr, keep


def prove(cfg, items):
    arr, keep = _tables(items)
    cap = lib().orc_proof_size_words(C.byref(cfg), C.c_uint32(len(items)), arr)
    out = np.zeros(cap, dtype=np.uint64)
    n = C.c_size_t(0)
    rc = lib().orc_prove(C.byref(cfg), C.c_uint32(len(items)), arr, _p(out), C.c_size_t(cap), C.byref(n))
    return rc, out[:n.value]


def verify(cfg, items, proof):
    arr, keep = _tables(items)
    proof = u64(proof)
    return lib().orc_verify(C.byref(cfg), C.c_uint32(len(items)), arr, _p(proof), C.c_size_t(proof.size))


def proof_size_words(cfg, items):
    arr, keep = _tables(items)
    return lib().orc_proof_size_words(C.byref(cfg), C.c_uint32(len(items)), arr)


def quotient(cfg, air_prog, trace_batch, aux_batch, public_inputs, challenges, alphas):
    n = 1 << trace_batch.log_n
    out = np.empty((cfg.num_challenges << cfg.rate_bits, n), dtype=np.uint64)
    pub, ch, al = u64(public_inputs), u64(challenges), u64(alphas)
    rc = lib().orc_quotient(C.byref(cfg), air_prog.ref, C.c_uint32(trace_batch.log_n), trace_batch.h,
                            aux_batch.h if aux_batch is not None else None, _p(pub), _p(ch), _p(al), _p(out))
    assert rc == 0
    return out


def pow_grind(state12, pos, min_zeros):
    st = u64(state12)
    w = C.c_uint64(0)
    rc = lib().orc_pow_grind(_p(st), C.c_uint32(pos), C.c_uint32(min_zeros), C.byref(w))
    assert rc == 0
    return w.value


def fri_prove(cfg, trace_batch, aux_batch, quot_batch, challenger, zeta, cap_words=1 << 22):
    out = np.empty(cap_words, dtype=np.uint64)
    z = u64(zeta)
    ln = C.c_size_t(0)
    rc = lib().orc_fri_prove(C.byref(cfg), trace_batch.h, aux_batch.h if aux_batch is not None else None, quot_batch.h,
                             C.byref(challenger.s), _p(z), _p(out), C.c_size_t(cap_words), C.byref(ln))
    return rc, out[:ln.value]


def _fri_batches(batches):
    class FriPoly(C.Structure):
        _fields_ = [("oracle_index", C.c_uint32), ("polynomial_index", C.c_uint32)]

    class FriBatch(C.Structure):
        _fields_ = [("point", C.c_uint64 * 2), ("n_polys", C.c_uint32), ("reserved", C.c_uint32), ("polys", C.POINTER(FriPoly))]
    arr = (FriBatch * len(batches))()
    keep = []
    for i, (pt, polys) in enumerate(batches):
        pa = (FriPoly * max(1, len(polys)))(*[FriPoly(int(o), int(k)) for o, k in polys])
        keep.append(pa)
        arr[i].point[0], arr[i].point[1] = int(pt[0]), int(pt[1])
        arr[i].n_polys = len(polys)
        arr[i].polys = C.cast(pa, C.POINTER(FriPoly))
    return arr, keep


def fri_combine(cfg, oracles, batches, alpha):
    """counterpart of zk_fri_combine: lde_final_values (L x 2, leaf order) for a given alpha"""
    arr, keep = _fri_batches(batches)
    hs = (C.c_void_p * len(oracles))(*[o.h for o in oracles])
    L = 1 << (oracles[0].log_n + cfg.rate_bits)
    out = np.empty((L, 2), dtype=np.uint64)
    al = u64(alpha)
    rc = lib().orc_fri_combine(C.byref(cfg), C.c_uint32(len(oracles)), hs, C.c_uint32(len(batches)), arr, _p(al), _p(out))
    assert rc == 0, rc
    return out


def fri_prove_instance(cfg, oracles, batches, challenger, cap_words=1 << 23):
    """counterpart of zk_fri_prove_instance: oracles = [Batch], batches = [(point_ext, [(oracle, poly)])]"""
    class FriPoly(C.Structure):
        _fields_ = [("oracle_index", C.c_uint32), ("polynomial_index", C.c_uint32)]

    class FriBatch(C.Structure):
        _fields_ = [("point", C.c_uint64 * 2), ("n_polys", C.c_uint32), ("reserved", C.c_uint32), ("polys", C.POINTER(FriPoly))]
    arr = (FriBatch * len(batches))()
    keep = []
    for i, (pt, polys) in enumerate(batches):
        pa = (FriPoly * max(1, len(polys)))(*[FriPoly(int(o), int(k)) for o, k in polys])
        keep.append(pa)
        arr[i].point[0], arr[i].point[1] = int(pt[0]), int(pt[1])
        arr[i].n_polys = len(polys)
        arr[i].polys = C.cast(pa, C.POINTER(FriPoly))
    hs = (C.c_void_p * len(oracles))(*[o.h for o in oracles])
    out = np.empty(cap_words, dtype=np.uint64)
    ln = C.c_size_t(0)
    rc = lib().orc_fri_prove_instance(C.byref(cfg), C.c_uint32(len(oracles)), hs, C.c_uint32(len(batches)), arr,
                                      C.byref(challenger.s), _p(out), C.c_size_t(cap_words), C.byref(ln))
    return rc, out[:ln.value]


# ---- Plonk-shaped proofs (oracle/plonk.c)
class PlonkState:
    """orc_plonk_state: constants_sigmas commitment + digest of one circuit"""

    def __init__(self, cfg, circuit):
        L = lib()
        L.orc_plonk_setup.restype = C.c_void_p
        self.circuit, self.cfg = circuit, cfg
        self.h = L.orc_plonk_setup(C.byref(cfg), circuit.ref)
        if not self.h:
            raise ValueError("orc_plonk_setup: malformed circuit")

    def digest(self):
        out = np.empty(4, dtype=np.uint64)
        lib().orc_plonk_digest(C.c_void_p(self.h), _p(out))
        return out

    def __del__(self):
        try:
            if self.h:
                lib().orc_plonk_state_free(C.c_void_p(self.h))
        except Exception:
            pass


def plonk_proof_size_words(cfg, circuit):
    f = lib().orc_plonk_proof_size_words
    f.restype = C.c_size_t
    return f(C.byref(cfg), circuit.ref)


def prove_plonk(state, wires, public_inputs):
    wires, pub = u64(wires), u64(public_inputs)
    n = plonk_proof_size_words(state.cfg, state.circuit)
    out = np.empty(n, dtype=np.uint64)
    ln = C.c_size_t(0)
    rc = lib().orc_prove_plonk(C.byref(state.cfg), C.c_void_p(state.h), _p(wires), _p(pub), C.c_uint32(pub.size), _p(out), C.c_size_t(n), C.byref(ln))
    return rc, out[:ln.value]


def verify_plonk(state, public_inputs, proof):
    pub, proof = u64(public_inputs), u64(proof)
    return lib().orc_verify_plonk(C.byref(state.cfg), C.c_void_p(state.h), _p(pub), C.c_uint32(pub.size), _p(proof), C.c_size_t(proof.size))


# ---- AllRecursiveCircuits::prove_root over stand-in circuits (checker of eth_tx_proof_amd.ops.prove_root)
_plonk_states = {}


def _plonk_state_for(rcfg, circ):
    k = id(circ)
    if k not in _plonk_states:
        _plonk_states[k] = (circ, PlonkState(rcfg, circ))
    return _plonk_states[k][1]


def words_digest(words):
    """the binding digest of a child proof (restated from plonk.SyntheticRecursion.digest): Merkle root over leaves of eight words"""
    w = u64(words).ravel()
    n_leaves = max(2, 1 << (max(1, -(-w.size // 8)) - 1).bit_length())
    leaves = np.zeros((n_leaves, 8), dtype=np.uint64)
    leaves.ravel()[:w.size] = w
    return merkle_build(leaves, 0)[0][0]


def _prove_bound(rcfg, circ, wires, pubs):
    """one stand-in circuit: the PublicInputGate cells (row 0 of circ.pi_wires) carry hash_no_pad(public inputs)"""
    pubs = u64(pubs)
    w = np.array(wires, dtype=np.uint64, copy=True)
    pih = hash_no_pad(pubs)
    for k, col in enumerate(circ.pi_wires):
        w[col, 0] = pih[k]
    st = _plonk_state_for(rcfg, circ)
    rc, p = prove_plonk(st, w, pubs)
    assert rc == 0, rc
    assert verify_plonk(st, pubs, p) == 0
    return p


def prove_root(cfg, rcfg, items, rec, steps=None):
    """The oracle's restatement of prove_root: STARK proofs of all tables (orc.prove), then per table the initial wrapper and the
    shrinking wrappers, then the root proof.  `rec` supplies DATA only -- which circuits exist (rec.shrink_chain) and each circuit with
    its witness (rec.circuit) -- the flow and the binding rule (public inputs = table index, size and Poseidon digests of the child
    proofs' words) are restated here.  Returns (root proof, all_proof, [shrunk proofs]); `steps` (a list) receives every proof of every
    chain as (table, step, degree_bits, proof)."""
    rc, all_proof = prove(cfg, items)
    assert rc == 0
    shrunk, at = [], 2
    for ti, it in enumerate(items):
        n = proof_size_words(cfg, [it]) - 2
        words = all_proof[at:at + n]
        at += n
        chain = rec.shrink_chain(ti, it[0], it[1])
        kind, key, bits = chain[0]
        circ, wires = rec.circuit(kind, key, bits)
        proof = _prove_bound(rcfg, circ, wires, np.concatenate([np.array([ti, it[1]], dtype=np.uint64), words_digest(words)]))
        if steps is not None:
            steps.append((ti, 0, bits, proof))
        for si, (kind, key, bits) in enumerate(chain[1:], 1):
            circ, wires = rec.circuit(kind, key, bits)
            proof = _prove_bound(rcfg, circ, wires, words_digest(proof))
            if steps is not None:
                steps.append((ti, si, bits, proof))
        shrunk.append(proof)
    assert at == all_proof.size
    circ, wires = rec.circuit("root")
    pubs = np.concatenate([np.array([it[1] for it in items], dtype=np.uint64)] + [words_digest(p) for p in shrunk])
    return _prove_bound(rcfg, circ, wires, pubs), all_proof, shrunk


def prove_block(cfg, rcfg, units, rec, height=1, prev=None):
    """The whole job of /root/reference/leader/src/prover.rs:20-43 through the oracle: map(prove_root) over the units, balanced
    pairwise fold with the aggregation circuit, block circuit.  -> (block proof, its public inputs, [root proofs])."""
    roots = [prove_root(cfg, rcfg, items, rec)[0] for items in units]

    def combine(a, b):
        circ, wires = rec.circuit("agg")
        return _prove_bound(rcfg, circ, wires, np.concatenate([words_digest(a), words_digest(b)]))
    level = list(roots)
    while len(level) > 1:
        nxt = [combine(level[j], level[j + 1]) for j in range(0, len(level) - 1, 2)]
        if len(level) & 1:
            nxt.append(level[-1])
        level = nxt
    circ, wires = rec.circuit("block")
    pubs = np.concatenate([np.array([height], dtype=np.uint64), words_digest(level[0]),
                           words_digest(prev) if prev is not None else np.zeros(4, dtype=np.uint64)])
    return _prove_bound(rcfg, circ, wires, pubs), pubs, roots
