"""ctypes binding of the product library `csrc/libzkgl_hip.so` (C ABI: include/zkgl.h).

No CPU fallback: a missing library raises at import of a symbol, a missing GPU raises ZkError(ZK_ERR_NO_DEVICE)
from Context().  Host arrays are numpy uint64; device pointers are plain ints (e.g. torch_tensor.data_ptr()).
"""
import ctypes as C
import os
import re
import subprocess

import numpy as np

from ._abi import ERRORS, StarkConfig

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
_SO = os.environ.get("ZK_LIB_PATH") or os.path.join(_CSRC, "libzkgl_hip.so")      # ZK_LIB_PATH: A/B experiments with another build of the library
_HEADER = os.path.join(os.path.dirname(_CSRC), "..", "include", "zkgl.h")
_lib = None


class ZkError(RuntimeError):
    def __init__(self, code, what="", detail=""):
        self.code = code
        super().__init__("%s failed: %s (%d) %s" % (what, ERRORS.get(code, "?"), code, detail))


def build(force=False, jobs=8):
    """Compile every HIP translation unit for gfx950 (hipcc cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(["make", "-C", _CSRC, "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", _CSRC, "-j%d" % jobs, "libzkgl_hip.so"], stdout=subprocess.DEVNULL)
    return _SO


def declared_symbols():
    """Every exported function name declared in include/zkgl.h (static inline helpers defined in the header are not exports)."""
    txt = open(_HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    inline = set(re.findall(r"static\s+inline\s+[a-z0-9_]+\s+(zk_[a-z0-9_]+)\s*\(", txt))
    return sorted(set(re.findall(r"\b(zk_[a-z0-9_]+)\s*\(", txt)) - inline)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            raise FileNotFoundError("%s not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                                    "(there is no CPU fallback)" % _SO)
        L = C.CDLL(_SO)
        L.zk_last_error.restype = C.c_char_p
        L.zk_ctx_stream.restype = C.c_void_p
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


class Context:
    """zk_ctx: one device + one HIP stream (one per worker thread, like a paladin worker)."""

    def __init__(self, device=0, stream=None):
        self.h = C.c_void_p()
        self.device = device
        rc = lib().zk_ctx_create(C.c_int32(device), C.c_void_p(stream), C.byref(self.h))
        if rc:
            self.h = None
            raise ZkError(rc, "zk_ctx_create")

    def upload_packed(self, columns):
        """zk_trace_upload_packed: `columns` = list of 1-D numpy arrays (uint8 / uint16 / uint32 / uint64, one per trace column, equal
        lengths 2^k) -> device pointer of the widened n_cols x n uint64 trace (zk_dev_free / dev_free it after the proof)"""
        n = len(columns[0])
        log_n = n.bit_length() - 1
        assert n == 1 << log_n and all(len(c) == n for c in columns)
        widths = np.array([np.dtype(c.dtype).itemsize for c in columns], dtype=np.uint8)
        buf, at = bytearray(), 0
        for c, w in zip(columns, widths):
            w = int(w)
            pad = (-at) % w
            buf += b"\0" * pad + np.ascontiguousarray(c).tobytes()
            at += pad + n * w
        raw = np.frombuffer(bytes(buf), dtype=np.uint8)
        p = C.c_void_p()
        self._chk(lib().zk_trace_upload_packed(self.h, raw.ctypes.data_as(C.c_void_p), C.c_uint32(log_n), C.c_uint32(len(columns)),
                                                widths.ctypes.data_as(C.POINTER(C.c_uint8)), C.byref(p)), "zk_trace_upload_packed")
        self.synchronize()               # `raw` is a temporary: the (pageable) source must outlive the staged copy
        return p.value

    def set_side_commit(self, on):
        """zk_ctx_set_side_commit: the widest table's trace commitment of a multi-table proof on a second stream (latency switch)"""
        self._chk(lib().zk_ctx_set_side_commit(self.h, C.c_int32(1 if on else 0)), "zk_ctx_set_side_commit")

    def trim(self):
        """zk_ctx_trim: cached device memory (free list, NTT tables of tall transforms) back to the driver; returns the bytes freed"""
        n = C.c_uint64(0)
        self._chk(lib().zk_ctx_trim(self.h, C.byref(n)), "zk_ctx_trim")
        return n.value

    def set_host_transcript(self, on):
        """zk_ctx_set_host_transcript: the Fiat-Shamir sponge of this context's proofs on a host thread (latency switch; same proof words)"""
        self._chk(lib().zk_ctx_set_host_transcript(self.h, C.c_int32(1 if on else 0)), "zk_ctx_set_host_transcript")

    def _chk(self, rc, what):
        if rc:
            raise ZkError(rc, what, lib().zk_last_error(self.h).decode())

    def close(self):
        if self.h:
            lib().zk_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        self._chk(lib().zk_ctx_synchronize(self.h), "zk_ctx_synchronize")

    @property
    def stream(self):
        return lib().zk_ctx_stream(self.h)

    # ---- profiling
    def profile(self, on=True):
        self._chk(lib().zk_profile_enable(self.h, C.c_int32(1 if on else 0)), "zk_profile_enable")

    def profile_reset(self):
        self._chk(lib().zk_profile_reset(self.h), "zk_profile_reset")

    def profile_report(self):
        n = lib().zk_profile_count(self.h)
        out = {}
        for i in range(n):
            name, launches, ms = C.c_char_p(), C.c_uint64(), C.c_double()
            lib().zk_profile_get(self.h, C.c_int32(i), C.byref(name), C.byref(launches), C.byref(ms))
            out[name.value.decode()] = (launches.value, ms.value)
        return out

    # ---- device memory
    def dev_alloc(self, nbytes):
        p = C.c_void_p()
        self._chk(lib().zk_dev_alloc(self.h, C.c_size_t(nbytes), C.byref(p)), "zk_dev_alloc")
        return p.value

    def dev_free(self, p):
        lib().zk_dev_free(self.h, C.c_void_p(p))

    def upload(self, dptr, arr):
        arr = np.ascontiguousarray(arr)
        self._chk(lib().zk_dev_upload(self.h, C.c_void_p(dptr), _p(arr), C.c_size_t(arr.nbytes)), "zk_dev_upload")

    def download(self, dptr, shape, dtype=np.uint64):
        out = np.empty(shape, dtype=dtype)
        self._chk(lib().zk_dev_download(self.h, _p(out), C.c_void_p(dptr), C.c_size_t(out.nbytes)), "zk_dev_download")
        return out

    # ---- field / NTT  (fft_with_options / ifft_with_options / coset_fft / coset_ifft)
    def ntt(self, cols, inverse=False, shift=0, zero_factor=0):
        """fft_with_options / ifft_with_options / coset_fft_with_options; zero_factor = r: only the first n >> r inputs count"""
        a = _u64(cols).copy()
        a2 = a.reshape(-1, a.shape[-1])
        log_n = a2.shape[1].bit_length() - 1
        assert 1 << log_n == a2.shape[1]
        self._chk(lib().zk_ntt_batch_opts(self.h, _p(a2), C.c_uint32(log_n), C.c_uint32(a2.shape[0]),
                                          C.c_int32(1 if inverse else 0), C.c_uint64(shift), C.c_uint32(zero_factor)), "zk_ntt_batch_opts")
        return a

    def ntt_dev(self, dptr, log_n, n_cols, inverse=False, shift=0):
        self._chk(lib().zk_ntt_batch_dev(self.h, C.c_void_p(dptr), C.c_uint32(log_n), C.c_uint32(n_cols),
                                         C.c_int32(1 if inverse else 0), C.c_uint64(shift)), "zk_ntt_batch_dev")

    # ---- Poseidon
    def poseidon(self, states):
        s = _u64(states).copy()
        self._chk(lib().zk_poseidon_permute_batch(self.h, _p(s), C.c_size_t(s.size // 12)), "zk_poseidon_permute_batch")
        return s

    def hash_rows(self, rows):
        rows = _u64(rows)
        out = np.empty((rows.shape[0], 4), dtype=np.uint64)
        self._chk(lib().zk_hash_no_pad_batch(self.h, _p(rows), C.c_size_t(rows.shape[0]), C.c_uint32(rows.shape[1]), _p(out)),
                  "zk_hash_no_pad_batch")
        return out

    def two_to_one(self, left, right):
        left, right = _u64(left), _u64(right)
        out = np.empty_like(left)
        self._chk(lib().zk_two_to_one_batch(self.h, _p(left), _p(right), C.c_size_t(left.size // 4), _p(out)), "zk_two_to_one_batch")
        return out

    def merkle_build(self, leaves, cap_height, want_digests=True):
        leaves = _u64(leaves)
        L, ll = leaves.shape
        cap = np.empty((1 << cap_height, 4), dtype=np.uint64)
        nd = 2 * (L - (1 << cap_height))
        dig = np.empty((max(nd, 1), 4), dtype=np.uint64)
        self._chk(lib().zk_merkle_build(self.h, _p(leaves), C.c_size_t(L), C.c_uint32(ll), C.c_uint32(cap_height), _p(cap),
                                        _p(dig) if want_digests else None), "zk_merkle_build")
        return cap, dig[:nd]

    # ---- PolynomialBatch
    def batch_from_values(self, cols, rate_bits=1, cap_height=4):
        return Batch(self, cols, rate_bits, cap_height, coeffs=False)

    def batch_from_coeffs(self, cols, rate_bits=1, cap_height=4):
        return Batch(self, cols, rate_bits, cap_height, coeffs=True)

    def batch_from_values_dev(self, dptr, log_n, n_cols, rate_bits=1, cap_height=4):
        return Batch(self, None, rate_bits, cap_height, coeffs=False, dev=(dptr, log_n, n_cols))

    def challenger(self):
        return Challenger(self)


def pinned_empty(ctx, shape, dtype=np.uint64):
    """numpy array in pinned host memory (zk_host_alloc): traces held here are uploaded by zk_prove_tables with one asynchronous
    copy at PCIe rate.  The memory is freed when the array (and every view of it) is gone."""
    import weakref
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    p = C.c_void_p()
    ctx._chk(lib().zk_host_alloc(ctx.h, C.c_size_t(n), C.byref(p)), "zk_host_alloc")
    buf = (C.c_uint8 * max(n, 1)).from_address(p.value)
    arr = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)
    L, addr = lib(), p.value

    def _free():
        if ctx.h:                                   # a closed context has nothing to free through: the runtime reclaims at exit
            L.zk_host_free(ctx.h, C.c_void_p(addr))
    weakref.finalize(buf, _free)
    return arr


def trace_upload_async(ctx, trace):
    """zk_trace_upload_async: start uploading a host trace on the context's copy stream; returns the device pointer (int)."""
    trace = np.ascontiguousarray(trace, dtype=np.uint64) if not (isinstance(trace, np.ndarray) and trace.dtype == np.uint64 and trace.flags.c_contiguous) else trace
    d = C.c_void_p()
    ctx._chk(lib().zk_trace_upload_async(ctx.h, trace.ctypes.data_as(C.c_void_p), C.c_size_t(trace.size), C.byref(d)), "zk_trace_upload_async")
    return d.value


class Batch:
    """zk_batch: PolynomialBatch::{from_values, from_coeffs}; device resident."""

    def __init__(self, ctx, cols, rate_bits, cap_height, coeffs, dev=None):
        self.ctx = ctx
        self.h = C.c_void_p()
        self.rate_bits, self.cap_height = rate_bits, cap_height
        cap = np.empty((1 << cap_height, 4), dtype=np.uint64)
        if dev is None:
            cols = _u64(cols)
            self.n_cols, n = cols.shape
            self.log_n = n.bit_length() - 1
            assert 1 << self.log_n == n
            f = lib().zk_batch_from_coeffs if coeffs else lib().zk_batch_from_values
            rc = f(ctx.h, _p(cols), C.c_uint32(self.log_n), C.c_uint32(self.n_cols), C.c_uint32(rate_bits),
                   C.c_uint32(cap_height), C.byref(self.h), _p(cap))
        else:
            dptr, self.log_n, self.n_cols = dev
            f = lib().zk_batch_from_coeffs_dev if coeffs else lib().zk_batch_from_values_dev
            rc = f(ctx.h, C.c_void_p(dptr), C.c_uint32(self.log_n), C.c_uint32(self.n_cols), C.c_uint32(rate_bits),
                   C.c_uint32(cap_height), C.byref(self.h), _p(cap))
        if rc:
            self.h = None
            ctx._chk(rc, "zk_batch_from_*")
        self.cap = cap
        self.L = 1 << (self.log_n + rate_bits)

    def free(self):
        if self.h:
            lib().zk_batch_free(self.h)
            self.h = None

    def __del__(self):
        try:
            if self.ctx.h:
                self.free()
        except Exception:
            pass

    def lde_rows(self, idx, step=1):
        idx = np.ascontiguousarray(idx, dtype=np.uint32)
        out = np.empty((idx.size, self.n_cols), dtype=np.uint64)
        self.ctx._chk(lib().zk_batch_get_lde_rows(self.h, _p(idx), C.c_uint32(idx.size), C.c_uint32(step), _p(out)), "zk_batch_get_lde_rows")
        return out

    def leaf(self, leaf_idx):
        out = np.empty(self.n_cols, dtype=np.uint64)
        self.ctx._chk(lib().zk_batch_get_leaf(self.h, C.c_uint32(leaf_idx), _p(out)), "zk_batch_get_leaf")
        return out

    def prove(self, leaf_idx):
        ns = self.log_n + self.rate_bits - self.cap_height
        out = np.empty((max(ns, 1), 4), dtype=np.uint64)
        self.ctx._chk(lib().zk_batch_merkle_prove(self.h, C.c_uint32(leaf_idx), _p(out)), "zk_batch_merkle_prove")
        return out[:ns]

    def digests(self):
        nd = 2 * (self.L - (1 << self.cap_height))
        out = np.empty((max(nd, 1), 4), dtype=np.uint64)
        self.ctx._chk(lib().zk_batch_get_digests(self.h, _p(out)), "zk_batch_get_digests")
        return out[:nd]

    def coeffs(self, col):
        out = np.empty(1 << self.log_n, dtype=np.uint64)
        self.ctx._chk(lib().zk_batch_get_coeffs(self.h, C.c_uint32(col), _p(out)), "zk_batch_get_coeffs")
        return out

    def eval_at(self, point):
        pt = _u64(point)
        out = np.empty((self.n_cols, 2), dtype=np.uint64)
        self.ctx._chk(lib().zk_batch_eval_at(self.h, _p(pt), _p(out)), "zk_batch_eval_at")
        return out


class Challenger:
    """zk_challenger: Challenger<F, PoseidonHash>, state resident in HBM."""

    def __init__(self, ctx):
        self.ctx = ctx
        self.h = C.c_void_p()
        ctx._chk(lib().zk_challenger_new(ctx.h, C.byref(self.h)), "zk_challenger_new")

    def observe(self, xs):
        xs = _u64(np.atleast_1d(xs)).ravel()
        self.ctx._chk(lib().zk_challenger_observe(self.h, _p(xs), C.c_uint32(xs.size)), "zk_challenger_observe")

    def get(self, n=1):
        out = np.empty(n, dtype=np.uint64)
        self.ctx._chk(lib().zk_challenger_get(self.h, C.c_uint32(n), _p(out)), "zk_challenger_get")
        return out

    def compact(self):
        out = np.empty(12, dtype=np.uint64)
        self.ctx._chk(lib().zk_challenger_compact(self.h, _p(out)), "zk_challenger_compact")
        return out

    def free(self):
        if self.h and self.ctx.h:
            lib().zk_challenger_free(self.h)
        self.h = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Table(C.Structure):
    """zk_table (include/zkgl.h)"""
    _fields_ = [("air", C.c_void_p), ("log_n", C.c_uint32), ("trace_on_device", C.c_uint32), ("trace", C.c_void_p),
                ("public_inputs", C.c_void_p)]


def _tables(items):
    """items: (AirProgram, log_n, trace, public_inputs); trace = numpy (n_cols, n) host array, or an int device pointer."""
    arr = (Table * len(items))()
    keep = []
    for i, (prog, log_n, trace, pub) in enumerate(items):
        pub = _u64(pub)
        keep.append(pub)
        arr[i].air = C.cast(prog.ref, C.c_void_p)
        arr[i].log_n = log_n
        if isinstance(trace, (int, np.integer)):
            arr[i].trace_on_device = 1
            arr[i].trace = int(trace)
        else:
            if not (isinstance(trace, np.ndarray) and trace.dtype == np.uint64 and trace.flags.c_contiguous):
                trace = _u64(trace)                 # (a pinned array passes through untouched: same address, no copy)
            assert trace.shape == (prog.n_trace_cols, 1 << log_n)
            keep.append(trace)
            arr[i].trace_on_device = 0
            arr[i].trace = trace.ctypes.data
        arr[i].public_inputs = pub.ctypes.data
    return arr, keep


def proof_size_words(cfg, items):
    arr, keep = _tables(items)
    f = lib().zk_proof_size_words
    f.restype = C.c_size_t
    return f(C.byref(cfg), C.c_uint32(len(items)), arr)


_TABLE_READY_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint64), C.c_size_t)


def prove_tables(ctx, cfg, items, out=None, on_table=None):
    """zk_prove_tables: the STARK proofs of all tables of one unit of work; returns the proof words (numpy uint64).
    on_table(t, words): zk_prove_tables_notify -- called (on this thread, tables in order) as soon as table t's StarkProof words are on
    the host, while the later tables are still being proved; `words` is a copy.  An exception it raises is re-raised after the call."""
    arr, keep = _tables(items)
    n = proof_size_words(cfg, items)
    if out is None:
        out = np.empty(n, dtype=np.uint64)
    ln = C.c_size_t(0)
    if on_table is None:
        rc = lib().zk_prove_tables(ctx.h, C.byref(cfg), C.c_uint32(len(items)), arr, _p(out), C.c_size_t(out.size), C.byref(ln))
        ctx._chk(rc, "zk_prove_tables")
        return out[:ln.value]
    raised = []

    def cb(_user, t, words, n_words):
        try:
            on_table(int(t), np.ctypeslib.as_array(words, shape=(int(n_words),)).copy())
        except BaseException as e:       # a ctypes callback cannot propagate: kept for the caller
            raised.append(e)
    fn = _TABLE_READY_FN(cb)
    rc = lib().zk_prove_tables_notify(ctx.h, C.byref(cfg), C.c_uint32(len(items)), arr, _p(out), C.c_size_t(out.size), C.byref(ln), fn, None)
    ctx._chk(rc, "zk_prove_tables_notify")
    if raised:
        raise raised[0]
    return out[:ln.value]


def prover_state_prepare(ctx, cfg, items):
    """zk_prover_state_prepare: compile the constraint kernels of the tables ahead of time (call from the main thread)."""
    arr, keep = _tables(items)
    ctx._chk(lib().zk_prover_state_prepare(ctx.h, C.byref(cfg), C.c_uint32(len(items)), arr), "zk_prover_state_prepare")


def synth_trace_dev(ctx, n_cols, log_n, seed, dptr=None):
    """zk_synth_trace_dev: returns (device pointer, public inputs)."""
    if dptr is None:
        dptr = ctx.dev_alloc(8 * n_cols << log_n)
    pub = np.empty(3, dtype=np.uint64)
    ctx._chk(lib().zk_synth_trace_dev(ctx.h, C.c_uint32(n_cols), C.c_uint32(log_n), C.c_uint64(seed), C.c_void_p(dptr), _p(pub)),
             "zk_synth_trace_dev")
    return dptr, pub


# ---- the pieces of a unit proof (zk_aux_columns, zk_quotient, zk_pow_grind, zk_fri_prove_openings)
def aux_columns(ctx, prog, challenges, trace, num_challenges=2):
    trace = _u64(trace)
    log_n = trace.shape[1].bit_length() - 1
    ch = _u64(challenges)
    out = np.empty((prog.n_aux_cols, trace.shape[1]), dtype=np.uint64)
    ctx._chk(lib().zk_aux_columns(ctx.h, prog.ref, C.c_uint32(num_challenges), _p(ch), _p(trace), C.c_uint32(log_n), _p(out)),
             "zk_aux_columns")
    return out


def quotient(ctx, cfg, prog, trace_batch, aux_batch, public_inputs, challenges, alphas):
    n = 1 << trace_batch.log_n
    out = np.empty((cfg.num_challenges << cfg.rate_bits, n), dtype=np.uint64)
    pub, ch, al = _u64(public_inputs), _u64(challenges), _u64(alphas)
    ctx._chk(lib().zk_quotient(ctx.h, C.byref(cfg), prog.ref, trace_batch.h, aux_batch.h if aux_batch is not None else None,
                               _p(pub), _p(ch), _p(al), _p(out)), "zk_quotient")
    return out


def pow_grind(ctx, state12, pos, min_zeros):
    st = _u64(state12)
    w = C.c_uint64(0)
    ctx._chk(lib().zk_pow_grind(ctx.h, _p(st), C.c_uint32(pos), C.c_uint32(min_zeros), C.byref(w)), "zk_pow_grind")
    return w.value


def fri_prove_openings(ctx, cfg, trace_batch, aux_batch, quot_batch, challenger, zeta):
    f = lib().zk_fri_proof_size_words
    f.restype = C.c_size_t
    n = f(C.byref(cfg), C.c_uint32(trace_batch.log_n), C.c_uint32(trace_batch.n_cols),
          C.c_uint32(aux_batch.n_cols if aux_batch is not None else 0))
    out = np.empty(n, dtype=np.uint64)
    z = _u64(zeta)
    ln = C.c_size_t(0)
    ctx._chk(lib().zk_fri_prove_openings(ctx.h, C.byref(cfg), trace_batch.h, aux_batch.h if aux_batch is not None else None,
                                         quot_batch.h, challenger.h, _p(z), _p(out), C.c_size_t(n), C.byref(ln)),
             "zk_fri_prove_openings")
    return out[:ln.value]


def fri_prove_instance(ctx, cfg, oracles, batches, challenger):
    """PolynomialBatch::prove_openings for any FriInstanceInfo: oracles = [Batch], batches = [(point_ext, [(oracle, poly)])]"""
    from . import _abi
    f = lib().zk_fri_instance_proof_size_words
    f.restype = C.c_size_t
    ncols = (C.c_uint32 * len(oracles))(*[o.n_cols for o in oracles])
    n = f(C.byref(cfg), C.c_uint32(oracles[0].log_n), C.c_uint32(len(oracles)), ncols)
    out = np.empty(n, dtype=np.uint64)
    hs = (C.c_void_p * len(oracles))(*[o.h for o in oracles])
    arr, keep = _abi.fri_instance(batches)
    ln = C.c_size_t(0)
    ctx._chk(lib().zk_fri_prove_instance(ctx.h, C.byref(cfg), C.c_uint32(len(oracles)), hs, C.c_uint32(len(batches)), arr,
                                         challenger.h, _p(out), C.c_size_t(n), C.byref(ln)), "zk_fri_prove_instance")
    del keep
    return out[:ln.value]


def prover_state_cache_dir(path):
    """zk_prover_state_cache_dir: directory for the on-disk prover state (compiled constraint kernels); None switches it off."""
    lib().zk_prover_state_cache_dir(None if path is None else str(path).encode())


MI355X_ARCH_NAME = "gfx950:sramecc+:xnack-"          # hipDeviceProp_t::gcnArchName on MI355X: part of a prover-state file's key


def prover_state_build_offline(cfg, progs, arch_name=MI355X_ARCH_NAME):
    """zk_prover_state_build_offline: compile the constraint kernels of these AIR programs into the prover-state directory without a GPU;
    returns (files written, files already there)"""
    arr = (Table * len(progs))()
    for i, p in enumerate(progs):
        arr[i].air = C.cast(p.ref, C.c_void_p)
    a, b = C.c_uint64(0), C.c_uint64(0)
    rc = lib().zk_prover_state_build_offline(C.byref(cfg), C.c_uint32(len(progs)), arr, arch_name.encode(), C.byref(a), C.byref(b))
    if rc != 0:
        raise RuntimeError("zk_prover_state_build_offline failed with %d" % rc)
    return a.value, b.value


def prover_state_stats():
    """(kernels compiled by hiprtc, kernels loaded from disk) in this process"""
    a, b = C.c_uint64(0), C.c_uint64(0)
    lib().zk_prover_state_stats(C.byref(a), C.byref(b))
    return a.value, b.value


# ---- Plonk-shaped proofs (zk_plonk_setup / zk_prove_plonk)
class PlonkState:
    """zk_plonk_state: the prover data of one circuit (constants_sigmas commitment, digest), device resident."""

    def __init__(self, ctx, cfg, circuit):
        self.ctx, self.cfg, self.circuit = ctx, cfg, circuit
        self.h = C.c_void_p()
        ctx._chk(lib().zk_plonk_setup(ctx.h, C.byref(cfg), circuit.ref, C.byref(self.h)), "zk_plonk_setup")

    def digest(self):
        out = np.empty(4, dtype=np.uint64)
        self.ctx._chk(lib().zk_plonk_circuit_digest(self.h, _p(out)), "zk_plonk_circuit_digest")
        return out

    def free(self):
        if self.h:
            lib().zk_plonk_state_free(self.h)
            self.h = None

    def __del__(self):
        try:
            if self.ctx.h:
                self.free()
        except Exception:
            pass


def plonk_proof_size_words(cfg, circuit):
    f = lib().zk_plonk_proof_size_words
    f.restype = C.c_size_t
    return f(C.byref(cfg), circuit.ref)


def prove_plonk(state, wires, public_inputs, ctx=None):
    """zk_prove_plonk: wires = (num_wires, n) uint64 host array or an int device pointer (canonical values).  ctx: the context
    (stream) to prove on -- any context of the device the circuit's prover data lives on; default: the one it was built with."""
    ctx = ctx or state.ctx
    pub = _u64(public_inputs)
    n = plonk_proof_size_words(state.cfg, state.circuit)
    out = np.empty(n, dtype=np.uint64)
    ln = C.c_size_t(0)
    if isinstance(wires, (int, np.integer)):
        wp, on_dev, keep = C.c_void_p(int(wires)), 1, None
    else:
        keep = wires if (isinstance(wires, np.ndarray) and wires.dtype == np.uint64 and wires.flags.c_contiguous) else _u64(wires)
        wp, on_dev = keep.ctypes.data_as(C.c_void_p), 0
    rc = lib().zk_prove_plonk(ctx.h, C.byref(state.cfg), state.h, wp, C.c_uint32(on_dev), _p(pub), C.c_uint32(pub.size), _p(out),
                              C.c_size_t(n), C.byref(ln))
    ctx._chk(rc, "zk_prove_plonk")
    return out[:ln.value]


# ---- the FRI prover piece by piece (zk_fri_combine, zk_fri_commit_*, zk_fri_fold_commit_layer, zk_fri_query)
def fri_combine(ctx, cfg, oracles, batches, alpha):
    """zk_fri_combine: the final values (L x 2 uint64, leaf order) of the batch combination for a given alpha"""
    from . import _abi
    hs = (C.c_void_p * len(oracles))(*[o.h for o in oracles])
    arr, keep = _abi.fri_instance(batches)
    L = 1 << (oracles[0].log_n + cfg.rate_bits)
    out = np.empty((L, 2), dtype=np.uint64)
    al = _u64(alpha)
    ctx._chk(lib().zk_fri_combine(ctx.h, C.byref(cfg), C.c_uint32(len(oracles)), hs, C.c_uint32(len(batches)), arr, _p(al), _p(out)), "zk_fri_combine")
    del keep
    return out


class FriCommit:
    """zk_fri_commit: the commit phase of one FRI proof, one reduction layer per call"""

    def __init__(self, ctx, cfg, log_n, final_values):
        self.ctx, self.cfg, self.log_n = ctx, cfg, log_n
        self.h = C.c_void_p()
        fv = _u64(final_values)
        ctx._chk(lib().zk_fri_commit_begin(ctx.h, C.byref(cfg), C.c_uint32(log_n), _p(fv), C.byref(self.h)), "zk_fri_commit_begin")

    def layers(self):
        a, b = C.c_uint32(0), C.c_uint32(0)
        lib().zk_fri_commit_layers(self.h, C.byref(a), C.byref(b))
        return a.value, b.value

    def fold_commit_layer(self, challenger):
        cap = np.empty((1 << self.cfg.cap_height, 4), dtype=np.uint64)
        beta = np.empty(2, dtype=np.uint64)
        self.ctx._chk(lib().zk_fri_fold_commit_layer(self.h, challenger.h, _p(cap), _p(beta)), "zk_fri_fold_commit_layer")
        return cap, beta

    def finish(self, challenger):
        out = np.empty(2 << (self.log_n + self.cfg.rate_bits), dtype=np.uint64)
        ln = C.c_uint32(0)
        self.ctx._chk(lib().zk_fri_commit_finish(self.h, challenger.h, _p(out), C.byref(ln)), "zk_fri_commit_finish")
        return out[:2 * ln.value].copy()

    def query(self, oracles, x_index):
        hs = (C.c_void_p * len(oracles))(*[o.h for o in oracles])
        f = lib().zk_fri_query_size_words
        f.restype = C.c_size_t
        n = f(self.h, C.c_uint32(len(oracles)), hs)
        out = np.empty(n, dtype=np.uint64)
        ln = C.c_size_t(0)
        self.ctx._chk(lib().zk_fri_query(self.h, C.c_uint32(len(oracles)), hs, C.c_uint32(x_index), _p(out), C.c_size_t(n), C.byref(ln)), "zk_fri_query")
        return out[:ln.value]

    def free(self):
        if self.h:
            lib().zk_fri_commit_free(self.h)
            self.h = None

    def __del__(self):
        try:
            if self.ctx.h:
                self.free()
        except Exception:
            pass
