"""Goldilocks arithmetic on numpy uint64 arrays (p = 2^64 - 2^32 + 1), canonical in / canonical out.

Host-side helper of the stand-in witness generators (plonk.py, gates.py): upstream's witness generators run on the CPU too
(plonky2::iop::generator; what /root/reference/ops/src/lib.rs:52 does before every Plonk proof), and the circuits' witnesses are inputs
of the prover, not part of the hot path.  Products go through four 32 x 32 -> 64 bit partial products, reduction as in
plonky2_field::goldilocks_field::reduce128 (2^64 = 2^32 - 1, 2^96 = -1 mod p)."""
import numpy as np

P_INT = 0xFFFFFFFF00000001
P = np.uint64(P_INT)
EPS = np.uint64(0xFFFFFFFF)
_M32 = np.uint64(0xFFFFFFFF)
_S32 = np.uint64(32)


def arr(x):
    return np.asarray(x, dtype=np.uint64)


def add(a, b):
    a, b = arr(a), arr(b)
    with np.errstate(over="ignore"):
        s = a + b
        return np.where(s < a, s + EPS, np.where(s >= P, s - P, s))


def sub(a, b):
    a, b = arr(a), arr(b)
    with np.errstate(over="ignore"):
        r = a - b
        return np.where(a < b, r + P, r)


def mul(a, b):
    a, b = arr(a), arr(b)
    with np.errstate(over="ignore"):
        a0, a1, b0, b1 = a & _M32, a >> _S32, b & _M32, b >> _S32
        p00, p01, p10, p11 = a0 * b0, a0 * b1, a1 * b0, a1 * b1
        mid = p01 + p10
        midc = (mid < p01).astype(np.uint64)
        lo = p00 + (mid << _S32)
        c1 = (lo < p00).astype(np.uint64)
        hi = p11 + (mid >> _S32) + (midc << _S32) + c1
        h0, h1 = hi & _M32, hi >> _S32
        t0 = lo - h1
        t0 = np.where(lo < h1, t0 - EPS, t0)
        t1 = h0 * EPS
        r = t0 + t1
        r = np.where(r < t0, r + EPS, r)
        return np.where(r >= P, r - P, r)


def pow7(x):
    x2 = mul(x, x)
    x3 = mul(x2, x)
    return mul(mul(x2, x2), x3)


def scalar(v, like):
    """a python int as an array shaped like `like`"""
    return np.full(np.shape(like), int(v) % P_INT, dtype=np.uint64)


def rand(rng, shape):
    """uniform canonical field elements"""
    a = rng.integers(0, 1 << 63, size=shape, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=shape, dtype=np.uint64)
    return np.where(a >= P, a - P, a)


def powers(base, n):
    """[base^0 .. base^(n-1)] by doubling"""
    out = np.ones(n, dtype=np.uint64)
    have, b = 1, int(base) % P_INT
    while have < n:
        step = pow(b, have, P_INT)
        m = min(have, n - have)
        out[have:have + m] = mul(out[:m], scalar(step, out[:m]))
        have += m
    return out
