// Goldilocks field arithmetic for gfx950 device code (and the few host-side table builders).
// F_p, p = 2^64 - 2^32 + 1; ext2 = F_p[x]/(x^2 - 7).  Mirrors plonky2_field 0.2.2 goldilocks_field.rs /
// extension/quadratic.rs (crate pinned at /root/reference/Cargo.lock:3466-3468; spec: SURVEY.md A1).
// Invariant: every value stored to memory is canonical (< p).  add/sub/mul take canonical inputs and
// return canonical outputs; gl_canon() is applied once where caller-supplied data enters.
#pragma once
#if !defined(__HIPCC_RTC__)      // hiprtc (the constraint-program JIT, prover_quotient_jit.hip) has the runtime header built in
#include <hip/hip_runtime.h>
#include <stdint.h>
#else
typedef unsigned long uint64_t;
typedef unsigned int uint32_t;
#endif

typedef uint64_t u64;
typedef uint32_t u32;

#define GL_P 0xFFFFFFFF00000001ULL
#define GL_EPS 0xFFFFFFFFULL
#define GL_GEN 14293326489335486720ULL
#define GL_POW2_GEN 7277203076849721926ULL

#define GLD __host__ __device__ __forceinline__

GLD u64 gl_canon(u64 x) { return x >= GL_P ? x - GL_P : x; }

GLD u64 gl_add(u64 a, u64 b) {
    u64 s = a + b;
    // carry: true sum = s + 2^64 = s + eps (mod p), and s + eps < p here; no carry: at most one subtraction of p, and
    // s - p == s + eps (mod 2^64) -- so both repairs are the same addition, selected without a branch
    u64 t = s + GL_EPS;
    return ((s < a) | (s >= GL_P)) ? t : s;
}
GLD u64 gl_sub(u64 a, u64 b) {
    u64 d = a - b;
    return a < b ? d - GL_EPS : d;   // borrow: d + p = d - eps (mod 2^64)
}
GLD u64 gl_neg(u64 a) { return a ? GL_P - a : 0; }
GLD u64 gl_dbl(u64 a) { return gl_add(a, a); }

// reduce a 128-bit value (hi:lo) modulo p; 2^64 = eps, 2^96 = -1.  Result canonical.
GLD u64 gl_reduce128(u64 lo, u64 hi) {
    u64 hi_hi = hi >> 32, hi_lo = hi & GL_EPS;
    u64 t0 = lo - hi_hi;
    if (lo < hi_hi) t0 -= GL_EPS;
    u64 t1 = (hi_lo << 32) - hi_lo;       // hi_lo * eps
    u64 r = t0 + t1;
    if (r < t1) r += GL_EPS;
    return r >= GL_P ? r - GL_P : r;
}
GLD u64 gl_mulhi(u64 a, u64 b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul64hi(a, b);
#else
    return (u64)(((unsigned __int128)a * b) >> 64);
#endif
}
// lo + h*eps (h a 32-bit value), repaired by +eps when the 64-bit sum wraps (it cannot wrap twice).
// v_mad_u64_u32 does the multiply-add AND delivers the wrap as an SGPR lane mask (not reachable from C++); the repair is a
// second v_mad_u64_u32 adding the zero-extended 32-bit mask.  s_nop 1: gfx950 needs two wait states between a VALU SGPR
// write and its use as a VALU operand.
__device__ __forceinline__ u64 gl_mad_eps(u32 h, u64 lo) {
#if defined(__HIP_DEVICE_COMPILE__)
    u64 r, k; u32 m;
    asm volatile(
        "v_mad_u64_u32 %[r], %[k], %[h], -1, %[lo]\n\t"
        "s_nop 1\n\t"
        "v_cndmask_b32_e64 %[m], 0, -1, %[k]\n\t"
        "v_mad_u64_u32 %[r], %[k], %[m], 1, %[r]\n\t"
        : [r] "=&v"(r), [k] "=&s"(k), [m] "=&v"(m)
        : [h] "v"(h), [lo] "v"(lo));
    return r;
#else
    return 0;   // host pass only parses device functions
#endif
}
// weak product: any representative in [0, 2^64).  Device path: four v_mad_u64_u32 whose running sums never overflow,
// then lo + h0*eps - h1 (2^64 = eps, 2^96 = -1) where each wrap is repaired by +-eps and cannot wrap twice.
GLD u64 gl_mul_weak(u64 a, u64 b) {
#if defined(__HIP_DEVICE_COMPILE__)
    u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    u64 A1 = (u64)a0 * b0;
    u64 A2 = (u64)a0 * b1 + (A1 >> 32);
    u64 A3 = (u64)a1 * b0 + (u32)A2;
    u64 hi = (u64)a1 * b1 + (A2 >> 32) + (A3 >> 32);
    u64 lo = (A3 << 32) | (u32)A1;
    u64 r1 = gl_mad_eps((u32)hi, lo);
    u64 r2; bool b1_ = __builtin_sub_overflow(r1, hi >> 32, &r2);
    r2 -= b1_ ? GL_EPS : 0ull;
    return r2;
#else
    return gl_reduce128(a * b, gl_mulhi(a, b));
#endif
}
GLD u64 gl_mul(u64 a, u64 b) { return gl_canon(gl_mul_weak(a, b)); }
GLD u64 gl_sqr(u64 a) { return gl_mul(a, a); }
// a*b + c (c canonical): one reduction
GLD u64 gl_mul_add(u64 a, u64 b, u64 c) {
#if defined(__HIP_DEVICE_COMPILE__)
    u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    u64 A1 = (u64)a0 * b0;
    u64 A2 = (u64)a0 * b1 + (A1 >> 32);
    u64 A3 = (u64)a1 * b0 + (u32)A2;
    u64 hi = (u64)a1 * b1 + (A2 >> 32) + (A3 >> 32);
    u64 lo = (A3 << 32) | (u32)A1;
    u64 l2 = lo + c;
    hi += (l2 < lo);
    u64 r1 = gl_mad_eps((u32)hi, l2);
    u64 r2; bool b1_ = __builtin_sub_overflow(r1, hi >> 32, &r2);
    r2 -= b1_ ? GL_EPS : 0ull;
    return gl_canon(r2);
#else
    u64 lo = a * b, hi = gl_mulhi(a, b);
    u64 l2 = lo + c;
    hi += (l2 < lo);
    return gl_reduce128(l2, hi);
#endif
}
GLD u64 gl_pow(u64 b, u64 e) {
    u64 r = 1;
    while (e) { if (e & 1) r = gl_mul(r, b); b = gl_mul(b, b); e >>= 1; }
    return r;
}
GLD u64 gl_inv(u64 a) { return gl_pow(a, GL_P - 2); }
GLD u64 gl_root(u32 k) { return gl_pow(GL_POW2_GEN, 1ULL << (32 - k)); }  // primitive_root_of_unity(k)

// Field product for the hot kernels.  Two of the repairs inside a product almost never fire: the borrow of "r - h1" (needs
// r < 2^32) and the final canonicalisation (needs the weak result in [p, 2^64)), each about one lane in 2^32.  glx<false>
// leaves both out -- the borrow mask of v_subbrev_co_u32 and the mask of one v_cmp_ge_u64 are OR-ed into the wave-uniform
// sticky word `fl` by SALU instructions -- and the kernel re-runs the same input through glx<true> when fl != 0, so the
// stored results are identical bit for bit.  16 instead of 23 VALU instructions per product.
// x + acc as ONE v_mad_u64_u32 (x*1 + acc): adds a 32-bit word to a 64-bit pair without first building the zero-extended
// register pair {x, 0} (one or two v_mov_b32 each time hipcc does it from the C++ spelling)
__device__ __forceinline__ u64 gl_madd1(u32 x, u64 acc) {
#if defined(__HIP_DEVICE_COMPILE__)
    u64 r, k;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %3" : "=v"(r), "=s"(k) : "v"(x), "v"(acc));
    return r;
#else
    return acc + x;
#endif
}
template <bool EXACT> struct glx {
    static constexpr bool exact = EXACT;
    u64 fl;
    // a*b = lo + hi*2^64 + cy*2^96: six v_mad_u64_u32; the mid column a0*b1 + a1*b0 keeps its carry-out as the lane mask cy
    // (2^96 = -1: it becomes the borrow-in of the last subtraction) instead of being propagated
    __device__ __forceinline__ u64 mul(u64 a, u64 b) {
        if (EXACT) return gl_mul(a, b);
        u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
        u64 P0 = (u64)a0 * b0;
        u64 A2 = gl_madd1((u32)(P0 >> 32), (u64)a0 * b1);
        u64 X, cy;
        asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(X), "=s"(cy) : "v"(a1), "v"(b0), "v"(A2));
        u64 A4 = gl_madd1((u32)(X >> 32), (u64)a1 * b1);
        return tail((X << 32) | (u32)P0, A4, cy);
    }
    // the same product as a WEAK residue (any representative below 2^64; the borrow repair is still flagged): for values that stay
    // inside a weak-arithmetic network and are checked once at its end (check_canon)
    __device__ __forceinline__ u64 mulw(u64 a, u64 b) {
        if (EXACT) return gl_mul(a, b);
        u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
        u64 P0 = (u64)a0 * b0;
        u64 A2 = gl_madd1((u32)(P0 >> 32), (u64)a0 * b1);
        u64 X, cy;
        asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(X), "=s"(cy) : "v"(a1), "v"(b0), "v"(A2));
        u64 A4 = gl_madd1((u32)(X >> 32), (u64)a1 * b1);
        u64 r = gl_mad_eps((u32)A4, (X << 32) | (u32)P0);
        u32 rl = (u32)r, rh = (u32)(r >> 32), h1 = (u32)(A4 >> 32), dl, dh;
        asm("v_subb_co_u32 %[dl], vcc, %[rl], %[h], %[cy]\n\t"
            "s_nop 1\n\t"
            "v_subbrev_co_u32 %[dh], vcc, 0, %[rh], vcc\n\t"
            "s_or_b64 %[fl], %[fl], vcc\n\t"
            : [dl] "=&v"(dl), [dh] "=&v"(dh), [fl] "+s"(fl)
            : [rl] "v"(rl), [rh] "v"(rh), [h] "v"(h1), [cy] "s"(cy) : "vcc", "scc");
        return ((u64)dh << 32) | dl;
    }
    // lo + hi*2^64 - cy  ->  canonical residue unless a rare repair was skipped (then fl != 0)
    __device__ __forceinline__ u64 tail(u64 lo, u64 hi, u64 cy) {
        u64 r = gl_mad_eps((u32)hi, lo);
        // the two masks go through vcc and straight into fl (SALU may read vcc right after the VALU write); general SGPR
        // outputs here would keep dozens of mask pairs alive across the interleaved products of a butterfly network
        u32 rl = (u32)r, rh = (u32)(r >> 32), h1 = (u32)(hi >> 32), dl, dh; u64 d;
        asm("v_subb_co_u32 %[dl], vcc, %[rl], %[h], %[cy]\n\t"
            "s_nop 1\n\t"
            "v_subbrev_co_u32 %[dh], vcc, 0, %[rh], vcc\n\t"
            "s_or_b64 %[fl], %[fl], vcc\n\t"
            : [dl] "=&v"(dl), [dh] "=&v"(dh), [fl] "+s"(fl)
            : [rl] "v"(rl), [rh] "v"(rh), [h] "v"(h1), [cy] "s"(cy) : "vcc", "scc");
        d = ((u64)dh << 32) | dl;
        asm("v_cmp_ge_u64 vcc, %[d], %[p]\n\t"
            "s_or_b64 %[fl], %[fl], vcc\n\t"
            : [fl] "+s"(fl) : [d] "v"(d), [p] "s"((u64)GL_P) : "vcc", "scc");
        return d;
    }
    // ---- weak butterflies for the NTT networks.  Operands and results are ANY representative in [0, 2^64) of their residue:
    // u + v wraps with carry c -> + c*eps (2^64 = eps); u - v wraps with borrow b -> - b*eps.  Each repair can wrap a second time only
    // for operands within 2^32 of the ends of the range (about one lane in 2^32): that second carry / borrow goes into the sticky
    // word and the kernel redoes the tile exactly.  Two butterflies per call: their four carry chains are interleaved so that every
    // scalar carry mask is read at least three instructions after it is written (gfx950 needs two wait states between a vector write
    // of an SGPR and a vector read of it; inline asm is not padded by the compiler).  9 instructions per butterfly instead of 14.
    // The EXACT variant is the canonical gl_add / gl_sub pair.
    __device__ __forceinline__ void bfly2(u64 &u1, u64 &v1, u64 &u2, u64 &v2) {
        if (EXACT) {
            u64 a1 = gl_add(u1, v1), d1 = gl_sub(u1, v1), a2 = gl_add(u2, v2), d2 = gl_sub(u2, v2);
            u1 = a1; v1 = d1; u2 = a2; v2 = d2;
            return;
        }
#if defined(__HIP_DEVICE_COMPILE__)
        u32 u1l = (u32)u1, u1h = (u32)(u1 >> 32), v1l = (u32)v1, v1h = (u32)(v1 >> 32);
        u32 u2l = (u32)u2, u2h = (u32)(u2 >> 32), v2l = (u32)v2, v2h = (u32)(v2 >> 32);
        u32 a1l, a1h, d1l, d1h, a2l, a2h, d2l, d2h, m0, m1, m2, m3;
        u64 k0, k1, k2, k3;
        asm("v_add_co_u32 %[a1l], %[k0], %[u1l], %[v1l]\n\t"
            "v_sub_co_u32 %[d1l], %[k1], %[u1l], %[v1l]\n\t"
            "v_add_co_u32 %[a2l], %[k2], %[u2l], %[v2l]\n\t"
            "v_sub_co_u32 %[d2l], %[k3], %[u2l], %[v2l]\n\t"
            "v_addc_co_u32 %[a1h], %[k0], %[u1h], %[v1h], %[k0]\n\t"
            "v_subb_co_u32 %[d1h], %[k1], %[u1h], %[v1h], %[k1]\n\t"
            "v_addc_co_u32 %[a2h], %[k2], %[u2h], %[v2h], %[k2]\n\t"
            "v_subb_co_u32 %[d2h], %[k3], %[u2h], %[v2h], %[k3]\n\t"
            "v_cndmask_b32_e64 %[m0], 0, -1, %[k0]\n\t"
            "v_cndmask_b32_e64 %[m1], 0, -1, %[k1]\n\t"
            "v_cndmask_b32_e64 %[m2], 0, -1, %[k2]\n\t"
            "v_cndmask_b32_e64 %[m3], 0, -1, %[k3]\n\t"
            : [a1l] "=&v"(a1l), [a1h] "=&v"(a1h), [d1l] "=&v"(d1l), [d1h] "=&v"(d1h), [a2l] "=&v"(a2l), [a2h] "=&v"(a2h),
              [d2l] "=&v"(d2l), [d2h] "=&v"(d2h), [m0] "=&v"(m0), [m1] "=&v"(m1), [m2] "=&v"(m2), [m3] "=&v"(m3),
              [k0] "=&s"(k0), [k1] "=&s"(k1), [k2] "=&s"(k2), [k3] "=&s"(k3)
            : [u1l] "v"(u1l), [u1h] "v"(u1h), [v1l] "v"(v1l), [v1h] "v"(v1h), [u2l] "v"(u2l), [u2h] "v"(u2h), [v2l] "v"(v2l), [v2h] "v"(v2h));
        u64 a1 = ((u64)a1h << 32) | a1l, a2 = ((u64)a2h << 32) | a2l;
        asm("v_mad_u64_u32 %[a1], %[k0], %[m0], 1, %[a1]\n\t"
            "v_sub_co_u32 %[d1l], %[k1], %[d1l], %[m1]\n\t"
            "v_mad_u64_u32 %[a2], %[k2], %[m2], 1, %[a2]\n\t"
            "v_sub_co_u32 %[d2l], %[k3], %[d2l], %[m3]\n\t"
            "s_or_b64 %[fl], %[fl], %[k0]\n\t"
            "v_subbrev_co_u32 %[d1h], %[k1], 0, %[d1h], %[k1]\n\t"
            "s_or_b64 %[fl], %[fl], %[k2]\n\t"
            "v_subbrev_co_u32 %[d2h], %[k3], 0, %[d2h], %[k3]\n\t"
            "s_or_b64 %[fl], %[fl], %[k1]\n\t"
            "s_or_b64 %[fl], %[fl], %[k3]\n\t"
            : [a1] "+v"(a1), [a2] "+v"(a2), [d1l] "+v"(d1l), [d1h] "+v"(d1h), [d2l] "+v"(d2l), [d2h] "+v"(d2h), [fl] "+s"(fl),
              [k0] "=&s"(k0), [k1] "=&s"(k1), [k2] "=&s"(k2), [k3] "=&s"(k3)
            : [m0] "v"(m0), [m1] "v"(m1), [m2] "v"(m2), [m3] "v"(m3) : "scc");
        u1 = a1; v1 = ((u64)d1h << 32) | d1l; u2 = a2; v2 = ((u64)d2h << 32) | d2l;
#endif
    }
    // one butterfly (radix-2 tails): the same arithmetic with explicit pads where no other chain fills the wait states
    __device__ __forceinline__ void bfly1(u64 &u, u64 &v) {
        if (EXACT) { u64 a = gl_add(u, v), d = gl_sub(u, v); u = a; v = d; return; }
#if defined(__HIP_DEVICE_COMPILE__)
        u32 ul = (u32)u, uh = (u32)(u >> 32), vl = (u32)v, vh = (u32)(v >> 32), al, ah, dl, dh, m0, m1;
        u64 k0, k1;
        asm("v_add_co_u32 %[al], %[k0], %[ul], %[vl]\n\t"
            "v_sub_co_u32 %[dl], %[k1], %[ul], %[vl]\n\t"
            "s_nop 0\n\t"
            "v_addc_co_u32 %[ah], %[k0], %[uh], %[vh], %[k0]\n\t"
            "v_subb_co_u32 %[dh], %[k1], %[uh], %[vh], %[k1]\n\t"
            "s_nop 0\n\t"
            "v_cndmask_b32_e64 %[m0], 0, -1, %[k0]\n\t"
            "v_cndmask_b32_e64 %[m1], 0, -1, %[k1]\n\t"
            : [al] "=&v"(al), [ah] "=&v"(ah), [dl] "=&v"(dl), [dh] "=&v"(dh), [m0] "=&v"(m0), [m1] "=&v"(m1), [k0] "=&s"(k0), [k1] "=&s"(k1)
            : [ul] "v"(ul), [uh] "v"(uh), [vl] "v"(vl), [vh] "v"(vh));
        u64 a = ((u64)ah << 32) | al;
        asm("v_sub_co_u32 %[dl], %[k1], %[dl], %[m1]\n\t"
            "v_mad_u64_u32 %[a], %[k0], %[m0], 1, %[a]\n\t"
            "s_or_b64 %[fl], %[fl], %[k0]\n\t"
            "v_subbrev_co_u32 %[dh], %[k1], 0, %[dh], %[k1]\n\t"
            "s_or_b64 %[fl], %[fl], %[k1]\n\t"
            : [a] "+v"(a), [dl] "+v"(dl), [dh] "+v"(dh), [fl] "+s"(fl), [k0] "=&s"(k0), [k1] "=&s"(k1)
            : [m0] "v"(m0), [m1] "v"(m1) : "scc");
        u = a; v = ((u64)dh << 32) | dl;
#endif
    }
    // x * 2^S as a weak residue, S a compile-time constant with 0 < S < 96 and S % 32 != 0 (the in-network twiddles of the radix-16
    // NTT networks: w_16 = 2^12).  The shifted value occupies three 32-bit words at word offset q = S / 32:
    //   q = 0:  lo64 + c 2^64                 = lo64 + c eps                                   5 instructions
    //   q = 1:  a 2^32 + b 2^64 + c 2^96      = (a:0) + b eps - c                              8
    //   q = 2:  a 2^64 + (c:b) 2^96           = a eps - (c:b) = (a:0) - ((c:b) + a)            8
    // against 12 for a general product.  Rare wraps (a result below 2^32 before the last subtraction) go to the sticky word.
    template <int S> __device__ __forceinline__ u64 mul_pow2(u64 x) {
        static_assert(S > 0 && S < 96 && (S % 32) != 0, "shift twiddle out of range");
        constexpr int q = S / 32, r = S % 32;
#if defined(__HIP_DEVICE_COMPILE__)
        const u32 x0 = (u32)x, x1 = (u32)(x >> 32);
        if (q == 0) return gl_mad_eps(x1 >> (32 - r), x << r);
        if (q == 1) {
            const u32 wa = x0 << r, wb = __builtin_amdgcn_alignbit(x1, x0, 32 - r), wc = x1 >> (32 - r);
            const u64 t = gl_mad_eps(wb, (u64)wa << 32);
            u32 tl = (u32)t, th = (u32)(t >> 32), dl, dh;
            asm("v_sub_co_u32 %[dl], vcc, %[tl], %[h]\n\t"
                "s_nop 1\n\t"
                "v_subbrev_co_u32 %[dh], vcc, 0, %[th], vcc\n\t"
                "s_or_b64 %[fl], %[fl], vcc\n\t"
                : [dl] "=&v"(dl), [dh] "=&v"(dh), [fl] "+s"(fl) : [tl] "v"(tl), [th] "v"(th), [h] "v"(wc) : "vcc", "scc");
            return ((u64)dh << 32) | dl;
        }
        {
            const u32 wa = x0 << r;
            const u64 T2 = gl_madd1(wa, x >> (32 - r));          // (c:b) + a < 2^(32 + r) + 2^32: no overflow
            u32 tl = (u32)T2, th = (u32)(T2 >> 32), dl, dh, m;
            // (a:0) - T2; a borrow means the true value is negative: + p = - eps on the wrapped difference, which is then at
            // least 2^64 - 2^53, so the second subtraction cannot borrow (r <= 20 for every S used)
            asm("v_sub_co_u32 %[dl], vcc, 0, %[tl]\n\t"
                "s_nop 1\n\t"
                "v_subb_co_u32 %[dh], vcc, %[a], %[th], vcc\n\t"
                "s_nop 1\n\t"
                "v_cndmask_b32_e64 %[m], 0, -1, vcc\n\t"
                "v_sub_co_u32 %[dl], vcc, %[dl], %[m]\n\t"
                "s_nop 1\n\t"
                "v_subbrev_co_u32 %[dh], vcc, 0, %[dh], vcc\n\t"
                : [dl] "=&v"(dl), [dh] "=&v"(dh), [m] "=&v"(m) : [tl] "v"(tl), [th] "v"(th), [a] "v"(wa) : "vcc");
            return ((u64)dh << 32) | dl;
        }
#else
        return 0;
#endif
    }
    // flags a weak value that is not canonical (about one in 2^32): results headed for memory must be canonical
    __device__ __forceinline__ void check_canon(u64 x) {
        if (EXACT) return;
#if defined(__HIP_DEVICE_COMPILE__)
        asm("v_cmp_ge_u64 vcc, %[x], %[p]\n\t"
            "s_or_b64 %[fl], %[fl], vcc\n\t"
            : [fl] "+s"(fl) : [x] "v"(x), [p] "s"((u64)GL_P) : "vcc", "scc");
#endif
    }
    // sum_i x_i * c_i with small constants c_i (their sum below 2^31; x_i any 64-bit representatives): the shape of an MDS row.  Two 64-bit
    // column sums -- low words times c, high words times c, one v_mad_u64_u32 each per term, no carries -- then one fold
    // al + ah 2^32 = low64 + high 2^64 -> low64 + high eps.  34 instructions for a 12-term row against 12 products + 11 additions (199).
    // Exact either way; the fast arithmetic only skips the final canonicalisation (flagged when it would have fired).
    static __device__ __forceinline__ void dot_acc(u64 &al, u64 &ah, u64 x, u32 c) { al += (u64)(u32)x * c; ah += (x >> 32) * c; }
    __device__ __forceinline__ u64 dot_fold(u64 al, u64 ah) {
#if defined(__HIP_DEVICE_COMPILE__)
        u32 th, cy;
        th = __builtin_addc((u32)(al >> 32), (u32)ah, 0u, &cy);
        const u32 high = (u32)(ah >> 32) + cy;
        const u64 r = gl_mad_eps(high, ((u64)th << 32) | (u32)al);
        if (EXACT) return gl_canon(r);
        check_canon(r);
        return r;
#else
        return 0;
#endif
    }
    // a*b + c (c canonical): c's words join the first two columns ((2^32-1)^2 + 2(2^32-1) = 2^64-1: no overflow)
    __device__ __forceinline__ u64 mul_add(u64 a, u64 b, u64 c) {
        if (EXACT) return gl_mul_add(a, b, c);
        u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
        u64 P0 = gl_madd1((u32)c, (u64)a0 * b0);
        u64 A2 = gl_madd1((u32)(c >> 32), gl_madd1((u32)(P0 >> 32), (u64)a0 * b1));
        u64 X, cy;
        asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(X), "=s"(cy) : "v"(a1), "v"(b0), "v"(A2));
        u64 A4 = gl_madd1((u32)(X >> 32), (u64)a1 * b1);
        return tail((X << 32) | (u32)P0, A4, cy);
    }
};

struct gl2 { u64 a, b; };
GLD gl2 gl2_make(u64 a, u64 b) { gl2 r; r.a = a; r.b = b; return r; }
GLD gl2 gl2_add(gl2 x, gl2 y) { return gl2_make(gl_add(x.a, y.a), gl_add(x.b, y.b)); }
GLD gl2 gl2_sub(gl2 x, gl2 y) { return gl2_make(gl_sub(x.a, y.a), gl_sub(x.b, y.b)); }
GLD gl2 gl2_mul(gl2 x, gl2 y) {
    u64 bb = gl_mul(x.b, y.b);
    u64 w = gl_add(gl_add(gl_dbl(gl_dbl(bb)), gl_dbl(bb)), bb);   // 7*bb
    return gl2_make(gl_add(gl_mul(x.a, y.a), w), gl_add(gl_mul(x.a, y.b), gl_mul(x.b, y.a)));
}
GLD gl2 gl2_scale(gl2 x, u64 s) { return gl2_make(gl_mul(x.a, s), gl_mul(x.b, s)); }
GLD gl2 gl2_inv(gl2 x) {
    u64 bb = gl_mul(x.b, x.b);
    u64 w = gl_add(gl_add(gl_dbl(gl_dbl(bb)), gl_dbl(bb)), bb);
    u64 d = gl_inv(gl_sub(gl_mul(x.a, x.a), w));
    return gl2_make(gl_mul(x.a, d), gl_mul(gl_neg(x.b), d));
}
GLD gl2 gl2_pow(gl2 b, u64 e) {
    gl2 r = gl2_make(1, 0);
    while (e) { if (e & 1) r = gl2_mul(r, b); b = gl2_mul(b, b); e >>= 1; }
    return r;
}

GLD u32 bitrev_u32(u32 x, u32 bits) {
#if defined(__HIP_DEVICE_COMPILE__)
    return bits ? (__brev(x) >> (32 - bits)) : 0;
#else
    u32 r = 0;
    for (u32 i = 0; i < bits; i++) { r = (r << 1) | (x & 1); x >>= 1; }
    return r;
#endif
}

// Launch index -> 256-point block of a compiled constraint kernel (prover_quotient_jit.hip).  In leaf (bit-reversed) order the `next` row
// of every point of block b lies in exactly one other block, the one whose bit-reversed block number is 2^rate_bits higher: blocks form a
// cycle, and in the plain order b = blockIdx a block's `next` rows are a quarter of the array away -- a second full read of the LDE
// (FETCH 1.7 x the local rows).  Here the launch walks the cycle: workgroups go to the eight XCDs round robin, so
// XCD x (launch indices x, x + 8, ..) gets a contiguous arc of the cycle, and what block k reads as `next` is what its neighbour on the
// same XCD (its L2) reads as `local` at about the same time.  A bijection on [0, 2^(lgL - 8)); needs lgL - 8 >= 3 + rate_bits.
GLD u32 zk_qblock(u32 j, u32 lgL, u32 rate_bits) {
    const u32 nbits = lgL - 8, nb = 1u << nbits;
    const u32 pos = (j & 7u) * (nb >> 3) + (j >> 3);
    const u32 c = ((pos << rate_bits) | (pos >> (nbits - rate_bits))) & (nb - 1);
    return bitrev_u32(c, nbits);
}
