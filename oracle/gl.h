/* ORACLE -- test infrastructure, not product code.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may link or call this.
 *
 * Goldilocks field F_p, p = 2^64 - 2^32 + 1, and its quadratic extension F_p[x]/(x^2 - 7).
 * Restates plonky2_field 0.2.2 (goldilocks_field.rs, extension/quadratic.rs), pinned at
 * /root/reference/Cargo.lock:3466-3468 and reached from /root/reference/ops/src/lib.rs:52.
 * The crate source is absent from the reference tree; spec = SURVEY.md Appendix A1.
 * gl_mul uses reduce128 as restated in A1; gl_mul_ref is the plain 128-bit '%' form and
 * orc_selftest() (oracle.c) checks the two against each other on edge and random values.
 */
#ifndef ORACLE_GL_H
#define ORACLE_GL_H
#include <stdint.h>
#include <stddef.h>

#define GL_P 0xFFFFFFFF00000001ULL
#define GL_GEN 14293326489335486720ULL            /* MULTIPLICATIVE_GROUP_GENERATOR = coset shift */
#define GL_POW2_GEN 7277203076849721926ULL        /* POWER_OF_TWO_GENERATOR, order 2^32 */
#define GL_W 7ULL                                 /* ext2: x^2 = 7 */

typedef unsigned __int128 u128;
typedef struct { uint64_t a, b; } gl2; /* a + b x */

static inline uint64_t gl_canon(uint64_t x) { return x >= GL_P ? x - GL_P : x; }
static inline uint64_t gl_add(uint64_t x, uint64_t y) {
    x = gl_canon(x); y = gl_canon(y);
    uint64_t s = x + y;
    return (s < x || s >= GL_P) ? s - GL_P : s;
}
static inline uint64_t gl_sub(uint64_t x, uint64_t y) {
    x = gl_canon(x); y = gl_canon(y);
    return x >= y ? x - y : x + (GL_P - y);
}
static inline uint64_t gl_neg(uint64_t x) { return gl_sub(0, x); }
static inline uint64_t gl_mul_ref(uint64_t x, uint64_t y) { return (uint64_t)(((u128)x * y) % GL_P); }
/* reduce128 (A1): 2^64 = 2^32 - 1, 2^96 = -1 (mod p).  Result canonical. */
static inline uint64_t gl_reduce128(u128 v) {
    uint64_t lo = (uint64_t)v, hi = (uint64_t)(v >> 64);
    uint64_t hi_hi = hi >> 32, hi_lo = hi & 0xFFFFFFFFULL;
    uint64_t t0 = lo - hi_hi;
    if (lo < hi_hi) t0 -= 0xFFFFFFFFULL;           /* borrow: subtract epsilon */
    uint64_t t1 = hi_lo * 0xFFFFFFFFULL;
    uint64_t r = t0 + t1;
    if (r < t1) r += 0xFFFFFFFFULL;                 /* carry: add epsilon */
    return gl_canon(r);
}
static inline uint64_t gl_mul(uint64_t x, uint64_t y) { return gl_reduce128((u128)x * y); }
static inline uint64_t gl_pow(uint64_t b, uint64_t e) {
    uint64_t r = 1; b = gl_canon(b);
    while (e) { if (e & 1) r = gl_mul(r, b); b = gl_mul(b, b); e >>= 1; }
    return r;
}
static inline uint64_t gl_inv(uint64_t x) { return gl_pow(x, GL_P - 2); }
/* primitive_root_of_unity(k) = POWER_OF_TWO_GENERATOR^(2^(32-k)) */
static inline uint64_t gl_root(unsigned k) { return gl_pow(GL_POW2_GEN, 1ULL << (32 - k)); }

static inline gl2 gl2_from(uint64_t a) { gl2 r = {gl_canon(a), 0}; return r; }
static inline gl2 gl2_add(gl2 x, gl2 y) { gl2 r = {gl_add(x.a, y.a), gl_add(x.b, y.b)}; return r; }
static inline gl2 gl2_sub(gl2 x, gl2 y) { gl2 r = {gl_sub(x.a, y.a), gl_sub(x.b, y.b)}; return r; }
static inline gl2 gl2_mul(gl2 x, gl2 y) {
    gl2 r = {gl_add(gl_mul(x.a, y.a), gl_mul(GL_W, gl_mul(x.b, y.b))), gl_add(gl_mul(x.a, y.b), gl_mul(x.b, y.a))};
    return r;
}
static inline gl2 gl2_scale(gl2 x, uint64_t s) { gl2 r = {gl_mul(x.a, s), gl_mul(x.b, s)}; return r; }
static inline gl2 gl2_inv(gl2 x) {
    uint64_t d = gl_inv(gl_sub(gl_mul(x.a, x.a), gl_mul(GL_W, gl_mul(x.b, x.b))));
    gl2 r = {gl_mul(x.a, d), gl_mul(gl_neg(x.b), d)};
    return r;
}
static inline gl2 gl2_pow(gl2 b, uint64_t e) {
    gl2 r = {1, 0};
    while (e) { if (e & 1) r = gl2_mul(r, b); b = gl2_mul(b, b); e >>= 1; }
    return r;
}
static inline int gl2_eq(gl2 x, gl2 y) { return gl_canon(x.a) == gl_canon(y.a) && gl_canon(x.b) == gl_canon(y.b); }

static inline uint32_t bitrev32(uint32_t i, unsigned bits) {
    uint32_t r = 0;
    for (unsigned t = 0; t < bits; t++) { r = (r << 1) | (i & 1); i >>= 1; }
    return r;
}
#endif
