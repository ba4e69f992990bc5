#!/usr/bin/env python3
"""Writes eth-tx-proof_amd/csrc/poseidon_wave_asm.inc: the two round bodies of the wave-wide Poseidon permutation (poseidon_wave.cuh)
as hand-scheduled gfx950 instruction sequences.

Why: the Fiat-Shamir transcript and the small Merkle levels run ONE permutation per wave, and a wave alone on a SIMD pays 5.5-9.5
cycles per instruction of ANY kind, s_nop included (tools/ubench_lat.hip).  Compiled from the shared inline-asm primitives the
round bodies carried 26-29 pad instructions and a dozen register moves per round; here every hazard gap that can be filled is filled
by the neighbouring independent chain (x^3 and x^4 of an s-box, the two column sums of the MDS row, v_readlane two terms ahead) and
only six pads per round remain.  The sequences use fixed scratch registers (v96..v127, s40..s51: clobbers of the asm statement).

Checked HERE, before anything is compiled: (1) a lane-accurate simulation of every emitted instruction against the field arithmetic
(full round = MDS(sbox(s)) + next constants, partial round = acc + c * x0^7), (2) the hazard rule the library's kernels obey on gfx950
(a vector-written SGPR / VCC is read by a vector instruction no sooner than two issue slots later).
usage: python tools/gen_wave_asm.py [--check]   (writes the .inc, or with --check compares it; exits non-zero if a check fails)"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = 0xFFFFFFFF00000001
M32 = 0xFFFFFFFF
M64 = (1 << 64) - 1
EPS = 0xFFFFFFFF
CIRC = [17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20]
DIAG0 = 8


# ------------------------------------------------------------------------------------------------ instruction builders
class Prog:
    def __init__(self):
        self.ins = []          # (op, dict)

    def add(self, op, **kw):
        self.ins.append((op, kw))


def pair(n):
    return "v[%d:%d]" % (n, n + 1)


def spair(n):
    return "s[%d:%d]" % (n, n + 1)


# fixed scratch registers
K, CY, K2, CY2, BA, BB = 40, 42, 44, 46, 52, 54       # SGPR pairs: junk carry, mid-column carry (two product slots), borrow masks
RL = [48, 49, 50, 51]                                 # v_readlane targets (two alternating pairs)


class Slot:
    """scratch registers of one product: P0, T1, T3, Xm pairs + mask temp"""
    def __init__(self, base, k, cy, b):
        self.P0, self.T1, self.T3, self.Xm, self.M = base, base + 2, base + 4, base + 6, base + 8
        self.K, self.CY, self.B = k, cy, b


SA = Slot(100, K, CY, BA)
SB = Slot(110, K2, CY2, BB)


def v(n):
    return "v%d" % n


def prod_steps(s, a0, a1, b0, b1, d0, d1, add_lo=None, add_hi=None):
    """the instruction list of one product (a1:a0) * (b1:b0) [+ (add_hi:add_lo)] -> weak residue in (d1:d0); names are register strings.
    Returned as a list of (op, kw, tag) where tag marks the two places that need two wait states BEFORE the instruction."""
    st = []
    st.append(("mad", dict(d=s.P0, co=s.K, a=a0, b=b0, c=None), None))
    if add_lo is not None:
        st.append(("mad", dict(d=s.P0, co=s.K, a=add_lo, b=1, c=s.P0), None))
    st.append(("mad", dict(d=s.T1, co=s.K, a=a0, b=b1, c=None), None))
    st.append(("mad", dict(d=s.T3, co=s.K, a=a1, b=b1, c=None), None))
    st.append(("mad", dict(d=s.T1, co=s.K, a=v(s.P0 + 1), b=1, c=s.T1), None))
    if add_hi is not None:
        st.append(("mad", dict(d=s.T1, co=s.K, a=add_hi, b=1, c=s.T1), None))
    st.append(("mad", dict(d=s.Xm, co=s.CY, a=a1, b=b0, c=s.T1), None))
    st.append(("mad", dict(d=s.T3, co=s.K, a=v(s.Xm + 1), b=1, c=s.T3), None))
    st.append(("mov", dict(d=v(s.P0 + 1), s=v(s.Xm)), None))
    st.append(("mad", dict(d=s.P0, co=s.K, a=v(s.T3), b=-1, c=s.P0), None))
    st.append(("cnd", dict(d=v(s.M), m=s.K), "gap"))
    st.append(("mad", dict(d=s.P0, co=s.K, a=v(s.M), b=1, c=s.P0), None))
    st.append(("subb", dict(d=d0, bo=s.B, a=v(s.P0), b=v(s.T3 + 1), bi=s.CY), None))
    st.append(("subbrev", dict(d=d1, bo=s.B, a=0, b=v(s.P0 + 1), bi=s.B), "gap"))
    st.append(("sor", dict(m=s.B), None))
    return st


def emit_single(p, steps):
    for op, kw, tag in steps:
        if tag == "gap":
            p.add("nop", n=1)
        p.add(op, **kw)


def emit_pair(p, sa, sb):
    """two independent products, instruction by instruction; a gap of one chain is the other chain's instruction + at most one pad"""
    assert len(sa) == len(sb)
    i = 0
    n = len(sa)
    out = []
    for k in range(n):
        out.append(("A", sa[k]))
        out.append(("B", sb[k]))
    # a "gap" instruction needs two slots after its chain's previous instruction: in the A B A B order there is exactly one (the other
    # chain's instruction) -> one pad before the A instruction of a gap pair serves both chains
    for who, (op, kw, tag) in out:
        if tag == "gap" and who == "A":
            p.add("nop", n=0)
        p.add(op, **kw)


def gen_full4(rows):
    """four full rounds in one block: s <- MDS(sbox(s)) + constants of the NEXT layer, four times.  rows[r] = row of PSD_WV_RC (in the LDS
    image at `addr` + 512 * row, `addr` = this lane's cell of row 0) that round r's MDS adds, or None for zeros; m0..m11 = this lane's
    MDS row.  The state enters and leaves in (sl, sh); between the rounds it stays in the MDS accumulator pair."""
    p = Prog()
    AL, AH = 124, 126
    RC = [92, 94]                                         # constant pairs, double-buffered
    X2 = (v(98), v(99)); X3 = (v(120), v(121)); X4 = (v(122), v(123)); X7 = (v(96), v(97))
    p.add("wait", n=0)                                    # nothing of the caller's outstanding: lgkmcnt below counts this block's reads only
    if rows[0] is not None:
        p.add("dsread", d=RC[0], addr="%[addr]", off=512 * rows[0])
    for r in range(4):
        sl, sh = ("%[sl]", "%[sh]") if r == 0 else (v(AL), v(AL + 1))
        if r + 1 < 4 and rows[r + 1] is not None:
            p.add("dsread", d=RC[(r + 1) % 2], addr="%[addr]", off=512 * rows[r + 1])
        emit_single(p, prod_steps(SA, sl, sh, sl, sh, X2[0], X2[1]))
        emit_pair(p, prod_steps(SA, X2[0], X2[1], sl, sh, X3[0], X3[1]), prod_steps(SB, X2[0], X2[1], X2[0], X2[1], X4[0], X4[1]))
        emit_single(p, prod_steps(SA, X3[0], X3[1], X4[0], X4[1], X7[0], X7[1]))
        # column sums start with the next layer's constants; v_readlane runs two terms ahead of the multiply-adds
        p.add("rl", d=RL[0], s=X7[0], lane=0); p.add("rl", d=RL[1], s=X7[1], lane=0)
        if rows[r] is not None:
            later = 1 if (r + 1 < 4 and rows[r + 1] is not None) else 0
            p.add("wait", n=later)
            p.add("mad", d=AL, co=K, a=v(RC[r % 2]), b=1, c=None)
            p.add("mad", d=AH, co=K, a=v(RC[r % 2] + 1), b=1, c=None)
        for i in range(12):
            if i + 1 < 12:
                nx = RL[2:4] if i % 2 == 0 else RL[0:2]
                p.add("rl", d=nx[0], s=X7[0], lane=i + 1); p.add("rl", d=nx[1], s=X7[1], lane=i + 1)
            cur = RL[0:2] if i % 2 == 0 else RL[2:4]
            first = i == 0 and rows[r] is None
            p.add("mad", d=AL, co=K, a="%%[m%d]" % i, b="s%d" % cur[0], c=None if first else AL)
            p.add("mad", d=AH, co=K, a="%%[m%d]" % i, b="s%d" % cur[1], c=None if first else AH)
        # al + ah 2^32 = low64 + high 2^64 -> low64 + high eps (high < 2^12), one wrap repaired
        p.add("addco", d=v(AL + 1), co="vcc", a=v(AL + 1), b=v(AH))
        p.add("nop", n=1)
        p.add("addc", d=v(108), co="vcc", a=0, b=v(AH + 1), ci="vcc")
        p.add("mad", d=AL, co=K, a=v(108), b=-1, c=AL)
        p.add("nop", n=1)
        p.add("cnd", d=v(109), m=K)
        p.add("mad", d=AL, co=K, a=v(109), b=1, c=AL)
    p.add("mov", d="%[sl]", s=v(AL)); p.add("mov", d="%[sh]", s=v(AL + 1))
    return p


NPART = 22


def gen_part_all():
    """the 22 partial rounds in one block: round q reads x0 = acc[lane q], then acc <- acc + c_q * x0^7 in every lane (the rounds' constants
    K[q] are folded into the starting values).  c_q = row q of PSD_WV_CHAIN, read from the LDS image at `addr` + 512 q one round ahead."""
    p = Prog()
    al, ah = "%[al]", "%[ah]"
    CB = [92, 94]
    X = (v(96), v(97)); X2 = (v(98), v(99)); X3 = (v(120), v(121)); X4 = (v(122), v(123)); U = (v(124), v(125))
    p.add("wait", n=0)
    p.add("dsread", d=CB[0], addr="%[addr]", off=0)
    for q in range(NPART):
        if q + 1 < NPART:
            p.add("dsread", d=CB[(q + 1) % 2], addr="%[addr]", off=512 * (q + 1))
        p.add("rl", d=RL[0], s=al, lane=q); p.add("rl", d=RL[1], s=ah, lane=q)
        p.add("nop", n=0)
        p.add("mov", d=X[0], s="s%d" % RL[0]); p.add("mov", d=X[1], s="s%d" % RL[1])
        emit_single(p, prod_steps(SA, X[0], X[1], X[0], X[1], X2[0], X2[1]))
        emit_pair(p, prod_steps(SA, X2[0], X2[1], X[0], X[1], X3[0], X3[1]), prod_steps(SB, X2[0], X2[1], X2[0], X2[1], X4[0], X4[1]))
        emit_single(p, prod_steps(SA, X3[0], X3[1], X4[0], X4[1], U[0], U[1]))
        p.add("wait", n=1 if q + 1 < NPART else 0)
        c = CB[q % 2]
        emit_single(p, prod_steps(SA, v(c), v(c + 1), U[0], U[1], al, ah, add_lo=al, add_hi=ah))
    return p


# ------------------------------------------------------------------------------------------------ text
def opnd(x):
    if isinstance(x, int):
        return str(x)
    return x


def text(p):
    out = []
    for op, k in p.ins:
        if op == "mad":
            c = "0" if k["c"] is None else pair(k["c"])
            out.append("v_mad_u64_u32 %s, %s, %s, %s, %s" % (pair(k["d"]), spair(k["co"]), opnd(k["a"]), opnd(k["b"]), c))
        elif op == "mov":
            out.append("v_mov_b32 %s, %s" % (k["d"], k["s"]))
        elif op == "cnd":
            out.append("v_cndmask_b32_e64 %s, 0, -1, %s" % (k["d"], spair(k["m"])))
        elif op == "subb":
            out.append("v_subb_co_u32 %s, %s, %s, %s, %s" % (k["d"], spair(k["bo"]), k["a"], k["b"], spair(k["bi"])))
        elif op == "subbrev":
            out.append("v_subbrev_co_u32 %s, %s, %s, %s, %s" % (k["d"], spair(k["bo"]), opnd(k["a"]), k["b"], spair(k["bi"])))
        elif op == "addco":
            out.append("v_add_co_u32 %s, vcc, %s, %s" % (k["d"], k["a"], k["b"]))
        elif op == "addc":
            out.append("v_addc_co_u32 %s, vcc, %s, %s, vcc" % (k["d"], opnd(k["a"]), k["b"]))
        elif op == "rl":
            out.append("v_readlane_b32 s%d, %s, %s" % (k["d"], k["s"], opnd(k["lane"])))
        elif op == "sor":
            out.append("s_or_b64 %%[fl], %%[fl], %s" % spair(k["m"]))
        elif op == "nop":
            out.append("s_nop %d" % k["n"])
        elif op == "dsread":
            out.append("ds_read_b64 %s, %s offset:%d" % (pair(k["d"]), k["addr"], k["off"]))
        elif op == "wait":
            out.append("s_waitcnt lgkmcnt(%d)" % k["n"])
        else:
            raise ValueError(op)
    return out


# ------------------------------------------------------------------------------------------------ hazard check
def sregs_of_pair(n):
    return ["s%d" % n, "s%d" % (n + 1)]


def check_hazards(p, name):
    """a vector instruction may read an SGPR / VCC that a vector instruction (v_readlane included) wrote only if at least two issue
    slots lie between them (s_nop n counts n + 1)"""
    last = {}          # sreg -> slot of the last VECTOR write
    slot = 0
    for op, k in p.ins:
        if op == "nop":
            slot += k["n"] + 1
            continue
        reads, writes = [], []
        if op == "mad":
            writes += sregs_of_pair(k["co"])
            for x in (k["a"], k["b"]):
                if isinstance(x, str) and x.startswith("s") and x[1:].isdigit():
                    reads.append(x)
        elif op == "mov":
            if k["s"].startswith("s") and k["s"][1:].isdigit():
                reads.append(k["s"])
        elif op == "cnd":
            reads += sregs_of_pair(k["m"])
        elif op in ("subb", "subbrev"):
            reads += sregs_of_pair(k["bi"]); writes += sregs_of_pair(k["bo"])
        elif op == "addco":
            writes += ["vcc"]
        elif op == "addc":
            reads += ["vcc"]; writes += ["vcc"]
        elif op == "rl":
            writes.append("s%d" % k["d"])
        elif op in ("dsread", "wait"):
            slot += 1
            continue
        elif op == "sor":
            slot += 1          # scalar instruction: reads a vector-written mask at once (as the library's kernels do), counts as a slot
            continue
        for r in reads:
            if r in last and slot - last[r] < 3:
                raise SystemExit("%s: hazard: %s read %d slot(s) after its vector write (%s %s)" % (name, r, slot - last[r], op, k))
        for w in writes:
            last[w] = slot
        slot += 1
    return slot


# ------------------------------------------------------------------------------------------------ simulation (64 lanes)
class Sim:
    def __init__(self, inputs):
        self.v = {}            # name -> [64 x u32]
        self.s = {}            # name -> u32
        self.lds = {}          # byte offset from a lane's address -> [64 x u64]
        for k, val in inputs.items():
            if isinstance(val, list):
                self.v[k] = list(val)
            else:
                self.s[k] = val

    def rd(self, x, lane):
        if isinstance(x, int):
            return x & M32
        if x in self.v:
            return self.v[x][lane]
        if x in self.s:
            return self.s[x]
        raise KeyError(x)

    def mask(self, n):
        if n == "vcc":
            return self.s.get("vcc_lo", 0) | (self.s.get("vcc_hi", 0) << 32)
        return self.s.get("s%d" % n, 0) | (self.s.get("s%d" % (n + 1), 0) << 32)

    def setmask(self, n, m):
        if n == "vcc":
            self.s["vcc_lo"], self.s["vcc_hi"] = m & M32, m >> 32
        else:
            self.s["s%d" % n], self.s["s%d" % (n + 1)] = m & M32, m >> 32

    def setv(self, name, lane, val):
        self.v.setdefault(name, [0] * 64)[lane] = val & M32

    def run(self, p):
        for op, k in p.ins:
            if op == "nop":
                continue
            if op == "mad":
                m = 0
                res = []
                for l in range(64):
                    c = 0 if k["c"] is None else (self.v[v(k["c"])][l] | (self.v[v(k["c"] + 1)][l] << 32))
                    t = self.rd(k["a"], l) * self.rd(k["b"], l) + c
                    if t >> 64:
                        m |= 1 << l
                    res.append(t & M64)
                for l in range(64):
                    self.setv(v(k["d"]), l, res[l]); self.setv(v(k["d"] + 1), l, res[l] >> 32)
                self.setmask(k["co"], m)
            elif op == "mov":
                vals = [self.rd(k["s"], l) for l in range(64)]
                for l in range(64):
                    self.setv(k["d"], l, vals[l])
            elif op == "cnd":
                m = self.mask(k["m"])
                for l in range(64):
                    self.setv(k["d"], l, M32 if (m >> l) & 1 else 0)
            elif op in ("subb", "subbrev"):
                bi = self.mask(k["bi"]); bo = 0
                res = []
                for l in range(64):
                    a, b = self.rd(k["a"], l), self.rd(k["b"], l)
                    if op == "subbrev":
                        a, b = b, a
                    t = a - b - ((bi >> l) & 1)
                    if t < 0:
                        bo |= 1 << l
                    res.append(t & M32)
                for l in range(64):
                    self.setv(k["d"], l, res[l])
                self.setmask(k["bo"], bo)
            elif op == "addco":
                co = 0; res = []
                for l in range(64):
                    t = self.rd(k["a"], l) + self.rd(k["b"], l)
                    if t >> 32:
                        co |= 1 << l
                    res.append(t & M32)
                for l in range(64):
                    self.setv(k["d"], l, res[l])
                self.setmask("vcc", co)
            elif op == "addc":
                ci = self.mask("vcc"); co = 0; res = []
                for l in range(64):
                    t = self.rd(k["a"], l) + self.rd(k["b"], l) + ((ci >> l) & 1)
                    if t >> 32:
                        co |= 1 << l
                    res.append(t & M32)
                for l in range(64):
                    self.setv(k["d"], l, res[l])
                self.setmask("vcc", co)
            elif op == "rl":
                lane = k["lane"] if isinstance(k["lane"], int) else self.s[k["lane"]]
                self.s["s%d" % k["d"]] = self.rd(k["s"], lane)
            elif op == "sor":
                self.s["fl"] = self.s.get("fl", 0) | self.mask(k["m"])
            elif op == "dsread":
                row = self.lds[k["off"]]                   # what each lane finds at its address + off
                for l in range(64):
                    self.setv(v(k["d"]), l, row[l]); self.setv(v(k["d"] + 1), l, row[l] >> 32)
            elif op == "wait":
                pass
            else:
                raise ValueError(op)


def mds_row_ref(state, lane):
    """out[lane] = sum_i CIRC[i] * s[(i + lane) % 12] + 8 s[0] for lane 0  (plonky2 poseidon.rs mds_row_shf); per-lane coefficients m[i] of s_i"""
    m = [0] * 12
    if lane < 12:
        for i in range(12):
            m[(i + lane) % 12] += CIRC[i]
        if lane == 0:
            m[0] += DIAG0
    return m


def selftest(halves, part, rnd, rounds=12):
    edge = [0, 1, P - 1, P, M64, 1 << 32, (1 << 32) - 1, 1 << 63]
    flagged = 0
    ms = [mds_row_ref(None, l) for l in range(64)]
    for it in range(rounds):
        for rows, full in halves:
            st = [rnd.choice(edge) if rnd.random() < 0.2 else rnd.getrandbits(64) for _ in range(64)]
            inp = {"%[sl]": [x & M32 for x in st], "%[sh]": [x >> 32 for x in st], "fl": 0}
            for i in range(12):
                inp["%%[m%d]" % i] = [ms[l][i] for l in range(64)]
            sim = Sim(inp)
            rcs = {}
            for row in rows:
                if row is not None:
                    rcs[row] = [rnd.randrange(P) if l < 12 else 0 for l in range(64)]
                    sim.lds[512 * row] = rcs[row]
            sim.run(full)
            want = [x % P for x in st[:12]]
            for row in rows:
                sb = [pow(x, 7, P) for x in want]
                want = [(sum(ms[l][j] * sb[j] for j in range(12)) + (rcs[row][l] if row is not None else 0)) % P for l in range(12)]
            if sim.s.get("fl", 0) & 0xFFF:
                flagged += 1
            else:
                for l in range(64):
                    got = sim.v["%[sl]"][l] | (sim.v["%[sh]"][l] << 32)
                    if got % P != (want[l] if l < 12 else 0):
                        raise SystemExit("full rounds %s: lane %d differs (iteration %d)" % (rows, l, it))
        # ---- the partial rounds
        acc = [rnd.choice(edge) if rnd.random() < 0.1 else rnd.getrandbits(64) for _ in range(64)]
        sim = Sim({"%[al]": [x & M32 for x in acc], "%[ah]": [x >> 32 for x in acc], "fl": 0})
        cs = [[rnd.randrange(P) for _ in range(64)] for _ in range(NPART)]
        for q in range(NPART):
            sim.lds[512 * q] = cs[q]
        sim.run(part)
        want = [x % P for x in acc]
        for q in range(NPART):
            u = pow(want[q], 7, P)
            want = [(want[l] + cs[q][l] * u) % P for l in range(64)]
        if sim.s.get("fl", 0):
            flagged += 1
        else:
            for l in range(64):
                got = sim.v["%[al]"][l] | (sim.v["%[ah]"][l] << 32)
                if got % P != want[l]:
                    raise SystemExit("partial rounds: lane %d differs (iteration %d)" % (l, it))
    return flagged


def main():
    # rows of PSD_WV_RC: 0..3 = rounds 0..3, 4..7 = rounds 26..29, 8 = PR_FIRST; a round's MDS adds what the NEXT layer needs
    halves = [([1, 2, 3, 8], None), ([5, 6, 7, None], None)]
    halves = [(rows, gen_full4(rows)) for rows, _ in halves]
    part = gen_part_all()
    slots = [check_hazards(f, "full rounds %s" % rows) for rows, f in halves] + [check_hazards(part, "partial rounds")]
    flagged = selftest(halves, part, random.Random(7))
    texts = [text(f) for _, f in halves] + [text(part)]
    clob_v = ", ".join('"v%d"' % i for i in range(92, 128))
    clob_s = ", ".join('"s%d"' % i for i in range(40, 56))
    lines = ["// GENERATED by tools/gen_wave_asm.py -- do not edit.  The round bodies of the wave-wide Poseidon permutation (poseidon_wave.cuh),",
             "// hand-scheduled: four full rounds per block (%d instructions, %d issue slots with pads), the 22 partial rounds in one block (%d, %d)." % (len(texts[0]), slots[0], len(texts[2]), slots[2]),
             "// Simulated lane by lane against the field arithmetic and checked against the two-wait-state rule for vector-written SGPRs by the generator.",
             "#define PSDW_ASM_CLOBBERS \"vcc\", \"scc\", \"memory\", %s, %s" % (clob_v, clob_s)]
    for name, t in zip(("PSDW_FULL_ROUNDS_FIRST_ASM", "PSDW_FULL_ROUNDS_LAST_ASM", "PSDW_PART_ROUNDS_ASM"), texts):
        lines.append("#define %s \\" % name)
        lines += ['    "%s\\n\\t" \\' % x for x in t]
        lines.append('    ""')
    path = os.path.join(ROOT, "eth-tx-proof_amd", "csrc", "poseidon_wave_asm.inc")
    if "--check" in sys.argv:        # the committed file must be what this generator emits (tests/test_abi_cpu.py)
        if open(path).read() != "\n".join(lines) + "\n":
            raise SystemExit("%s is not what tools/gen_wave_asm.py emits: run the generator" % path)
        print("ok: %s is current; simulation and hazard check passed" % path)
        return
    open(path, "w").write("\n".join(lines) + "\n")
    print("wrote %s: four full rounds %d / %d instructions (%d / %d slots), 22 partial rounds %d (%d); %d random cases raised the sticky flag"
          % (path, len(texts[0]), len(texts[1]), slots[0], slots[1], len(texts[2]), slots[2], flagged))


if __name__ == "__main__":
    main()
