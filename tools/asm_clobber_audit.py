#!/usr/bin/env python3
"""Every inline-asm statement of the device sources against the implicit registers its text touches.

Round 4's memory fault: asm blocks of gl64.cuh ended in `s_or_b64 fl, fl, vcc` -- a scalar instruction, which
writes SCC -- and declared only "vcc" as clobbered.  hipcc forms 64-bit scalar addresses as s_add_u32 / s_addc_u32 with the carry in SCC
and was free to schedule such a block between the two: the carry into the upper half of a pointer was lost whenever the lower half
wrapped, i.e. for accesses behind a 4 GiB boundary inside a block that straddles one.  This audit parses every asm statement and fails when
  * its text names vcc and "vcc" is not among its clobbers (or operands),
  * it contains a scalar ALU / compare instruction (anything s_* that writes SCC) and "scc" is not among its clobbers,
  * it writes exec or m0 without declaring it.
tests/test_abi_cpu.py runs it over csrc/ and tools/; tools/scc_pairs_in_code_objects() checks the built code objects for the same
pattern from the other side (an SCC writer between a scalar add and its add-with-carry)."""
import glob
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# scalar instructions that leave SCC alone; every other s_* mnemonic is taken to write it
NO_SCC = set("""s_nop s_waitcnt s_mov_b32 s_mov_b64 s_barrier s_sleep s_setprio s_cselect_b32 s_cselect_b64 s_mul_i32 s_mul_hi_u32 s_mul_hi_i32
s_movk_i32 s_getpc_b64 s_branch s_cbranch_scc0 s_cbranch_scc1 s_cbranch_vccz s_cbranch_vccnz s_cbranch_execz s_cbranch_execnz s_endpgm
s_bitset1_b32 s_bitset0_b32 s_sext_i32_i16 s_sext_i32_i8 s_pack_ll_b32_b16 s_cmov_b32 s_cmov_b64 s_brev_b32 s_brev_b64 s_load_dword
s_load_dwordx2 s_load_dwordx4 s_load_dwordx8 s_load_dwordx16 s_memtime s_memrealtime s_dcache_wb s_setreg_b32 s_getreg_b32 s_sethalt
s_mulk_i32 s_ff1_i32_b32 s_ff1_i32_b64 s_flbit_i32_b32 s_flbit_i32_b64 s_icache_inv s_trap s_code_end s_set_gpr_idx_on s_set_gpr_idx_off""".split())


def asm_statements(text):
    """(offset, statement text) of every asm(...) / asm volatile(...) in a C++ source, macros expanded by the caller"""
    out = []
    for m in re.finditer(r"\basm(?:\s+volatile)?\s*\(", text):
        j, depth = m.end(), 1
        while depth and j < len(text):
            c = text[j]
            if c == '"':
                j += 1
                while text[j] != '"':
                    j += 2 if text[j] == "\\" else 1
            elif c == "(":
                depth += 1
            elif c == ")":
                depth -= 1
            j += 1
        out.append((m.start(), text[m.start():j]))
    return out


def split_sections(stmt):
    """template string, outputs, inputs, clobbers of one asm statement (top-level colons)"""
    body = stmt[stmt.index("(") + 1:-1]
    parts, cur, depth, i = [], [], 0, 0
    while i < len(body):
        c = body[i]
        if c == '"':
            k = i + 1
            while body[k] != '"':
                k += 2 if body[k] == "\\" else 1
            cur.append(body[i:k + 1]); i = k + 1; continue
        if c in "([":
            depth += 1
        elif c in ")]":
            depth -= 1
        if c == ":" and depth == 0 and not (i + 1 < len(body) and body[i + 1] == ":") and not (i and body[i - 1] == ":"):
            parts.append("".join(cur)); cur = []
        else:
            cur.append(c)
        i += 1
    parts.append("".join(cur))
    parts += [""] * (4 - len(parts))
    template = "".join(re.findall(r'"((?:[^"\\]|\\.)*)"', parts[0]))
    return template, parts[1], parts[2], parts[3]


def check_statement(stmt, macros):
    for name, val in macros.items():              # generated bodies and clobber lists come in through macros
        stmt = re.sub(r"\b%s\b" % re.escape(name), lambda _m, v=val: v, stmt)
    template, outs, ins, clob = split_sections(stmt)
    template = template.replace("\\n", "\n").replace("\\t", " ")
    clobbers = set(re.findall(r'"([^"]+)"', clob))
    problems = []
    ops = [ln.strip().split()[0] for ln in template.split("\n") if ln.strip() and not ln.strip().startswith((";", "//"))]
    if re.search(r"\bvcc(_lo|_hi)?\b", template) and "vcc" not in clobbers:
        problems.append('names vcc without a "vcc" clobber')
    scc = sorted({o for o in ops if o.startswith("s_") and o not in NO_SCC})
    if scc and "scc" not in clobbers:
        problems.append('%s write(s) SCC without an "scc" clobber' % ", ".join(scc))
    # registers named outright (not through an operand) must be declared as well
    named = set()
    for kind, lo, hi in re.findall(r"(?<![\w%\[])([sv])\[(\d+):(\d+)\]", template):
        named.update("%s%d" % (kind, i) for i in range(int(lo), int(hi) + 1))
    named.update("%s%s" % (k, n) for k, n in re.findall(r"(?<![\w%\[.])([sv])(\d+)\b(?!\s*:)", template))
    missing = sorted(named - clobbers, key=lambda r: (r[0], int(r[1:])))
    if missing:
        problems.append("names %s without declaring %s" % (", ".join(missing[:6]) + (" ..." if len(missing) > 6 else ""), "them" if len(missing) > 1 else "it"))
    for reg in ("exec", "m0"):
        if re.search(r"\b(s_\w+|v_cmpx\w*|v_readfirstlane\w*)\s+%s\b" % reg, template) and reg not in clobbers:
            problems.append("writes %s without declaring it" % reg)
    if re.search(r"\bv_cmpx", template) and "exec" not in clobbers:
        problems.append("v_cmpx writes exec without declaring it")
    return problems


def collect_macros(paths):
    macros = {}
    for p in paths:
        txt = open(p, errors="replace").read()
        for m in re.finditer(r"^#define\s+(\w+)\s+((?:.*\\\n)*.*)$", txt, re.M):
            body = m.group(2).replace("\\\n", " ")
            if '"' in body and ("ASM" in m.group(1) or "CLOBBER" in m.group(1)):
                macros[m.group(1)] = body
    return macros


def audit(dirs=None):
    dirs = dirs or [os.path.join(ROOT, "eth-tx-proof_amd", "csrc"), os.path.join(ROOT, "tools")]
    files = [f for d in dirs for ext in ("*.cuh", "*.hip", "*.h", "*.inc") for f in sorted(glob.glob(os.path.join(d, ext)))]
    macros = collect_macros(files)
    bad, n = [], 0
    for f in files:
        if f.endswith("gl64_src.inc"):
            continue                               # gl64.cuh as a string literal for hiprtc: the same text
        txt = open(f, errors="replace").read()
        for off, stmt in asm_statements(txt):
            n += 1
            for pr in check_statement(stmt, macros):
                bad.append("%s:%d: %s" % (os.path.relpath(f, ROOT), txt.count("\n", 0, off) + 1, pr))
    return n, bad


_DISASM = {}


def disassemble_code_objects(lib=None, objdump="/opt/rocm/lib/llvm/bin/llvm-objdump"):
    """llvm-objdump -d of every gfx950 code object of a library / object file / bare .co, memoised per (path, mtime): the scans below share it"""
    import subprocess
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_resources as kr
    lib = lib or os.path.join(ROOT, "eth-tx-proof_amd", "csrc", "libzkgl_hip.so")
    key = (os.path.abspath(lib), os.path.getmtime(lib))
    if key not in _DISASM:
        blobs = [open(lib, "rb").read()] if lib.endswith(".co") else [[x for x in co if isinstance(x, (bytes, bytearray))][0] for co in kr.code_objects(lib)]
        texts = []
        for blob in blobs:
            with tempfile.NamedTemporaryFile(suffix=".co") as tf:
                tf.write(blob); tf.flush()
                texts.append(subprocess.run([objdump, "-d", "--mcpu=gfx950", tf.name], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.decode(errors="replace"))
        _DISASM[key] = texts
    return _DISASM[key]


def scc_pairs_in_code_objects(lib=None, objdump="/opt/rocm/lib/llvm/bin/llvm-objdump"):
    """The same defect seen from the built code: an instruction that writes SCC -- other than the compiler's own save/restore idiom
    (s_cselect_b64 mask, -1, 0 ... s_cmp_lg_u64 mask, 0) -- between an s_add_u32 / s_sub_u32 and the s_addc_u32 / s_subb_u32 that consumes its carry.
    Returns (number of add/addc pairs seen, list of offending sites)."""
    pairs, sites = 0, []
    for asm in disassemble_code_objects(lib, objdump):
        fn, pend, dirty, saved = None, None, None, None
        for ln in asm.splitlines():
            m = re.match(r"^[0-9a-f]+ <(\S+)>:", ln)
            if m:
                fn, pend = m.group(1), None
                continue
            t = ln.strip().split("//")[0].strip()
            if not t or t.endswith(":"):
                continue
            op = t.split()[0]
            if op in ("s_add_u32", "s_sub_u32"):
                pend, dirty, saved = [t], None, None
                continue
            if pend is None:
                continue
            if op in ("s_addc_u32", "s_subb_u32"):
                pairs += 1
                if dirty:
                    sites.append("%s: %s ... %s ... %s" % (fn, pend[0], dirty, t))
                pend = None
                continue
            pend.append(t)
            if op == "s_cselect_b64" and t.replace(" ", "").endswith(",-1,0") and dirty is None:
                saved = t.split()[1].rstrip(",")                         # the compiler parks SCC in a mask ...
            elif op == "s_cmp_lg_u64" and saved and t.split()[1].rstrip(",") == saved and dirty is None:
                saved = None                                             # ... and brings it back: SCC is the carry again
            elif op.startswith("s_") and op not in NO_SCC:
                dirty = dirty or t
            if len(pend) > 400 or op in ("s_endpgm", "s_branch", "s_setpc_b64", "s_swappc_b64") or op.startswith("s_cbranch"):
                pend = None
    return pairs, sites


# ---- rules 2-3 of DESIGN.md section 9 from the built code.  On gfx950 the result registers of a matrix instruction are
# written when it finishes, and nothing in the hardware holds back a vector instruction that reads OR WRITES one of them earlier: hipcc
# pads its own instructions (GCNHazardRecognizer: for an N-pass XDL instruction N + 3 wait states before a VALU read or write of vDst),
# but instructions inside inline-asm blocks are invisible to it.  An inline-asm write into a padding register of an in-flight result was
# round 4's "second permutation of every pair is wrong".  The scan below needs no knowledge of where asm blocks are: for every v_mfma it
# walks the following instructions, counts wait states (one per instruction, s_nop N = N + 1) and reports any instruction other than a
# matrix instruction that names a register of the result before the budget is used up.
# passes of the instructions this code base uses on gfx950 (4 cycles each: v_mfma_i32_32x32x32_i8 holds the matrix pipe 32 cycles,
# tools/mfma_overlap.hip); hipcc's own first readers sit 12 wait states behind it, budget 8 + 3 = 11
MFMA_PASSES = {"32x32": 8, "16x16": 4, "4x4": 2}


def _regs(tok):
    """the VGPRs / AGPRs an operand token names: 'v5' -> {('v',5)}, 'a[0:15]' -> {('a',0)..('a',15)}"""
    out = set()
    for kind, lo, hi in re.findall(r"\b([va])\[(\d+):(\d+)\]", tok):
        out.update((kind, i) for i in range(int(lo), int(hi) + 1))
    for kind, n in re.findall(r"(?<![\w\[:])([va])(\d+)\b", tok):
        out.add((kind, int(n)))
    return out


def mfma_result_hazards(asm_text):
    """(matrix instructions seen, offending sites) of one disassembly (llvm-objdump -d)"""
    seen, sites = 0, []
    fn, lines = None, []
    for ln in asm_text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", ln)
        if m:
            fn = m.group(1)
            continue
        t = ln.strip().split("//")[0].strip()
        if t and not t.endswith(":"):
            lines.append((fn, t))
    for i, (fn, t) in enumerate(lines):
        op = t.split()[0]
        if not op.startswith("v_mfma") and not op.startswith("v_smfmac"):
            continue
        seen += 1
        shape = re.search(r"_(\d+x\d+)x\d+", op)
        passes = MFMA_PASSES.get(shape.group(1) if shape else "", 8)
        budget = passes + 3
        dst = _regs(t[len(op):].split(",")[0])
        ws, j = 0, i + 1
        while j < len(lines) and ws < budget and lines[j][0] == fn:
            u = lines[j][1]
            uop = u.split()[0]
            if uop == "s_nop":
                ws += int(u.split()[1], 0) + 1
            else:
                if uop in ("s_endpgm", "s_branch", "s_setpc_b64", "s_swappc_b64") or uop.startswith("s_cbranch"):
                    break                                               # straight-line code only (the kernels of interest are)
                if uop.startswith("v_mfma") or uop.startswith("v_smfmac"):
                    if _regs(u[len(uop):].split(",")[0]) & dst:
                        break                                           # the next link of an accumulation chain: it is checked as its own start
                elif _regs(u[len(uop):]) & dst:
                    sites.append("%s: %s  ...  %d wait state(s) later (< %d): %s" % (fn, t, ws, budget, u))
                    break
                ws += 1
            j += 1
    return seen, sites


def mfma_result_hazards_in_code_objects(lib=None, objdump="/opt/rocm/lib/llvm/bin/llvm-objdump"):
    seen, sites = 0, []
    for asm in disassemble_code_objects(lib, objdump):
        n, s = mfma_result_hazards(asm)
        seen += n; sites += s
    return seen, sites


if __name__ == "__main__":
    n, bad = audit()
    print("%d asm statements, %d problems" % (n, len(bad)))
    for b in bad:
        print("  " + b)
    if "--code" in sys.argv:
        pairs, sites = scc_pairs_in_code_objects(sys.argv[sys.argv.index("--code") + 1] if len(sys.argv) > sys.argv.index("--code") + 1 else None)
        print("%d scalar add / add-with-carry pairs in the code objects, %d with SCC overwritten in between" % (pairs, len(sites)))
        for s in sites[:20]:
            print("  " + s)
        bad += sites
        lib = sys.argv[sys.argv.index("--code") + 1] if len(sys.argv) > sys.argv.index("--code") + 1 else None
        seen, sites = mfma_result_hazards_in_code_objects(lib)
        print("%d matrix instructions in the code objects, %d with a result register named by a vector instruction inside the hazard window" % (seen, len(sites)))
        for s in sites[:20]:
            print("  " + s)
        bad += sites
    sys.exit(1 if bad else 0)
