#!/usr/bin/env python3
"""Resource usage of every gfx950 kernel inside a HIP shared library, read from the code objects' own metadata (the NT_AMDGPU_METADATA
note, what `llvm-readelf --notes` prints): VGPRs, AGPRs, SGPRs, spilled registers, scratch bytes per lane, static LDS, workgroup size.
No GPU and no LLVM tools needed: the library's .hip_fatbin section is cut into its offload bundles (one per translation unit), each
bundle's gfx950 ELF is parsed by hand, the note is msgpack.

    tools/kernel_resources.py [library.so] [--scratch] [--csv]

--scratch lists only kernels whose private_segment_fixed_size is > 0 (scratch on every launch).  Used by tests/test_abi_cpu.py:
the shipped kernels must be scratch-free (scratch + many streams is the pattern of every GPU memory fault seen)."""
import struct
import sys
import os

import msgpack

MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _elf_sections(b):
    assert b[:4] == b"\x7fELF" and b[4] == 2, "not a 64-bit ELF"
    shoff, = struct.unpack_from("<Q", b, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", b, 0x3A)
    secs = []
    for i in range(shnum):
        name, typ, flags, addr, off, size = struct.unpack_from("<IIQQQQ", b, shoff + i * shentsize)
        secs.append((name, typ, off, size))
    stro = secs[shstrndx][2]
    out = []
    for name, typ, off, size in secs:
        end = b.index(b"\0", stro + name)
        out.append((b[stro + name:end].decode(), typ, off, size))
    return out


def _notes(b, off, size):
    p, end = off, off + size
    while p + 12 <= end:
        namesz, descsz, typ = struct.unpack_from("<III", b, p)
        p += 12
        name = b[p:p + namesz].rstrip(b"\0")
        p += (namesz + 3) & ~3
        desc = b[p:p + descsz]
        p += (descsz + 3) & ~3
        yield name, typ, desc


def code_objects(path):
    """the gfx950 ELF images of every offload bundle in the library"""
    blob = open(path, "rb").read()
    secs = _elf_sections(blob)
    fat = [(o, s) for n, t, o, s in secs if n == ".hip_fatbin"]
    assert fat, "no .hip_fatbin section in %s" % path
    off, size = fat[0]
    fb = blob[off:off + size]
    out, p = [], fb.find(MAGIC)
    while p >= 0:
        n, = struct.unpack_from("<Q", fb, p + len(MAGIC))
        q = p + len(MAGIC) + 8
        for _ in range(n):
            eo, es, ts = struct.unpack_from("<QQQ", fb, q)
            triple = fb[q + 24:q + 24 + ts].decode()
            q += 24 + ts
            if "amdgcn" in triple and es:
                out.append((triple, fb[p + eo:p + eo + es]))
        p = fb.find(MAGIC, p + len(MAGIC))
    return out


def kernels(path):
    """[{name, vgpr, agpr, sgpr, vgpr_spill, sgpr_spill, scratch, lds, wg}] over all code objects of the library"""
    res = []
    for triple, elf in code_objects(path):
        for name, typ, off, size in _elf_sections(elf):
            if typ != 7:           # SHT_NOTE
                continue
            for nname, ntyp, desc in _notes(elf, off, size):
                if nname != b"AMDGPU" or ntyp != 32:
                    continue
                md = msgpack.unpackb(desc, raw=False, strict_map_key=False)
                for k in md.get("amdhsa.kernels", []):
                    res.append({"name": k.get(".name", ""), "vgpr": k.get(".vgpr_count", 0), "agpr": k.get(".agpr_count", 0),
                                "sgpr": k.get(".sgpr_count", 0), "vgpr_spill": k.get(".vgpr_spill_count", 0),
                                "sgpr_spill": k.get(".sgpr_spill_count", 0), "scratch": k.get(".private_segment_fixed_size", 0),
                                "lds": k.get(".group_segment_fixed_size", 0), "wg": k.get(".max_flat_workgroup_size", 0),
                                "dynamic_stack": bool(k.get(".uses_dynamic_stack", False)), "arch": triple})
    return res


def demangle(names):
    import subprocess
    for tool in ("/opt/rocm/lib/llvm/bin/llvm-cxxfilt", "c++filt"):
        try:
            out = subprocess.run([tool], input="\n".join(names), capture_output=True, text=True, check=True).stdout.split("\n")
            return out[:len(names)]
        except Exception:
            continue
    return list(names)


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    path = args[0] if args else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "eth-tx-proof_amd", "csrc", "libzkgl_hip.so")
    ks = kernels(path)
    if "--scratch" in sys.argv:
        ks = [k for k in ks if k["scratch"] > 0 or k["dynamic_stack"]]
    names = demangle([k["name"] for k in ks])
    sep = "," if "--csv" in sys.argv else "  "
    print(sep.join(["vgpr", "agpr", "sgpr", "vspill", "sspill", "scratch_B", "lds_B", "wg", "kernel"]))
    for k, nm in sorted(zip(ks, names), key=lambda kn: (-kn[0]["scratch"], kn[1])):
        print(sep.join(str(x) for x in (k["vgpr"], k["agpr"], k["sgpr"], k["vgpr_spill"], k["sgpr_spill"], k["scratch"], k["lds"], k["wg"], nm[:150])))
    print("# %d kernels, %d with scratch" % (len(ks), sum(1 for k in ks if k["scratch"] > 0)), file=sys.stderr)


if __name__ == "__main__":
    main()
