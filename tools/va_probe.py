"""Debug probe: what lives in the first megabytes above each 4 GiB-aligned address of this process's GPU-visible address space?

The three unexplained "Memory access fault by GPU" reports of rounds 3 and 4 name page-aligned addresses 2.07 / 2.20 / 2.49 MB above a
4 GiB boundary (0x78a200213000, 0x723900232000, 0x72bb0027d000).  This probe, called from a live process (bench.py with ZK_VA_PROBE=1, in the
middle of the transaction record's timed region), prints /proc/self/maps for those windows and asks the HIP runtime page by page what it
knows about them (hipPointerGetAttributes / hipMemGetAddressRange): device block, pinned host block, or memory HIP does not know
(= owned by the layers below it: queues, signals, kernel-argument pools, code objects, scratch).  Together with ZK_ALLOC_TRACE=1 (the
library logs every allocation and release it makes) this names the owner of the faulting window.  Output: stderr."""
import ctypes
import sys

WINDOW = 0x1000000        # first 16 MiB above a 4 GiB boundary


class _Attr(ctypes.Structure):
    _fields_ = [("type", ctypes.c_int), ("device", ctypes.c_int), ("devicePointer", ctypes.c_void_p), ("hostPointer", ctypes.c_void_p),
                ("isManaged", ctypes.c_int), ("allocationFlags", ctypes.c_uint)]


def probe(tag=""):
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipPointerGetAttributes.argtypes = [ctypes.POINTER(_Attr), ctypes.c_void_p]
    hip.hipMemGetAddressRange.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t), ctypes.c_void_p]
    out = sys.stderr
    maps = []
    for ln in open("/proc/self/maps"):
        f = ln.split()
        a, b = (int(x, 16) for x in f[0].split("-"))
        maps.append((a, b, f[1], f[5] if len(f) > 5 else ""))
    floors = sorted({a & ~0xffffffff for a, b, _, name in maps if a >= 0x700000000000 and (a & 0xffffffff) < WINDOW and ("kfd" in name or "render" in name or name == "")})
    print("[va_probe %s] %d mappings, %d windows" % (tag, len(maps), len(floors)), file=out)
    for fl in floors:
        rows = [(a, b, p, n) for a, b, p, n in maps if a < fl + WINDOW and b > fl]
        if not any("kfd" in n or "render" in n for _, _, _, n in rows):
            continue
        print("[va_probe] window 0x%x" % fl, file=out)
        for a, b, p, n in rows:
            print("[va_probe]   map +0x%07x .. +0x%07x %s %s" % (a - fl, b - fl, p, n), file=out)
        run = None
        for addr in range(fl, fl + WINDOW, 0x1000):
            at = _Attr()
            rc = hip.hipPointerGetAttributes(ctypes.byref(at), ctypes.c_void_p(addr))
            base = ctypes.c_void_p(); size = ctypes.c_size_t()
            rc2 = hip.hipMemGetAddressRange(ctypes.byref(base), ctypes.byref(size), ctypes.c_void_p(addr))
            key = (rc, at.type if rc == 0 else -1, at.allocationFlags if rc == 0 else 0, (base.value or 0) - fl if rc2 == 0 else None, size.value if rc2 == 0 else None)
            if run and run[0] == key:
                run[2] = addr + 0x1000
            else:
                if run:
                    _emit(out, fl, run)
                run = [key, addr, addr + 0x1000]
        if run:
            _emit(out, fl, run)
    out.flush()


def _emit(out, fl, run):
    (rc, ty, flags, base, size), a, b = run
    if rc != 0 and base is None:
        what = "unknown to HIP"
    else:
        what = "type %s flags 0x%x block +0x%x size %s" % ({0: "unregistered", 1: "host", 2: "device", 3: "managed", 4: "array"}.get(ty, ty), flags, base if base is not None else -1, size)
    print("[va_probe]   hip +0x%07x .. +0x%07x %s" % (a - fl, b - fl, what), file=out)
