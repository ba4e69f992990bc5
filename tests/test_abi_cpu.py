"""CPU suite: the C-ABI library loads and exports every symbol include/zkgl.h declares; no compute without a GPU."""
import ctypes as C

import pytest


def test_library_exports_every_declared_symbol(zk):
    L = zk.lib()
    names = zk.declared_symbols()
    assert len(names) >= 30
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing


def test_no_cpu_fallback(zk):
    """Without a HIP device the product refuses to create a context (ZK_ERR_NO_DEVICE) -- it never computes on the host."""
    import torch
    if torch.cuda.device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(zk.ZkError) as e:
        zk.Context(0)
    assert e.value.code == -7


def test_null_arguments_are_rejected(zk):
    L = zk.lib()
    assert L.zk_ctx_create(C.c_int32(0), None, None) == -1
    assert L.zk_ctx_destroy(None) == -1
    assert L.zk_batch_free(None) == -1
    assert L.zk_ntt_batch(None, None, C.c_uint32(3), C.c_uint32(1), C.c_int32(0), C.c_uint64(0)) == -1


def test_standard_fast_config(zk):
    from eth_tx_proof_amd._abi import StarkConfig, standard_fast_config
    c = StarkConfig()
    zk.lib().zk_stark_config_standard_fast(C.byref(c))
    ref = standard_fast_config()
    assert [getattr(c, f) for f, _ in c._fields_] == [getattr(ref, f) for f, _ in ref._fields_] == [100, 2, 1, 4, 16, 4, 5, 84]


def _build_ops_smoke(tmp_path):
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "eth-tx-proof_amd", "csrc")
    exe = str(tmp_path / "ops_smoke")
    subprocess.check_call(["g++", "-std=c++17", "-O1", os.path.join(root, "tests", "cpp", "ops_smoke.cpp"), "-o", exe,
                           "-L" + csrc, "-lzkgl_hip", "-Wl,-rpath," + csrc, "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_cpp_operator_mirror_builds_and_fails_loudly_without_gpu(zk, tmp_path):
    """ops.hpp compiles against include/zkgl.h and links the product library; without a GPU TxProof::execute raises
    FatalError (exit code 1) -- there is no host fallback behind the operator either."""
    import subprocess, torch
    exe = _build_ops_smoke(tmp_path)
    rc = subprocess.run([exe], capture_output=True, text=True)
    if torch.cuda.device_count() == 0:
        assert rc.returncode == 1 and "FatalError" in rc.stdout
    else:
        assert rc.returncode == 0, rc.stdout


def test_build_time_constraint_kernels_are_what_the_generator_emits(zk):
    """csrc/prover_quotient_aot.hip (compiled with the library) must hold, verbatim, the kernel source the library's own generator
    (zk_quotient_kernel_source, host only) emits for each build-time AIR shape -- so build-time and hiprtc kernels cannot drift apart,
    and a change of the generator without `python tools/gen_aot_quotient.py` is caught here."""
    import os
    from eth_tx_proof_amd import air
    L = zk.lib()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "eth-tx-proof_amd", "csrc", "prover_quotient_aot.hip")).read()
    prog = air.cpu_like_air(93, 19, 2)
    ln, h = C.c_size_t(0), C.c_uint64(0)
    assert L.zk_quotient_kernel_source(prog.ref, C.c_uint32(2), C.c_uint32(1), None, C.c_size_t(0), C.byref(ln), C.byref(h)) == 0
    buf = C.create_string_buffer(ln.value + 1)
    assert L.zk_quotient_kernel_source(prog.ref, C.c_uint32(2), C.c_uint32(1), buf, C.c_size_t(ln.value + 1), C.byref(ln), C.byref(h)) == 0
    src = buf.value.decode()
    assert "quotient_aot_%016x" % h.value in src
    assert src in text
    assert "0x%016xULL" % h.value in text           # registry entry
    # too small a buffer is refused, not overrun
    small = C.create_string_buffer(16)
    assert L.zk_quotient_kernel_source(prog.ref, C.c_uint32(2), C.c_uint32(1), small, C.c_size_t(16), C.byref(ln), C.byref(h)) != 0
