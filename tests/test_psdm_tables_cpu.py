"""The constant tables of the matrix-core Poseidon permutation (eth-tx-proof_amd/csrc/poseidon_mfma_tables.h) checked on the
CPU: tools/psdm_tables_check.hip emulates v_mfma_i32_32x32x32_i8 with integer loops and compares every linear layer (MDS +
round constants, the partial rounds' entry values E, the triangular feedback PR_D, the exit map) with direct modular
arithmetic on the round constants pinned by tests/golden/poseidon_rc.json.  Host code only: no GPU involved."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_matrix_core_tables_reproduce_the_linear_layers(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    exe = str(tmp_path / "psdm_tables_check")
    subprocess.check_call([hipcc, "-O2", "-std=c++17", "-Wno-unused-value", os.path.join(ROOT, "tools", "psdm_tables_check.hip"), "-o", exe],
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "OK (0 mismatches)" in out.stdout
