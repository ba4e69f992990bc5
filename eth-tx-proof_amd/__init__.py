"""MI355X-native STARK prover backend: host-side Python harness over the C ABI in include/zkgl.h.

The product is the HIP shared library `eth-tx-proof_amd/csrc/libzkgl_hip.so`; this package is the thin ctypes
binding plus the mirror of the reference's operator surface (ops.TxProof, /root/reference/ops/src/lib.rs:22-62).
There is no CPU fallback: loading or creating a context without the HIP library / a GPU raises.
"""
from . import _abi, air  # noqa: F401
from ._abi import P, COSET_SHIFT, standard_fast_config, standard_recursion_config  # noqa: F401
