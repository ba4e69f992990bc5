"""Import shim: the package directory is named `eth-tx-proof_amd/` (not a valid Python identifier),
so `import eth_tx_proof_amd` loads it from there and replaces this module in sys.modules."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "eth-tx-proof_amd")
_spec = importlib.util.spec_from_file_location(
    "eth_tx_proof_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["eth_tx_proof_amd"] = _mod
_spec.loader.exec_module(_mod)
