import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

P = 0xFFFFFFFF00000001


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` through gpurun)")


def pytest_collection_modifyitems(config, items):
    """A GPU test that stops making progress (a wait on the device that never returns) must end the run, not sit until the harness kills the
    box: ten minutes per test through pytest-timeout's watchdog thread (stack dump, then the process exits) when the plugin is there."""
    if not config.pluginmanager.hasplugin("timeout"):
        return
    for it in items:
        if it.get_closest_marker("gpu") and not it.get_closest_marker("timeout"):
            it.add_marker(pytest.mark.timeout(600, method="thread"))


def field_rand(rng, shape):
    """uniform canonical field elements"""
    a = rng.integers(0, 1 << 64, size=shape, dtype=np.uint64)
    while True:
        bad = a >= np.uint64(P)
        if not bad.any():
            return a
        a[bad] = rng.integers(0, 1 << 64, size=int(bad.sum()), dtype=np.uint64)


EDGE = np.array([0, 1, P - 1, (1 << 32) - 1, 1 << 32, 1 << 63, P, (1 << 64) - 1], dtype=np.uint64)


@pytest.fixture(scope="session")
def orc():
    import orc as _orc
    # the checker runs small cases here: on a many-core host (the GPU boxes show 256 hardware threads) its OpenMP regions cost more in
    # team start-up than in work, so the team is capped
    _orc.lib().orc_set_threads(min(16, os.cpu_count() or 1))
    return _orc


@pytest.fixture(scope="session")
def zk():
    import eth_tx_proof_amd as E
    from eth_tx_proof_amd import lib as zl
    return zl


@pytest.fixture(scope="session")
def ctx(zk):
    c = zk.Context(0)          # raises without a GPU: there is no CPU fallback
    yield c
    c.close()
