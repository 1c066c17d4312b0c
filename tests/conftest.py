import os
import sys

# The CPU suite shares its cores with whatever else runs on the box: OpenMP workers that spin while they wait turn contention into
# minutes.  Measured here with eight busy processes beside the suite: 1 087 s spinning (the 1 020 s of round 5's review), 310 s with
# passive waiting; on idle cores 115 - 140 s against 180 s.  (bench.py's cpu_baseline runs in its own process with the defaults.)
for _k, _v in (("OMP_WAIT_POLICY", "PASSIVE"), ("GOMP_SPINCOUNT", "0"), ("KMP_BLOCKTIME", "0")):
    os.environ.setdefault(_k, _v)

import numpy as np  # noqa: E402
import pytest  # noqa: E402

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    return np.load(os.path.join(GOLDEN, "golden_esm.npz"))


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def lib():
    """libpgmi.so, built on demand (hipcc cross-compiles without a GPU)."""
    from proteingym_amd import build_native, _lib
    build_native.build(verbose=False)
    return _lib.load()


@pytest.fixture
def gemm_option(lib):
    """Sets a test hook of the GEMM launchers on the live library (pgmi_set_option) and restores the defaults afterwards."""
    def set_(name, value):
        from proteingym_amd import _lib
        _lib.check(lib.pgmi_set_option(name.encode(), int(value)))
    yield set_
    lib.pgmi_set_option(b"gemm_half_tail", -1)           # -1: back to the environment / the default (an explicit value outlives model creation)
    lib.pgmi_set_option(b"gemm_max_rows", -1)
