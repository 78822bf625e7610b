import os
import sys

import pytest

try:  # load torch's HIP runtime before libmkckks_hip.so (see ppqsflhe_amd/binding.py: load_library)
    import torch  # noqa: F401
except Exception:
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def ref_ctx():
    """Oracle context at the reference's own parameters (CC.json: N=2^14, depth 2, 40-bit scaling, dnum 2)."""
    from oracle.oracle import OracleContext
    return OracleContext(14, 2, 40, 60, dnum=2, aux_bits=60, extra_bits=20)


@pytest.fixture(scope="session")
def small_ctx():
    """Small oracle context for exhaustive checks (N=2^10, 5 Q limbs, dnum 2)."""
    from oracle.oracle import OracleContext
    return OracleContext(10, 3, 40, 60, dnum=2, aux_bits=60, extra_bits=20)
