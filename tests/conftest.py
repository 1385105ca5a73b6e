import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu through gpurun)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(autouse=True)
def _contracts_back_to_default():
    """The oracle's contraction contract and the HIP library's engine are process-wide switches: whatever a test selected, the next
    test starts from the defaults (both sides: FASTKV_CONTRACTION, fmaf unless set)."""
    yield
    import sys as _sys
    orc = _sys.modules.get("oracle.fastkv_oracle")
    if orc is not None and orc._lib is not None:
        from helpers import default_contraction
        orc.set_contraction(default_contraction())
        orc.set_softmax("contract")                              # (the reference-order modes are for the stage-level pins only)
    ops = _sys.modules.get("fastkv_amd.ops")
    if ops is not None:
        ops._engine = ops.ENGINE[os.environ.get("FASTKV_SCORE_ENGINE", "auto")]
