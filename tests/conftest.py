import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def cuda():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU is visible — the HIP path must run, there is no fallback")
    return torch.device("cuda:0")


GEMM_MODES = (6, 0)   # the shipped split-bf16 mode (6 plane products) and the fp32 MFMA mode; same tolerances in both


@pytest.fixture(params=GEMM_MODES, ids=lambda m: {6: "split6", 9: "split9", 0: "fp32mfma"}[m])
def gemm_mode(request):
    """Runs the test once per GEMM mode of mit_conv_gemm (include/mit_hip.h, mit_gemm_mode_set).  Engines the test builds pack
    their weights in that mode; module-scoped engines are built in the shipped mode (``shipped_mode``) and follow the switch."""
    from manga_image_translator_amd import ops

    prev = ops.set_split_mode(request.param)
    yield request.param
    ops.set_split_mode(prev)


@pytest.fixture(scope="session")
def shipped_mode():
    """Context-manager factory for module-scoped engine fixtures: ``with shipped_mode(): build`` packs the weights with their
    split planes whatever mode the first test using the fixture happens to run in."""
    from manga_image_translator_amd import ops

    return lambda: ops.gemm_mode(6)


_ORACLE_MEMO = {}


@pytest.fixture(scope="session")
def oracle_memo():
    """Cache for CPU-oracle results shared by the runs of one test in the two GEMM modes (the oracle does not depend on the mode):
    ``oracle_memo(key, fn)`` -> fn() computed once per key."""
    def get(key, fn):
        if key not in _ORACLE_MEMO:
            _ORACLE_MEMO[key] = fn()
        return _ORACLE_MEMO[key]

    return get
