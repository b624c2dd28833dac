import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    os.environ.setdefault("VSM_POISON_WORK", "1")   # host layer fills work buffers with NaN before handing them to the library
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _poison_lds_before_gpu_tests(request):
    """GPU tests start from an LDS full of NaNs on every CU: a kernel that reads LDS it has not written (rows >= N of a
    vector, a padding column) fails here instead of depending on what the previous test left behind."""
    if request.node.get_closest_marker("gpu") is None:
        yield
        return
    import torch
    if torch.cuda.is_available():
        import vsmartmom_jl_amd as vsm
        vsm._lib.check(vsm._lib.lib().vsm_test_poison_lds(None))
        torch.cuda.synchronize()
    yield
