import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_cases(name):
    """tests/golden/<name>.npz written by oracle/gen_golden.py -> list of dicts."""
    z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    n = int(z["n"])
    cases = [dict() for _ in range(n)]
    for k in z.files:
        if k == "n":
            continue
        idx, field = k.split("_", 1)
        v = z[k]
        cases[int(idx[1:])][field] = v.item() if v.ndim == 0 else v
    return cases


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _release_gpu_memory_between_tests(request):
    """GPU tests build whole networks, batch-64 activations and hipGraph memory pools; objects caught in reference cycles
    (captured graphs <-> closures) outlive their test until the collector runs.  Collect and hand the cached blocks back
    after every GPU test so that a later full-size test does not start on a device that is mostly reserved."""
    yield
    if request.node.get_closest_marker("gpu") is not None:
        import gc
        import torch
        gc.collect()
        if torch.cuda.is_available():
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
