import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            with np.load(os.path.join(GOLDEN, name + ".npz")) as z:
                cache[name] = {k: z[k] for k in z.files}
        return cache[name]
    return load


@pytest.fixture(scope="module")
def gpu():
    """mx.gpu(0) on a real device; GPU-marked tests are skipped without one."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from quantization.mxnet_amd import mx
    return mx.gpu(0)
