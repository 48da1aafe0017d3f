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


def pytest_sessionstart(session):
    """The library is a build artefact (git-ignored): on a clean checkout build it once (hipcc cross-compiles gfx950
    without a GPU, ~90 s).  Only when the file is ABSENT - a present library is never rebuilt behind the tests' back."""
    try:
        from quantization.mxnet_amd.csrc import build
        if not os.path.exists(build.OUT):
            build.build_library(force=True, verbose=False)
    except Exception as e:                                   # no hipcc: test_abi reports the missing library
        sys.stderr.write("conftest: could not build libfakequant.so: %s\n" % (e,))


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
