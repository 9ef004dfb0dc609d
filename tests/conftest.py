import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def _has_gpu() -> bool:
    import ctypes

    import resnet_c_amd as R

    n = ctypes.c_int(0)
    try:
        st = R._lib.lib().rn_device_count(ctypes.byref(n))
    except Exception:
        return False
    return st == 0 and n.value > 0


def pytest_collection_modifyitems(config, items):
    # A gpu-marked test on a box without a GPU is an error of the run, not a skip:
    # the product has no CPU fallback.  Only skip when the user did not ask for gpu tests.
    if config.getoption("-m") and "gpu" in config.getoption("-m") and "not gpu" not in config.getoption("-m"):
        return
    skip = None
    for item in items:
        if "gpu" in item.keywords:
            if skip is None:
                skip = None if _has_gpu() else pytest.mark.skip(reason="no GPU in this container")
            if skip is not None:
                item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def finch():
    import numpy as np

    return np.fromfile(os.path.join(GOLDEN, "finch_224.bin"), dtype=np.float32).reshape(1, 3, 224, 224)


@pytest.fixture(scope="session")
def state50():
    import resnet_c_amd as R

    return R.weights.generate_state("resnet50", seed=0)


@pytest.fixture(scope="session")
def state152():
    import resnet_c_amd as R

    return R.weights.generate_state("resnet152", seed=0)
