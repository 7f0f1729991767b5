import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with gpurun)")


def _hip_device_visible():
    """True when this host exposes an AMD GPU to user space (the KFD node).  Deliberately NOT a probe
    through libkpl: on a GPU box a missing or broken libkpl.so must make the gpu tests FAIL, not skip."""
    return os.path.exists("/dev/kfd")


def pytest_collection_modifyitems(config, items):
    if _hip_device_visible():
        return
    skip = pytest.mark.skip(reason="no HIP device on this host (/dev/kfd absent): gpu tests run under gpurun")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def kpl():
    """The product binding (keypoint-learning_amd).  Import fails loudly if libkpl.so is missing."""
    return importlib.import_module("keypoint-learning_amd")


@pytest.fixture(scope="session")
def oracle():
    from oracle import kplo
    kplo.lib()
    return kplo


@pytest.fixture(scope="session")
def cases():
    from tests import helpers
    return helpers
