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
