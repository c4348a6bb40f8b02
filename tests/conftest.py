import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: takes more than a few seconds")


@pytest.fixture(scope="session")
def pkg():
    """The product package (directory name contains a hyphen, hence importlib)."""
    return importlib.import_module("eradiate-kernel_amd")


@pytest.fixture(scope="session")
def oracle():
    import tests.oracle_binding as ob
    ob.lib()
    return ob


@pytest.fixture(scope="session")
def gpu_rgb(pkg):
    """variant fixture in the spirit of /root/reference/src/conftest.py:35-90: skip when no GPU is visible."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    pkg.set_variant("gpu_rgb")
    return pkg
