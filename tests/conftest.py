"""pytest configuration: `gpu` marker, import helpers, oracle build."""

import importlib
import sys
from pathlib import Path

import os

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    """The product package (its directory name starts with a digit -> importlib)."""
    return importlib.import_module("21cmfast_amd")


@pytest.fixture(scope="session")
def oracle():
    """CPU oracle (test infrastructure only)."""
    mod = importlib.import_module("oracle.oracle")
    mod.load()
    return mod


@pytest.fixture(scope="session")
def gpu_lib(pkg):
    """The HIP library on a box with a GPU; fails loudly otherwise (no CPU fallback)."""
    return pkg.load(require_gpu=True)
