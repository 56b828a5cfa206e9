import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as graft  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def pkg():
    return graft.load_package()


@pytest.fixture(scope="session")
def fo():
    return graft.load_oracle()


@pytest.fixture(scope="session")
def wl(pkg):
    import importlib
    return importlib.import_module(graft.PKG_NAME + ".workloads")


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(ROOT, "tests", "golden", "golden.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def lib(pkg):
    if not os.path.exists(pkg.library_path()):
        graft.build()
    return pkg.load_library()


@pytest.fixture(scope="session")
def gpu(lib):
    if lib.feasthip_device_count() < 1:
        pytest.fail("no gfx950 device: GPU tests must run on the MI355X box (there is no CPU fallback)")
    return True
