import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_pkg():
    """Import the package directory 'digital-subband-video-2_amd' (hyphenated, so by path) as dsv2_amd."""
    if "dsv2_amd" in sys.modules:
        return sys.modules["dsv2_amd"]
    pkgdir = os.path.join(ROOT, "digital-subband-video-2_amd")
    spec = importlib.util.spec_from_file_location("dsv2_amd", os.path.join(pkgdir, "__init__.py"),
                                                  submodule_search_locations=[pkgdir])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["dsv2_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="session")
def pkg():
    return load_pkg()
