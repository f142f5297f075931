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


@pytest.fixture(autouse=True)
def _gpu_tests_need_the_reference(request):
    """The `-m gpu` tests are parity tests: they compare libdsv2hip.so with the real reference compiled into oracle/_ref (which
    travels to the GPU box like the library itself).  Without it they must FAIL -- a skip would read as green on a box whose
    snapshot lost the checker."""
    if request.node.get_closest_marker("gpu") is not None:
        import dsvabi as A
        missing = [p for p in (A.REF_SO,) if not os.path.exists(p)]
        if request.node.fspath.basename == "test_gpu_cli.py":
            missing += [p for p in (A.REF_CLI, os.path.join(A.ROOT, "oracle", "_ref", "dsv2_dropin")) if not os.path.exists(p)]
        if missing:
            pytest.fail("oracle/_ref is not built (%s): run `python __graft_entry__.py` where /root/reference exists; the GPU parity "
                        "tests have nothing to compare with" % ", ".join(os.path.relpath(m, A.ROOT) for m in missing), pytrace=False)
