import os
import sys
import subprocess
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


def pytest_sessionfinish(session, exitstatus):
    # the library's service threads are joined and its device memory handed back before the interpreter unloads the HIP runtime
    # (j40hip_shutdown); nothing else is needed for a clean exit
    import gc
    gc.collect()
    try:
        import j40_amd
        j40_amd.shutdown()
    except Exception:
        pass


def pytest_collection_modifyitems(config, items):
    # tests not marked gpu must pass on a CPU-only box
    pass


@pytest.fixture(scope="session")
def built():
    """builds the product library, the generator and the checkers once per session"""
    import __graft_entry__
    __graft_entry__.build()
    return True


@pytest.fixture(scope="session")
def ref(built):
    from refdec import Ref, REF_SO
    if not os.path.exists(REF_SO):
        pytest.skip("oracle/_ref/libj40ref.so is not available")
    return Ref()
