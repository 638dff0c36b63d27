import os
import sys
import subprocess
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


def pytest_unconfigure(config):
    # The run is over and reported. If tearing the process down wedges (device runtimes unload with helper threads around), let
    # the kernel end it after three minutes rather than hang whoever waits for it: SIGALRM's default action terminates.
    import signal
    if hasattr(signal, "alarm"):
        signal.signal(signal.SIGALRM, signal.SIG_DFL)
        signal.alarm(180)


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
