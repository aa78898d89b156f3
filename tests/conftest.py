import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_abi():
    from oracle import oracle
    return oracle.abi()


@pytest.fixture(scope="session")
def hip_abi():
    import shutil
    import sdplrplus_jl_amd as sj
    if not os.path.exists(sj.hip_library_path()) and shutil.which("hipcc"):
        # a fresh checkout: compile the product library (hipcc cross-compiles gfx950 without a GPU);
        # on the GPU box the prebuilt .so travels with the snapshot
        import __graft_entry__
        __graft_entry__.build()
    return sj.load_hip()  # raises if the HIP library is not built: no fallback
