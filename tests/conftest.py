import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The library replays hipGraph batches only on instances large enough to pay for the capture; the tests'
    # instances are tiny, and the graph route is the one they are meant to exercise (SDPLR_HIP_NO_GRAPH=1 in
    # test_code_paths_agree selects the eager route explicitly).
    os.environ.setdefault("SDPLR_HIP_FORCE_GRAPH", "1")


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
