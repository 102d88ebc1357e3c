import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # torch's own DataLoader pin-memory thread warns once per batch about an argument torch itself passes (14 000 lines per suite run)
    config.addinivalue_line("filterwarnings", "ignore:The argument 'device' of Tensor:DeprecationWarning")
    # The GPU boxes show 256 hardware threads and grant the container 16 CPUs of cgroup quota: a 256-thread OpenMP team under
    # that quota is throttled in bursts (the CPU oracle's forward, and every rank subprocess's start-up, ran several times
    # slower than they need to).  Set before torch is imported, inherited by the rank subprocesses.
    from tise_toolbox_amd.hostinfo import usable_cpus
    os.environ.setdefault("OMP_NUM_THREADS", str(usable_cpus()))


def pytest_collection_modifyitems(config, items):
    """GPU tests must FAIL, not skip, when selected on a box without a GPU or without the HIP
    library -- a silent skip would hide 'native code not loaded'."""
    return


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def cuda_device():
    import torch
    assert torch.cuda.is_available(), "GPU test selected but no HIP device is visible"
    from tise_toolbox_amd import _lib
    _lib.load()
    return torch.device("cuda", 0)
