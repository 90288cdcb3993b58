import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """Some GPU tests drive the library from torch streams and graphs.  torch ships a HIP runtime of its own: when it
    initialises after libslam_hip.so has brought up the system one it finds no device, so on a GPU box torch goes first."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure): compiled on demand with gcc."""
    from oracle import pyoracle
    pyoracle.build()
    pyoracle.lib()
    return pyoracle


@pytest.fixture(scope="session")
def ctx():
    """One GPU context for the whole GPU test session.  No skip: without the HIP library or
    without a GPU these tests must fail, not pass silently."""
    from slam_amd import api
    c = api.Context(0)
    yield c
    c.close()


# Added to the seeds of the randomised differential tests (SLAM_TEST_SEED=n python -m pytest ...): other inputs,
# same checks - for soak runs beyond the default stream.
SOAK_SEED = int(os.environ.get("SLAM_TEST_SEED", "0"))

# True when no environment knob moves work between the lean and the general kernels.  Assertions about which kernel took
# how many jobs hold for the default split only; the parity assertions hold under every knob.
DEFAULT_KERNEL_SPLIT = not any(os.environ.get(k) for k in (
    "SLAM_HIP_NO_LEAN", "SLAM_HIP_FORCE_DIRECT", "SLAM_HIP_WALK_LDS_BYTES", "SLAM_HIP_WALK_LDS_KB",
    "SLAM_HIP_SEARCH_LDS_BYTES", "SLAM_HIP_SEARCH_LDS_KB", "SLAM_HIP_MATCH_LDS_KB", "SLAM_HIP_SEARCH_LEAN_MIN"))
