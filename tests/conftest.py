import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


# Collection order of the GPU tier (the driver runs `pytest tests/ -x -q -m gpu`: with -x a failure hides everything behind
# it, and in round 4 a two-process log-scraping test sat alphabetically in FRONT of the float64 checks of the headline
# kernels).  Parity evidence first, process-spawning tests last:
#   0  oracle / golden parity of the product path        1  kernel-vs-float64 numerics of the split-f16 kernels
#   2  remaining single-process op / optimiser / graph    3  CLI subprocesses     4  multi-rank (spawned ranks, torchrun)
_ORDER = {"test_gpu_frontend": 0, "test_gpu_model": 0, "test_gpu_ops": 0, "test_gpu_gru": 0,
          "test_gpu_sf16": 1,
          "test_gpu_optim": 2, "test_gpu_graph": 2,
          "test_gpu_cli": 3,
          "test_gpu_parallel": 4}


def collection_rank(nodeid):
    """Sort key of a test id: (tier, original position is kept by the stable sort)."""
    module = os.path.splitext(os.path.basename(nodeid.split("::")[0]))[0]
    return _ORDER.get(module, 2)


def pytest_collection_modifyitems(config, items):
    items.sort(key=lambda it: collection_rank(it.nodeid))       # stable: the order inside a module is untouched
