import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped (not failed) when no device is visible, e.g. `pytest tests/` here.  The bench rehearsals run FIRST: they put up to
    eight bench.py ranks on the one GPU, whose exchange kernels wait for each other on the device -- with the pytest process itself holding a
    ninth GPU context (any in-process GPU test before them) the driver time-slices whole processes and the ranks' bounded waits run out
    (5 of 7 in-suite runs failed that way on the round-5 boxes, 0 of 5 with the rehearsals first or alone)."""
    first = [it for it in items if it.fspath.basename == "test_bench_gpu.py"]
    if first:
        items[:] = first + [it for it in items if it.fspath.basename != "test_bench_gpu.py"]
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:  # pragma: no cover
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
