import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def dev():
    import torch
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _oracle_threads(request):
    """Host threads for the CPU oracle.  Most parity tests evaluate it on a handful of sequences (64-row matrices): on the GPU
    box's 64 hardware threads the fork / join of every small torch op then dominates (a 1000-step fp64 walk: 400 s on 64
    threads, 18 s on 8).  Default 16; a module that runs the oracle at B = 256 sets ORACLE_THREADS = 64."""
    try:
        import torch
    except Exception:
        yield
        return
    before = torch.get_num_threads()
    want = int(getattr(request.module, "ORACLE_THREADS", 16))
    torch.set_num_threads(max(1, min(want, os.cpu_count() or 1)))
    try:
        yield
    finally:
        torch.set_num_threads(before)
