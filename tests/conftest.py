import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _cpu_share() -> int:
    """Host threads this process may actually run: the cgroup quota (a 1-GPU box: 16 of the machine's 256 logical CPUs), the
    affinity mask, the CPU count -- whichever is smallest.  torch sizes its intra-op pool by the machine's count; throttled down to
    the quota, the oracle's CPU passes (the checkers of the parity tests) then run ~10x slower than on 16 threads."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return max(1, n)


# before torch is imported anywhere (OpenMP reads it at load time); inherited by the spawned ranks and the fresh-process tests
os.environ.setdefault("OMP_NUM_THREADS", str(_cpu_share()))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    try:
        import torch
        torch.set_num_threads(min(torch.get_num_threads(), _cpu_share()))
    except Exception:
        pass


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped (not failed) when no GPU is visible, so a plain `pytest tests/` works
    on the CPU container; `-m gpu` on the GPU box runs them."""
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
