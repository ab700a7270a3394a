import faulthandler
import os
import signal
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
TESTS = os.path.join(ROOT, "tests")
if TESTS not in sys.path:
    sys.path.insert(0, TESTS)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# single box: rendezvous and gloo pairs over loopback, bounded group timeouts (fiber_amd/parallel.py reads the last one)
os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
os.environ.setdefault("FIBER_DIST_TIMEOUT", "120")

# torch imports sympy (torch.fx.experimental.symbolic_shapes) lazily inside the first Tensor.backward(gradient) of a process.  On a box
# whose image is still paging in that import alone has taken minutes (round 6: it ran into the per-test alarm below, was left half
# imported, and every later backward failed with "module 'sympy' has no attribute 'printing'"): pay for it here, at collection time.
try:
    import torch  # noqa: F401
    import torch.fx.experimental.symbolic_shapes  # noqa: F401
except Exception:                                          # (a CPU-only or trimmed install: nothing to pre-load)
    pass

# The oracle's many small CPU ops get SLOWER with more threads (profiles/r06_summary.md section 6: one FIBER-Base step 12 s on 32 threads,
# 21 s on 64, 47 s on 128): on a many-core GPU host the suite's oracle work runs on 16.
try:
    if (os.cpu_count() or 1) > 16 and "OMP_NUM_THREADS" not in os.environ:
        torch.set_num_threads(16)
except Exception:
    pass

# No single test may take longer than this (seconds).  SIGALRM fails the test and the run goes on; a test stuck inside native
# code that never returns to the interpreter is ended by faulthandler (stack dump + exit) a little later.
TEST_LIMIT_S = int(os.environ.get("FIBER_TEST_LIMIT", "420"))

# Collection order of the GPU files: the hot path (ops -> blocks / path -> streams -> trainer) first, the
# (f)-row files after it, multi-process tests last -- a problem in an out-of-core-scope file cannot hide a hot-path result.
ORDER = ["test_hip_ops", "test_hip_modules", "test_hip_stream", "test_hip_trainer", "test_input_pipeline", "test_hip_fusion",
         "test_hip_dcn", "test_hip_graph", "test_hip_ddp"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    def key(item):
        name = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        if name in ORDER:
            return (1, ORDER.index(name))
        return (0, 0)                       # CPU files keep their place ahead of the GPU files
    items.sort(key=key)                     # stable: order inside a file is unchanged


class _TestTimeout(Exception):
    pass


@pytest.fixture(autouse=True)
def _watchdog(request):
    def on_alarm(signum, frame):
        raise _TestTimeout(f"{request.node.nodeid}: exceeded {TEST_LIMIT_S} s")

    use_alarm = hasattr(signal, "SIGALRM")
    if use_alarm:
        old = signal.signal(signal.SIGALRM, on_alarm)
        signal.alarm(TEST_LIMIT_S)
    faulthandler.dump_traceback_later(TEST_LIMIT_S + 60, exit=True)
    try:
        yield                               # (the alarm raises inside the test body: the test fails with its own stack)
    finally:
        faulthandler.cancel_dump_traceback_later()
        if use_alarm:
            signal.alarm(0)
            signal.signal(signal.SIGALRM, old)


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))
    return load
