import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def _usable_cpus():
    """Affinity mask capped by the cgroup CPU quota (the GPU box shows 256 cores but grants a 16-CPU share)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, round(int(q) / int(per))))
    except (OSError, ValueError):
        pass
    return n


# the checker's OpenMP loops default to one thread per visible core: do not oversubscribe a CPU quota
os.environ.setdefault("OMP_NUM_THREADS", str(_usable_cpus()))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    import __graft_entry__ as g
    return g.load_package()


@pytest.fixture(scope="session")
def O():
    """The CPU oracle (checker only)."""
    import __graft_entry__ as g
    return g.load_oracle()


@pytest.fixture(scope="session")
def data(O):
    return O.MMData.load(os.path.join(GOLDEN, "mm_data.npz"))


@pytest.fixture(scope="session")
def golden_run():
    return np.load(os.path.join(GOLDEN, "mm_ref_run_n1000.npz"))


@pytest.fixture(scope="session")
def known_answers():
    return np.load(os.path.join(GOLDEN, "mm_known_answers.npz"))


@pytest.fixture(scope="session")
def prior_pdf_golden():
    return np.load(os.path.join(GOLDEN, "mm_prior_pdf.npz"))


def relerr(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b) / np.maximum(1.0, np.abs(b))
