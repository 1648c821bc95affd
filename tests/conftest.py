import glob
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(autouse=True)
def _few_host_threads():
    """The CPU double / oracle work on matrices of a few hundred rows: with the 256 host threads of a GPU box every small
    torch op costs milliseconds of thread hand-over (a 9-step schedule test took 370 s instead of 5).  Eight threads for
    everything; the one full-size oracle evaluation raises the count itself."""
    n = torch.get_num_threads()
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    yield
    torch.set_num_threads(n)


def golden_names():
    """The bound / gradient / predictive fixtures of make_golden.py (posterior_*.npz are the sampler's exact-moment fixtures)."""
    names = sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))
    return [n for n in names if not n.startswith("posterior_")]


def load_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def engine():
    """The one product engine.  GPU tests fail (not skip) when the HIP library cannot be used on a GPU box."""
    if not torch.cuda.is_available():
        pytest.skip("no HIP device in this container")
    import ggp_amd
    return ggp_amd.HipEngine()


def dev(a, engine):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64).to(engine.device).contiguous()
