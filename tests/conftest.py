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


def golden_names():
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))


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
