import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "deep-q-learning_tron_amd")
GOLDEN = os.path.join(ROOT, "tests", "golden")

# repo root (for `import oracle`) and the source root that mirrors the
# reference's Deep-Q-learning_TRON/ directory (for `import tron`, `config`, `Net`)
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def load_golden(name):
    import numpy as np
    return np.load(os.path.join(GOLDEN, name + ".npz"))
