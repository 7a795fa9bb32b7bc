import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ref: cross-checks against the live reference tree (dev container only)")


def load_golden(name):
    import json
    import numpy as np
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    d = {k: z[k] for k in z.files}
    cfg = json.loads(str(d["cfg"]))
    sd = {k[3:]: v for k, v in d.items() if k.startswith("sd/")}
    return d, cfg, sd


@pytest.fixture(scope="session")
def golden_loader():
    return load_golden
