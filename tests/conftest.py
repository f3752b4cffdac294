import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


class Golden:
    """Lazy reader of one tests/golden/*.npz; `case(prefix)` returns torch tensors."""

    def __init__(self, name):
        self.z = np.load(os.path.join(GOLDEN, name + ".npz"))

    def keys(self):
        return list(self.z.keys())

    def __getitem__(self, k):
        return torch.from_numpy(self.z[k])

    def case(self, prefix):
        pre = prefix + ":"
        return {k[len(pre):]: torch.from_numpy(self.z[k]) for k in self.z.keys()
                if k.startswith(pre)}


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = Golden(name)
        return cache[name]

    return get


def pytest_sessionfinish(session, exitstatus):
    """Leave the achieved parity errors of a GPU run behind (tests/parity.py)."""
    import json

    import parity
    if not parity.RECORDS:
        return
    out = os.environ.get("FZ_PARITY_OUT", os.path.join(ROOT, "gpurun_out", "parity.json"))
    os.makedirs(os.path.dirname(out), exist_ok=True)
    worst = {}
    for r in parity.RECORDS:
        if "rel_err" in r and r["rel_err"] > worst.get(r["test"], {}).get("rel_err", -1.0):
            worst[r["test"]] = r
    with open(out, "w") as f:
        json.dump({"device": torch.cuda.get_device_name(0) if torch.cuda.is_available() else "cpu",
                   "n_comparisons": len(parity.RECORDS),
                   "worst_rel_err": max((r["rel_err"] for r in parity.RECORDS if "rel_err" in r), default=0.0),
                   "worst_per_test": worst, "records": parity.RECORDS}, f, indent=1)
