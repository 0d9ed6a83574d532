import json
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
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    """npz -> dict of torch tensors; '__bf16' keys hold raw bf16 bit patterns."""
    z = np.load(os.path.join(GOLDEN, name))
    out = {}
    for k in z.files:
        a = z[k]
        if k.endswith("__bf16"):
            out[k[:-6]] = torch.from_numpy(a.copy()).view(torch.bfloat16)
        elif k == "cfg_json":
            out[k] = json.loads(bytes(a.tolist()).decode())
        else:
            out[k] = torch.from_numpy(a.copy()) if a.shape != () else torch.tensor(a.item())
    return out


@pytest.fixture(scope="session")
def golden():
    return load_golden


def has_gpu():
    return torch.cuda.is_available()
