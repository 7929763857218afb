import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an MI355X (runs the HIP kernels through the C ABI)")


def load_golden(name):
    """Returns (case dict of numpy arrays, weights dict of torch tensors)."""
    import torch
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    case = {k: z[k] for k in z.files}
    wname = str(case["meta/weights"])
    wpath = os.path.join(GOLDEN, wname)
    if wname.endswith(".pt"):
        sd = torch.load(wpath, map_location="cpu", weights_only=True)["model_state_dict"]
        weights = {k: v.float() for k, v in sd.items()}
    else:
        wz = np.load(wpath + ".npz")
        weights = {k: torch.from_numpy(wz[k]) for k in wz.files}
    return case, weights


@pytest.fixture(scope="session")
def device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
