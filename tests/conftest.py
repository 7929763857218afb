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


_CHILDREN = None


def pytest_sessionstart(session):
    """The two-rank rehearsal (tests/test_gpu_multirank.py) runs in child processes that must be started by a process
    that has not touched the GPU yet: they are started here, when a session that will run GPU tests begins, and run
    beside the other tests (three small processes on the card).  torch.cuda.device_count() does not initialise the
    GPU."""
    global _CHILDREN
    expr = session.config.getoption("markexpr", "") or ""
    if "not gpu" in expr or session.config.getoption("collectonly", False):
        return
    try:
        import torch
        if torch.cuda.device_count() < 1:
            return
    except Exception:
        return
    import socket
    import subprocess
    import tempfile
    out_dir = tempfile.mkdtemp(prefix="dgl_multirank_")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    child = os.path.join(ROOT, "tests", "multirank_child.py")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PYTHONUNBUFFERED="1")
    procs = {}
    for name, ranks in (("one", 1), ("two", 2)):
        for r in range(ranks):
            tag = name if ranks == 1 else f"{name}_rank{r}"
            log = open(os.path.join(out_dir, tag + ".log"), "w")
            procs[tag] = subprocess.Popen(["timeout", "-k", "10", "420", sys.executable, child, out_dir],
                                          env=dict(env, RANK=str(r), WORLD_SIZE=str(ranks)), stdout=log,
                                          stderr=subprocess.STDOUT)
    _CHILDREN = (out_dir, procs)


def pytest_sessionfinish(session, exitstatus):
    if _CHILDREN is not None:
        for p in _CHILDREN[1].values():
            if p.poll() is None:
                p.kill()


@pytest.fixture(scope="session")
def multirank_children():
    if _CHILDREN is None:
        pytest.skip("the two-rank children were not started (no GPU when the session began)")
    return _CHILDREN


@pytest.fixture
def ieee_roots(device):
    """Correctly rounded square roots for the duration of a test (the engines default to the model of the reference's
    torch build, dgl_amd/sqrt_model.py): for tests that compare bit for bit with the C / PyTorch oracles, which restate
    the reference's algorithm with correctly rounded roots, and for the fused decode kernel, which is built for them."""
    from dgl_amd import sqrt_model
    sqrt_model.use("ieee", device)
    yield sqrt_model
    sqrt_model.use(sqrt_model.default_model(), device)


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
