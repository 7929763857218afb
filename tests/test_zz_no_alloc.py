"""dgl_policy_general_static (the entry point that allocates nothing: its workspace is a device array inside
the library) against dgl_policy_general with a caller-owned workspace, through the C ABI: two calls in a
row, the second with a smaller input, then a call beyond the static workspace's size."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _cvrp_case(B, N, knn, kd, device, seed):
    from dgl_amd.engine import CvrpEngine
    g = torch.Generator().manual_seed(seed)
    xy = torch.rand(B, N + 1, 2, generator=g).to(device)
    dem = torch.cat((torch.zeros(B, 1), torch.randint(1, 200, (B, N), generator=g) / 500.0), 1).to(device)
    eng = CvrpEngine(xy, dem, 1, knn, kd)
    eng.reset()
    start = torch.randint(1, N + 1, (B,), generator=g) + N + 1
    eng.step(start.to(device=device, dtype=torch.int32))
    return eng


def _run(eng, tr, teacher, static):
    """One training call (forward + backward); returns logits, probabilities, pick, teacher probability, gradient."""
    from dgl_amd._lib import check, ptr, stream_ptr
    from dgl_amd.engine import general_workspace
    eng.logits.zero_(); eng.prob.zero_(); eng.selected.zero_()
    tr.zero_grad()
    io = eng.policy_io()
    tp = torch.zeros(eng.R, dtype=torch.float32, device=eng.device)
    if static:
        check(eng.lib.dgl_policy_general_static(ctypes.byref(tr.w), ctypes.byref(tr.g), ctypes.byref(io), ptr(teacher),
                                                1.0 / eng.R, ptr(tp), stream_ptr()), "dgl_policy_general_static")
    else:
        ws = general_workspace(eng, io.R, io.n)
        check(eng.lib.dgl_policy_general(ctypes.byref(tr.w), ctypes.byref(tr.g), ctypes.byref(io), ptr(teacher),
                                         1.0 / eng.R, ptr(ws), ptr(tp), stream_ptr()), "dgl_policy_general")
    torch.cuda.synchronize()
    n = io.n
    return eng.logits[:, :n].clone(), eng.prob[:, :n].clone(), eng.selected.clone(), tp, tr.grads.clone()


def test_static_workspace_entry_point_matches_general(device):
    from dgl_amd import _lib
    from dgl_amd.train import PolicyTrainer
    from oracle.dgl_oracle import random_weights
    w = random_weights("CVRP", 21)
    tr = PolicyTrainer(w, _lib.PROBLEM_CVRP, device)
    # first call: 10 rows of 202 tokens (nearly the whole static workspace); second: 5 rows of 32 tokens
    for B, N, knn, kd in ((10, 240, 100, 100), (5, 90, 15, 15)):
        eng = _cvrp_case(B, N, knn, kd, device, seed=3 + B)
        assert eng.lib.dgl_policy_general_workspace_floats(B, knn + kd + 2) <= _lib.STATIC_WORKSPACE_FLOATS
        teacher = eng.idx_last[:, 1].clone()
        teacher[0] = eng.idx_dep[0, 2] + N + 1            # a move via the depot
        ref = _run(eng, tr, teacher, static=False)
        got = _run(eng, tr, teacher, static=True)
        for a, b, what in zip(got[:4], ref[:4], ("logits", "probabilities", "greedy pick", "teacher probability")):
            assert torch.equal(a, b), f"B={B}: {what} differ"          # the forward pass is deterministic: same bits
        assert float(ref[3].min()) > 1e-20 and float(ref[4].abs().max()) > 0
        # weight and bias gradients are ordered sums of partial sums: the same bits from every launch
        assert torch.equal(got[4], ref[4]), f"B={B}: gradients differ"


def test_static_workspace_rejects_oversize(device):
    from dgl_amd import _lib
    from dgl_amd.train import PolicyTrainer
    from oracle.dgl_oracle import random_weights
    tr = PolicyTrainer(random_weights("CVRP", 21), _lib.PROBLEM_CVRP, device)
    eng = _cvrp_case(64, 240, 100, 100, device, seed=1)
    assert eng.lib.dgl_policy_general_workspace_floats(64, 202) > _lib.STATIC_WORKSPACE_FLOATS
    io = eng.policy_io()
    rc = eng.lib.dgl_policy_general_static(ctypes.byref(tr.w), None, ctypes.byref(io), 0, 0.0, 0, _lib.stream_ptr())
    assert rc == -1 and b"static workspace" in eng.lib.dgl_last_error_string()
