"""GPU tests of the policy kernels' two arithmetic modes (include/dgl_amd.h, DGL_ARITH_*): the
f16 split-operand kernel must agree with the f32-input kernel far inside the 1e-3 logit bound,
and the f32-input kernel keeps its parity with the golden vectors."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


def _tsp_engine(R, N, knn, device, seed=0):
    from dgl_amd.engine import TspEngine
    g = torch.Generator().manual_seed(seed)
    eng = TspEngine(torch.rand(R, N, 2, generator=g).to(device), 1, knn)
    eng.reset()
    eng.step(torch.randint(0, N, (R,), generator=g, dtype=torch.int32).to(device))
    return eng


@pytest.mark.parametrize("R", [5, 600])          # one row per workgroup / two rows per workgroup (+ ragged tail)
@pytest.mark.parametrize("weights", ["random", "trained"])
def test_tsp_split_f16_agrees_with_f32(R, weights, device):
    from dgl_amd import _lib
    from dgl_amd.engine import PackedPolicy
    from dgl_amd.weights import load_checkpoint, random_state_dict
    import os
    from conftest import GOLDEN
    sd = random_state_dict("TSP", 3) if weights == "random" else load_checkpoint(os.path.join(GOLDEN, "ckpt_tsp_nn.pt"))
    eng = _tsp_engine(R, 300, 30, device)
    pol = PackedPolicy(sd, _lib.PROBLEM_TSP, device)
    out = {}
    for mode in (_lib.ARITH_F32, _lib.ARITH_SPLIT_F16):
        eng.arithmetic = mode
        eng.logits.zero_(); eng.prob.zero_(); eng.selected.zero_()
        eng.forward(pol)
        torch.cuda.synchronize()
        out[mode] = (eng.logits[:, :31].cpu().numpy().copy(), eng.prob[:, :31].cpu().numpy().copy(),
                     eng.selected.cpu().numpy().copy())
    a, b = out[_lib.ARITH_F32], out[_lib.ARITH_SPLIT_F16]
    scale = np.abs(a[0]).max(axis=1, keepdims=True)
    rel = float((np.abs(a[0] - b[0]) / scale).max())
    print(f"R={R} {weights}: max relative logit difference split-f16 vs f32 kernels {rel:.2e}")
    assert rel < 2e-5                               # both are f32-accurate evaluations of the same sums
    np.testing.assert_allclose(a[1], b[1], rtol=5e-4, atol=1e-7)   # relative probability error = absolute logit error
    assert (a[2] == b[2]).mean() >= 0.99


@pytest.mark.parametrize("R,knn", [(600, 30), (512, 30), (64, 50), (37, 100), (9, 127)])     # the 32-token kernel (two shapes); rows of 51 / 101 / 128 tokens (two and four tiles)
def test_forward_gives_the_same_bits_on_every_call(R, knn, device):
    """Nothing in the forward pass depends on timing: no float atomics, and no instruction reads a register another
    one is still writing (a build that placed hand-written assembly next to matrix instructions with results in VGPRs
    once failed exactly this: see dgl_amd/csrc/split_f16.h).  Twenty calls, one digest."""
    from dgl_amd import _lib
    from dgl_amd.engine import PackedPolicy
    from dgl_amd.weights import random_state_dict
    eng = _tsp_engine(R, 300, knn, device, seed=3)
    pol = PackedPolicy(random_state_dict("TSP", 5), _lib.PROBLEM_TSP, device)
    first = None
    for call in range(20):
        eng.logits.zero_(); eng.prob.zero_(); eng.selected.zero_()
        eng.forward(pol, want_logits=True, want_prob=True)
        got = (eng.logits.clone(), eng.prob.clone(), eng.selected.clone())
        if first is None:
            first = got
        else:
            assert all(torch.equal(a, b) for a, b in zip(first, got)), f"call {call} differs from the first"


def test_cvrp_split_f16_agrees_with_f32(device):
    from dgl_amd import _lib
    from dgl_amd.engine import CvrpEngine, PackedPolicy
    from dgl_amd.weights import random_state_dict
    g = torch.Generator().manual_seed(1)
    B, N = 520, 200
    xy = torch.rand(B, N + 1, 2, generator=g)                      # depot first
    demand = torch.cat([torch.zeros(B, 1), torch.randint(1, 10, (B, N), generator=g).float() / 40.0], 1)
    eng = CvrpEngine(xy.to(device), demand.to(device), 1, 15, 15)
    eng.reset()
    eng.step(torch.randint(1, N + 1, (B,), generator=g, dtype=torch.int32).to(device) + N + 1)
    pol = PackedPolicy(random_state_dict("CVRP", 5), _lib.PROBLEM_CVRP, device)
    out = {}
    for mode in (_lib.ARITH_F32, _lib.ARITH_SPLIT_F16):
        eng.arithmetic = mode
        eng.forward(pol)
        torch.cuda.synchronize()
        out[mode] = (eng.logits[:, :32].cpu().numpy().copy(), eng.selected.cpu().numpy().copy())
    a, b = out[_lib.ARITH_F32], out[_lib.ARITH_SPLIT_F16]
    rel = float((np.abs(a[0] - b[0]) / np.abs(a[0]).max(axis=1, keepdims=True)).max())
    print(f"CVRP: max relative logit difference split-f16 vs f32 kernels {rel:.2e}")
    assert rel < 2e-5
    assert (a[1] == b[1]).mean() >= 0.99


@pytest.mark.parametrize("name", ["tsp_n100_b8", "tsp_n100_b64_nn"])
def test_f32_kernel_logits_match_golden(name, device):
    """The f32-input kernel (selectable, not the default) against the reference's logits."""
    from dgl_amd import _lib
    from dgl_amd.engine import PackedPolicy, TspEngine
    case, weights = load_golden(name)
    eng = TspEngine(torch.from_numpy(case["xy"]).to(device), 1, int(case["meta/knn"]))
    eng.arithmetic = _lib.ARITH_F32
    pol = PackedPolicy(weights, _lib.PROBLEM_TSP, device)
    tour = torch.from_numpy(case["tour"]).to(device=device, dtype=torch.int32)
    eng.reset()
    worst = 0.0
    for t in range(1, eng.N):
        eng.step(tour[:, t - 1].contiguous())
        k = eng.k
        ok = (eng.idx.cpu().numpy()[:, :k] == case["idx"][t - 1][:, :k]).all(axis=1)
        eng.forward(pol)
        lg, g = eng.logits.cpu().numpy()[:, :k + 1], case["logits"][t - 1][:, :k + 1]
        if ok.any():
            worst = max(worst, float((np.abs(lg[ok] - g[ok]) / np.abs(g[ok]).max(axis=1, keepdims=True)).max()))
    print(f"{name}: f32 kernel max relative logit error {worst:.2e}")
    assert worst < 1e-3


def test_range_guard_reruns_in_binary32(device):
    """Weights scaled so that activations leave the binary16 range: the split-f16 kernel reports it
    (dgl_policy_io_t.overflow_out) and the rollout is repeated with the binary32 kernel, so the result is the
    binary32 kernel's; with ordinary weights nothing is reported."""
    from dgl_amd import _lib
    from dgl_amd.engine import PackedPolicy, TspEngine
    from oracle.dgl_oracle import random_weights
    g = torch.Generator().manual_seed(3)
    xy = torch.rand(8, 60, 2, generator=g).to(device)
    start = torch.zeros(8, dtype=torch.int32, device=device)
    w = random_weights("TSP", 5)
    big = dict(w)
    big["encoder.embedding.weight"] = w["encoder.embedding.weight"] * 3e5      # tokens are O(1): activations ~1e5
    eng = TspEngine(xy, 1, 30)
    assert eng.arithmetic == _lib.ARITH_SPLIT_F16
    eng.greedy_rollout(PackedPolicy(w, _lib.PROBLEM_TSP, device), start)
    assert eng.reran_f32 is False and int(eng.overflow.item()) == 0
    eng.greedy_rollout(PackedPolicy(big, _lib.PROBLEM_TSP, device), start)
    assert eng.reran_f32 is True
    tour = eng.state["tour"].clone()
    ref = TspEngine(xy, 1, 30)
    ref.arithmetic = _lib.ARITH_F32
    ref.greedy_rollout(PackedPolicy(big, _lib.PROBLEM_TSP, device), start)
    assert ref.reran_f32 is False and torch.equal(ref.state["tour"], tour)
    # the reference-protocol path checks per evaluation
    from dgl_amd.engine import forward_range_guarded
    eng.reset(); eng.step(start)
    forward_range_guarded(eng, PackedPolicy(big, _lib.PROBLEM_TSP, device))
    sel = eng.selected.clone()
    ref.reset(); ref.step(start); ref.forward(PackedPolicy(big, _lib.PROBLEM_TSP, device))
    assert torch.equal(sel, ref.selected)


def test_range_guard_sees_weights_beyond_binary16(device):
    """A matrix entry of 1e5 saturates its binary16 image when dgl_policy_pack splits it, whatever the activations are:
    the pack leaves a mark in the packed buffer and the split-f16 kernel reports it like an activation overflow."""
    from dgl_amd import _lib
    from dgl_amd.engine import PackedPolicy, TspEngine, forward_range_guarded
    from oracle.dgl_oracle import random_weights
    g = torch.Generator().manual_seed(4)
    xy = torch.rand(6, 60, 2, generator=g).to(device)
    start = torch.zeros(6, dtype=torch.int32, device=device)
    w = dict(random_weights("TSP", 5))
    m = w["decoder.layers.1.feedForward.W2.weight"].clone()
    m[3, 17] = 1e5
    m[9, 400] = 0.0
    w["decoder.layers.1.feedForward.W2.weight"] = m
    w["decoder.layers.1.feedForward.W1.bias"] = torch.full_like(w["decoder.layers.1.feedForward.W1.bias"], -1e3)  # hidden = 0
    pol = PackedPolicy(w, _lib.PROBLEM_TSP, device)
    eng = TspEngine(xy, 1, 30)
    eng.reset(); eng.step(start)
    eng.overflow.zero_()
    eng.forward(pol)
    assert int(eng.overflow.item()) == 1            # no activation is large (that block's hidden layer is all zeros)
    forward_range_guarded(eng, pol)
    sel = eng.selected.clone()
    ref = TspEngine(xy, 1, 30)
    ref.arithmetic = _lib.ARITH_F32
    ref.reset(); ref.step(start); ref.forward(pol)
    assert torch.equal(sel, ref.selected)


def _wide_vs_general(eng, pol, n_cols):
    """One evaluation through dgl_policy_forward's wide kernel and one through the general path's launches."""
    import ctypes
    from dgl_amd._lib import check, ptr, stream_ptr
    from dgl_amd.engine import general_workspace
    outs = []
    for wide in (True, False):
        eng.logits.zero_(); eng.prob.zero_(); eng.selected.zero_()
        io = eng.policy_io()
        if wide:
            check(eng.lib.dgl_policy_forward(ptr(pol.packed), ctypes.byref(io), stream_ptr()), "dgl_policy_forward")
        else:
            ws = general_workspace(eng, io.R, io.n)
            check(eng.lib.dgl_policy_general(ctypes.byref(pol.w), None, ctypes.byref(io), 0, 0.0, ptr(ws), 0, stream_ptr()),
                  "dgl_policy_general")
        torch.cuda.synchronize()
        outs.append((eng.logits[:, :n_cols].cpu().numpy().copy(), eng.prob[:, :n_cols].cpu().numpy().copy(),
                     eng.selected.cpu().numpy().copy()))
    return outs


@pytest.mark.parametrize("knn", [32, 50, 63, 64, 70, 95, 96, 100, 127])      # 33 .. 128 tokens: 2, 3 and 4 tiles, full and ragged
@pytest.mark.parametrize("weights", ["random", "trained"])
def test_tsp_wide_kernel_agrees_with_general_path(knn, weights, device):
    """Rows of 33..128 tokens in one launch (one workgroup per row, policy16_wide.h) against the general path's
    sequence of kernels: two f32-accurate evaluations of the same network (TSP/TSPModel.py:238-457)."""
    import os
    from conftest import GOLDEN
    from dgl_amd import _lib
    from dgl_amd.engine import PackedPolicy
    from dgl_amd.weights import load_checkpoint, random_state_dict
    sd = random_state_dict("TSP", 3) if weights == "random" else load_checkpoint(os.path.join(GOLDEN, "ckpt_tsp_nn.pt"))
    eng = _tsp_engine(37, 300, knn, device, seed=knn)
    pol = PackedPolicy(sd, _lib.PROBLEM_TSP, device)
    assert eng.overflow.item() == 0
    (la, pa, sa), (lb, pb, sb) = _wide_vs_general(eng, pol, knn + 1)
    scale = np.abs(lb).max(axis=1, keepdims=True)
    rel = float((np.abs(la - lb) / scale).max())
    print(f"knn={knn} {weights}: max relative logit difference wide kernel vs general path {rel:.2e}")
    assert rel < 3e-5
    # a probability moves by (its value) x (twice the largest logit difference): logits reach several hundred here
    np.testing.assert_allclose(pa, pb, rtol=5e-4 + 2 * float(np.abs(la - lb).max()), atol=1e-7)
    assert (sa == sb).mean() >= 0.97 and eng.overflow.item() == 0


@pytest.mark.parametrize("knn,kd", [(16, 16), (30, 30), (50, 50), (40, 63), (63, 63)])
def test_cvrp_wide_kernel_agrees_with_general_path(knn, kd, device):
    """The CVRP token layout (direct candidates, current node, via-depot candidates, depot; two anchors re-embedded at
    the decoder's entry, infeasible direct moves masked: CVRP/CVRPModel.py:407-523) through the wide kernel."""
    import os
    from conftest import GOLDEN
    from dgl_amd import _lib
    from dgl_amd.engine import CvrpEngine, PackedPolicy
    from dgl_amd.weights import load_checkpoint
    g = torch.Generator().manual_seed(knn + kd)
    R, N = 29, 200
    xy = torch.rand(R, N + 1, 2, generator=g)
    dem = torch.cat((torch.zeros(R, 1), torch.randint(1, 200, (R, N), generator=g) / 500.0), 1)
    eng = CvrpEngine(xy.to(device), dem.to(device), 1, knn, kd)
    eng.reset()
    eng.step((torch.randint(1, N + 1, (R,), generator=g, dtype=torch.int32) + N + 1).to(device))
    eng.step(torch.randint(1, N + 1, (R,), generator=g, dtype=torch.int32).to(device))
    pol = PackedPolicy(load_checkpoint(os.path.join(GOLDEN, "ckpt_cvrp_nn.pt")), _lib.PROBLEM_CVRP, device)
    n = eng.policy_io().n
    assert n == knn + kd + 2
    (la, pa, sa), (lb, pb, sb) = _wide_vs_general(eng, pol, n)
    scale = np.abs(lb).max(axis=1, keepdims=True)
    rel = float((np.abs(la - lb) / scale).max())
    print(f"knn={knn}/{kd}: max relative logit difference wide kernel vs general path {rel:.2e}")
    assert rel < 3e-5
    # a probability moves by (its value) x (twice the largest logit difference): logits reach several hundred here
    np.testing.assert_allclose(pa, pb, rtol=5e-4 + 2 * float(np.abs(la - lb).max()), atol=1e-7)
    assert (sa == sb).mean() >= 0.97


def test_wide_kernel_range_guard_falls_back_to_the_general_path(device, monkeypatch):
    """Rows of more than 32 tokens with activations beyond the binary16 range: the one-launch kernel reports them
    (overflow_out) and the rollout is repeated through the general path, whose scaled splits take any magnitude - the
    result is what the general path alone gives."""
    from dgl_amd import _lib
    from dgl_amd.engine import PackedPolicy, TspEngine
    from oracle.dgl_oracle import random_weights
    g = torch.Generator().manual_seed(6)
    xy = torch.rand(6, 120, 2, generator=g).to(device)
    start = torch.zeros(6, dtype=torch.int32, device=device)
    w = random_weights("TSP", 5)
    big = dict(w)
    big["encoder.embedding.weight"] = w["encoder.embedding.weight"] * 3e5
    eng = TspEngine(xy, 1, 50)
    eng.greedy_rollout(PackedPolicy(w, _lib.PROBLEM_TSP, device), start)
    assert eng.reran_f32 is False and int(eng.overflow.item()) == 0
    eng.greedy_rollout(PackedPolicy(big, _lib.PROBLEM_TSP, device), start)
    assert eng.reran_f32 is True
    tour = eng.state["tour"].clone()
    monkeypatch.setenv("DGL_WIDE_FORWARD", "0")
    ref = TspEngine(xy, 1, 50)
    ref.greedy_rollout(PackedPolicy(big, _lib.PROBLEM_TSP, device), start)
    assert torch.equal(ref.state["tour"], tour)
