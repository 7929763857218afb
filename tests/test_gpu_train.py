"""GPU parity of the general-n forward and the training step (dgl_policy_general, dgl_adam_step)
against the oracle's autograd (the reference's loss / backward / Adam are torch's own)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _tsp_pair(B, N, knn, device, seed):
    from dgl_amd.engine import TspEngine
    from oracle.dgl_oracle import TSPOracle, enable_grad, random_weights
    g = torch.Generator().manual_seed(seed)
    xy = torch.rand(B, N, 2, generator=g)
    w = random_weights("TSP", seed)
    orc = TSPOracle(w, knn)
    orc.w = enable_grad(w)
    orc.reset(xy)
    start = torch.randint(0, N, (B,), generator=g)
    orc.step(start)
    eng = TspEngine(xy.to(device), 1, knn)
    eng.reset()
    eng.step(start.to(device=device, dtype=torch.int32))
    return w, orc, eng


def _cvrp_pair(B, N, knn, kd, device, seed):
    from dgl_amd.engine import CvrpEngine
    from oracle.dgl_oracle import CVRPOracle, enable_grad, random_weights
    g = torch.Generator().manual_seed(seed)
    depot, nodes = torch.rand(B, 2, generator=g), torch.rand(B, N, 2, generator=g)
    demand = torch.randint(1, 200, (B, N), generator=g) / 500.0
    w = random_weights("CVRP", seed)
    orc = CVRPOracle(w, knn, kd)
    orc.w = enable_grad(w)
    orc.reset(depot, nodes, demand)
    start = torch.randint(1, N + 1, (B,), generator=g) + N + 1
    orc.step(start.clone())
    xy = torch.cat((depot[:, None, :], nodes), 1).to(device)
    dem = torch.cat((torch.zeros(B, 1), demand), 1).to(device)
    eng = CvrpEngine(xy, dem, 1, knn, kd)
    eng.reset()
    eng.step(start.to(device=device, dtype=torch.int32))
    return w, orc, eng


def _compare_grads(trainer, orc, rtol):
    worst = 0.0
    # some gradients vanish identically (softmax gradients sum to zero, so does the last bias's): errors are
    # measured against the tensor's own scale, but not below 1e-3 of the largest gradient entry of the model
    floor = 1e-3 * max(float(v.grad.abs().max()) for v in orc.w.values() if v.grad is not None)
    for k, g in trainer.gradient_dict().items():
        ref = orc.w[k].grad
        ref = torch.zeros_like(orc.w[k]) if ref is None else ref
        scale = max(float(ref.abs().max()), floor)
        diff = g.cpu() - ref
        err = float(diff.abs().max()) / scale
        # a hidden unit whose pre-activation is within rounding of 0 may fall on the other side of the ReLU: the
        # bulk must agree tightly (Frobenius), single entries loosely
        fro = float(diff.norm()) / max(float(ref.norm()), floor)
        worst = max(worst, fro)
        assert fro < rtol, f"{k}: relative Frobenius gradient error {fro:.2e}"
        assert err < 20 * rtol, f"{k}: relative gradient error {err:.2e}"
    return worst


@pytest.mark.parametrize("N,knn", [(60, 30), (120, 50), (230, 100)])
def test_tsp_train_step_matches_autograd(N, knn, device):
    from dgl_amd._lib import PROBLEM_TSP
    from dgl_amd.train import PolicyTrainer
    from oracle.dgl_oracle import train_forward_backward
    B = 6
    w, orc, eng = _tsp_pair(B, N, knn, device, seed=11 + N)
    tr = PolicyTrainer(w, PROBLEM_TSP, device)
    idx = orc.knn_indices()
    teacher = idx[torch.arange(B), torch.arange(B) % idx.shape[1]]
    teacher[-1] = orc.tour[-1, 0]                    # an already visited node: probability 1e-20, no gradient
    prob_ref, loss_ref = train_forward_backward(orc, teacher)
    tr.zero_grad()
    prob = tr.forward_backward(eng, teacher)
    np.testing.assert_allclose(prob.cpu().numpy(), prob_ref.numpy(), rtol=2e-4)
    worst = _compare_grads(tr, orc, 2e-3)
    print(f"TSP N={N} n={knn + 1}: loss {loss_ref:.6f}, worst relative gradient error {worst:.2e}")


@pytest.mark.parametrize("N,knn,kd", [(80, 15, 15), (120, 25, 25), (260, 100, 100), (300, 127, 127)])      # 32, 52, 202, 256 tokens (the maximum)
def test_cvrp_train_step_matches_autograd(N, knn, kd, device):
    from dgl_amd._lib import PROBLEM_CVRP
    from dgl_amd.train import PolicyTrainer
    from oracle.dgl_oracle import train_forward_backward
    B = 5
    w, orc, eng = _cvrp_pair(B, N, knn, kd, device, seed=5 + N)
    tr = PolicyTrainer(w, PROBLEM_CVRP, device)
    il, idp = orc.knn_indices()
    teacher = il[torch.arange(B), torch.arange(B) % il.shape[1]].clone()
    teacher[1] = idp[1, 2] + N + 1                   # a via-depot move
    prob_ref, loss_ref = train_forward_backward(orc, teacher)
    tr.zero_grad()
    prob = tr.forward_backward(eng, teacher)
    np.testing.assert_allclose(prob.cpu().numpy(), prob_ref.numpy(), rtol=2e-4)
    worst = _compare_grads(tr, orc, 2e-3)
    print(f"CVRP N={N} n={knn + kd + 2}: loss {loss_ref:.6f}, worst relative gradient error {worst:.2e}")
    # sums over tokens are ordered sums of partial sums (no float atomics): a second evaluation gives the same bits
    first = tr.grads.clone()
    tr.zero_grad()
    prob2 = tr.forward_backward(eng, teacher)
    assert torch.equal(prob2, prob) and torch.equal(tr.grads, first), "gradients are not reproducible"


@pytest.mark.parametrize("N,knn,kd", [(30, 1, 1), (60, 15, 16), (120, 25, 25), (260, 100, 100), (300, 127, 127)])   # 4 ... 256 tokens
def test_fused_chains_match_separate_launches(N, knn, kd, device, monkeypatch):
    """Everything after a block's attention runs as one launch per direction; DGL_FF_FUSED=0 and DGL_GRAD_PAIRED=1 keep
    the earlier sequences of launches.  Same products and sums up to the scaling of intermediate operands (by the
    largest value of a workgroup's 32 tokens instead of the tensor's) and the order of the hidden layer's two halves:
    probabilities, logits and gradients agree to rounding."""
    from dgl_amd._lib import PROBLEM_CVRP
    from dgl_amd.engine import PackedPolicy
    from dgl_amd.train import PolicyTrainer
    B = 5
    w, orc, eng = _cvrp_pair(B, N, knn, kd, device, seed=5 + N)
    il, _ = orc.knn_indices()
    teacher = il[torch.arange(B), torch.arange(B) % il.shape[1]].clone()
    out = {}
    for name, env in (("fused", {}), ("separate", {"DGL_FF_FUSED": "0"}), ("paired", {"DGL_GRAD_PAIRED": "1"})):
        with monkeypatch.context() as m:
            for k, v in env.items():
                m.setenv(k, v)
            tr = PolicyTrainer(w, PROBLEM_CVRP, device)
            tr.zero_grad()
            prob = tr.forward_backward(eng, teacher)
            eng.forward(PackedPolicy(w, PROBLEM_CVRP, device), want_logits=True, want_prob=True)     # the inference forward
            out[name] = (prob.clone(), tr.grads.double().clone(), eng.logits.clone())
    n = knn + kd + 2
    for other in ("separate", "paired"):
        np.testing.assert_allclose(out["fused"][0].cpu().numpy(), out[other][0].cpu().numpy(), rtol=5e-6)
        ga, gb = out["fused"][1], out[other][1]
        assert float((ga - gb).norm() / gb.norm()) < 2e-6
        la, lb = out["fused"][2][:, :n], out[other][2][:, :n]
        assert float((la - lb).abs().max() / lb.abs().max()) < 2e-6


def test_adam_matches_torch(device):
    from dgl_amd._lib import PROBLEM_TSP
    from dgl_amd.train import PolicyTrainer
    from oracle.dgl_oracle import random_weights
    w = random_weights("TSP", 3)
    tr = PolicyTrainer(w, PROBLEM_TSP, device)
    params = [torch.nn.Parameter(w[k].clone()) for k in tr.keys]
    opt = torch.optim.Adam(params, lr=1e-4, weight_decay=1e-6)          # CVRP/train.py:49-54
    g = torch.Generator().manual_seed(0)
    for step in range(3):
        flat = torch.randn(tr.numel, generator=g) * 1e-2
        tr.grads.copy_(flat.to(device))
        off = 0
        for p in params:
            p.grad = flat[off:off + p.numel()].view_as(p).clone()
            off += p.numel()
        opt.step()
        tr.optimizer_step()
    got = tr.params.cpu()
    ref = torch.cat([p.detach().reshape(-1) for p in params])
    assert float((got - ref).abs().max()) < 2e-7
    sd = tr.state_dict()
    assert list(sd.keys()) == list(w.keys())
    assert torch.equal(sd["decoder.k_1.weight"].cpu(), w["decoder.k_1.weight"])


def test_general_forward_matches_fused(device):
    """n <= 32: the general f32 path and the fused kernels must pick the same actions."""
    import ctypes
    from dgl_amd import _lib
    from dgl_amd.engine import PackedPolicy, general_workspace
    from dgl_amd._lib import check, ptr, stream_ptr
    w, orc, eng = _tsp_pair(40, 100, 30, device, seed=2)
    pol = PackedPolicy(w, _lib.PROBLEM_TSP, device)
    eng.forward(pol)
    torch.cuda.synchronize()
    lg, pr, sel = eng.logits.clone(), eng.prob.clone(), eng.selected.clone()
    eng.logits.zero_(); eng.prob.zero_(); eng.selected.zero_()
    io = eng.policy_io()
    ws = general_workspace(eng, io.R, io.n)
    check(eng.lib.dgl_policy_general(ctypes.byref(pol.w), None, ctypes.byref(io), 0, 0.0, ptr(ws), 0, stream_ptr()),
          "dgl_policy_general")
    torch.cuda.synchronize()
    scale = lg[:, :31].abs().max()
    assert float((eng.logits[:, :31] - lg[:, :31]).abs().max() / scale) < 2e-5
    np.testing.assert_allclose(eng.prob[:, :31].cpu().numpy(), pr[:, :31].cpu().numpy(), rtol=3e-4, atol=1e-7)
    assert torch.equal(eng.selected, sel)


def test_greedy_rollout_with_50_neighbours(device):
    """knn = 50 (51 tokens per row): the engine switches to the general path; tours follow the oracle."""
    from dgl_amd import _lib
    from dgl_amd.engine import PackedPolicy, TspEngine
    from oracle.dgl_oracle import TSPOracle, random_weights, tsp_rollout
    g = torch.Generator().manual_seed(9)
    B, N, knn = 4, 90, 50
    xy = torch.rand(B, N, 2, generator=g)
    w = random_weights("TSP", 4)
    start = torch.zeros(B, dtype=torch.long)
    ref = tsp_rollout(TSPOracle(w, knn), xy, start[:1])               # one start node, tiled over the batch
    eng = TspEngine(xy.to(device), 1, knn)
    pol = PackedPolicy(w, _lib.PROBLEM_TSP, device)
    length = eng.greedy_rollout(pol, start.to(device=device, dtype=torch.int32))
    tour = eng.state["tour"].cpu().numpy()
    same = (tour == ref["tour"].numpy()).all(axis=1).mean()
    print(f"knn=50 greedy tours identical to the oracle: {same:.2f}")
    # random-init weights have arg-max margins at rounding level, so only the first decisions are compared
    assert (tour[:, :3] == ref["tour"].numpy()[:, :3]).all()
    assert np.isfinite(length.cpu().numpy()).all()


def test_cvrp_training_batch_matches_reference_golden(device):
    """CVRPTrainer._train_one_batch on the device against the batch captured from the reference
    (tests/golden/train_cvrp_n40_b4.npz): the training reset's (x, y, demand) distances, teacher-forced
    decode, one Adam update per decode step."""
    import os
    from conftest import GOLDEN
    from dgl_amd._lib import PROBLEM_CVRP
    from dgl_amd.engine import CvrpEngine
    from dgl_amd.train import PolicyTrainer
    g = np.load(os.path.join(GOLDEN, "train_cvrp_n40_b4.npz"))
    w0 = np.load(os.path.join(GOLDEN, "weights_cvrp_init789.npz"))
    w = {k: torch.from_numpy(w0[k]) for k in w0.files}
    knn, kd = int(g["meta/knn"]), int(g["meta/depot_knn"])
    xy = torch.from_numpy(g["aug_xy"]).to(device)
    dem = torch.from_numpy(g["problems"][:, :, 2]).to(device)
    eng = CvrpEngine(xy, dem, 1, knn, kd, demand_in_distance=True)
    tr = PolicyTrainer(w, PROBLEM_CVRP, device)
    N = eng.N
    sol, flag = torch.from_numpy(g["solution"]), torch.from_numpy(g["solution_flag"])
    act = (sol + flag * (N + 1)).to(device=device, dtype=torch.int32)
    eng.reset()
    eng.step(act[:, 0].contiguous())
    probs = []
    for t in range(1, N):
        tr.zero_grad()
        prob = tr.forward_backward(eng, act[:, t].contiguous())
        if t <= 3:
            # (the last block's W2 bias gradient vanishes identically: tolerances are floored by the model's scale)
            top = max(float(g[f"grad{t}/norm/{k}"]) for k in tr.keys)
            for k, gv in tr.gradient_dict().items():
                ref = float(g[f"grad{t}/norm/{k}"])
                got = float(gv.double().norm())
                assert abs(got - ref) <= 2e-3 * max(ref, 1e-3 * top), f"step {t} {k}: gradient norm {got} vs {ref}"
                if f"grad{t}/full/{k}" in g:
                    full = g[f"grad{t}/full/{k}"]
                    np.testing.assert_allclose(gv.cpu().numpy(), full, rtol=5e-3, atol=2e-3 * np.abs(full).max() + 1e-6 * top)
        tr.optimizer_step()
        if t <= 3:
            sd = tr.state_dict()
            for k in tr.keys:
                if f"param{t}/full/{k}" in g:
                    # Adam's first updates are lr * g / (|g| + eps): where a gradient entry is rounding noise of the
                    # size of eps the update is too, so a few entries may differ by a fraction of t * lr.  (Sums over tokens
                    # of softmax gradients cancel heavily; the products carry 22 significant bits, the reference's 24.)
                    d = np.abs(sd[k].cpu().numpy() - g[f"param{t}/full/{k}"])
                    noise_only = float(g[f"grad{t}/norm/{k}"]) < 1e-3 * top       # identically vanishing gradient
                    assert d.max() < t * 1e-4 and (noise_only or np.quantile(d, 0.98) < t * 6e-6), (k, t, d.max())
        probs.append(prob.cpu().numpy())
        eng.step(act[:, t].contiguous())
    probs = np.stack(probs, 0)
    rel = np.abs(probs - g["prob"]) / g["prob"]
    print(f"teacher probabilities over {N - 1} Adam steps: max relative difference {rel.max():.2e}, "
          f"median {np.median(rel):.2e}")
    loss = -np.log(probs.astype(np.float64)).mean(axis=1)
    np.testing.assert_allclose(loss, g["loss"], rtol=2e-3, atol=2e-4)
    assert np.median(rel) < 1e-4 and rel.max() < 2e-2
    sd = tr.state_dict()
    # the gradients of the two last biases vanish identically (softmax gradients sum to zero): Adam turns their
    # rounding noise into steps of up to lr each, so those two tensors random-walk
    noise_driven = ("decoder.Linear_final.bias", "decoder.layers.2.feedForward.W2.bias")
    for k in tr.keys:
        ref = float(g[f"final/norm/{k}"])
        slack = (N * 1e-4 * np.sqrt(sd[k].numel())) if k in noise_driven else 0.0
        assert abs(float(sd[k].double().norm()) - ref) <= 1e-3 * ref + slack, k
