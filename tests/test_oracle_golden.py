"""Pins the CPU oracle (oracle/dgl_oracle.py) to the vectors captured from the imported
reference (tests/golden/*.npz, written by oracle/make_goldens.py).

On the machine/torch build that generated the vectors the restatement is bit-equal to the
reference (same ATen/BLAS kernels on the same shapes); elsewhere float results may differ
in the last bits (different SIMD width / BLAS kernels), so indices are compared exactly
only where the recorded decision margin allows and floats with a tight tolerance.
"""
import glob
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_golden
from oracle.dgl_oracle import CVRPOracle, TSPOracle, cvrp_rollout, tsp_rollout

TSP_CASES = sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(GOLDEN, "tsp_*.npz")))
CVRP_CASES = sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(GOLDEN, "cvrp_*.npz")))


def _same_platform(case):
    return (str(case["meta/torch"]) == torch.__version__
            and str(case["meta/cpu"]) == torch.backends.cpu.get_cpu_capability())


def _cmp(a, b, exact, what, rtol=2e-5, atol=2e-6):
    if exact:
        assert np.array_equal(a, b, equal_nan=True), what
    else:
        np.testing.assert_allclose(a, b, rtol=rtol, atol=atol, err_msg=what)


@pytest.mark.parametrize("name", TSP_CASES)
def test_tsp_oracle_reproduces_reference(name):
    case, weights = load_golden(name)
    exact = _same_platform(case)
    if "idx" not in case and int(case["meta/beam"]) == 1:     # population case: tours only; a slice keeps the CPU suite short
        m = 32 if case["xy"].shape[1] < 1000 else 3
        case = dict(case, xy=case["xy"][:m], tour=case["tour"][:m], length=case["length"][:m])
    if case["xy"].shape[1] >= 1000 and case["xy"].shape[0] > 4:
        pytest.skip("large case: covered by the GPU tests (keeps the CPU suite short)")
    if case["xy"].shape[1] >= 10000 and int(case["meta/beam"]) > 1 and os.environ.get("DGL_LONG_TESTS", "0") != "1":
        pytest.skip("TSP-10000 beam 16: 107 s of oracle time (passes; DGL_LONG_TESTS=1 runs it); the GPU tests compare the "
                    "device with this capture directly")
    oracle = TSPOracle(weights, int(case["meta/knn"]))
    rec = []
    out = tsp_rollout(oracle, torch.from_numpy(case["xy"]), torch.from_numpy(case["start"]).long(),
                      beam=int(case["meta/beam"]), pomo=int(case["meta/pomo"]),
                      record=lambda s, o, r: rec.append({k: v.numpy().copy() for k, v in r.items() if k != "probs"}))
    if exact:
        assert np.array_equal(out["tour"].numpy(), case["tour"])
    _cmp(out["length"].numpy(), case["length"], exact, "tour length", rtol=1e-5)
    for s, r in enumerate(rec):
        k = r["idx"].shape[1]
        if exact and "idx" in case:
            assert np.array_equal(r["idx"], case["idx"][s][:, :k]), f"kNN set, step {s + 1}"
        if "logits" in case and exact:
            _cmp(r["logits"], case["logits"][s][:, :k + 1], exact, f"logits, step {s + 1}")
            _cmp(r["p"], case["p"][s][:, :k], exact, f"probabilities, step {s + 1}")
        if "tile" in case and exact:
            _cmp(r["tile"].reshape(r["tile"].shape[0], -1), case["tile"][s][:, :(k + 1) * 7], exact, "tile")


@pytest.mark.parametrize("name", CVRP_CASES)
def test_cvrp_oracle_reproduces_reference(name):
    case, weights = load_golden(name)
    exact = _same_platform(case)
    if "idx_last" not in case and int(case["meta/beam"]) == 1:
        m = 32 if case["node_xy"].shape[1] < 1000 else 2
        case = dict(case, depot_xy=case["depot_xy"][:m], node_xy=case["node_xy"][:m], demand=case["demand"][:m],
                    nodes=case["nodes"][:m], flags=case["flags"][:m], length=case["length"][:m])
    if case["node_xy"].shape[1] >= 1000 and case["node_xy"].shape[0] > 2:
        pytest.skip("large case: covered by the GPU tests (keeps the CPU suite short)")
    oracle = CVRPOracle(weights, int(case["meta/knn"]), int(case["meta/depot_knn"]))
    rec = []
    out = cvrp_rollout(oracle, torch.from_numpy(case["depot_xy"]), torch.from_numpy(case["node_xy"]),
                       torch.from_numpy(case["demand"]), torch.from_numpy(case["start"]).long(),
                       beam=int(case["meta/beam"]), pomo=int(case["meta/pomo"]),
                       record=lambda s, o, r: rec.append(
                           {k: v.numpy().copy() for k, v in r.items() if k != "probs"} | {"cap": o.capacity.numpy().copy()}))
    if exact:
        assert np.array_equal(out["nodes"].numpy(), case["nodes"])
        assert np.array_equal(out["flags"].numpy(), case["flags"])
    _cmp(out["length"].numpy(), case["length"], exact, "route length", rtol=1e-5)
    for s, r in enumerate(rec):
        kl, kd = r["idx_last"].shape[1], r["idx_dep"].shape[1]
        if exact and "idx_last" in case:
            assert np.array_equal(r["idx_last"], case["idx_last"][s][:, :kl])
            assert np.array_equal(r["idx_dep"], case["idx_dep"][s][:, :kd])
            if "logits" in case:
                _cmp(r["logits"], case["logits"][s][:, :kl + kd + 2], exact, f"logits, step {s + 1}")
                _cmp(r["cap"], case["capacity"][s], exact, "capacity")


def test_golden_feasibility_cvrp():
    """Sanity of the captured CVRP routes themselves: every customer once, loads <= capacity."""
    for name in CVRP_CASES:
        case, _ = load_golden(name)
        nodes, flags, dem = case["nodes"], case["flags"], case["demand"]
        rpi = nodes.shape[0] // dem.shape[0]
        for r in range(nodes.shape[0]):
            assert sorted(nodes[r].tolist()) == list(range(1, nodes.shape[1] + 1))
            load = 0.0
            for p in range(nodes.shape[1]):
                if flags[r, p]:
                    load = 0.0
                load += float(dem[r // rpi, nodes[r, p] - 1])
                assert load <= 1.0 + 1e-5


def test_cvrp_training_batch_oracle_reproduces_reference():
    """One batch of CVRPTrainer._train_one_batch captured from the reference (oracle/make_train_goldens.py):
    teacher probabilities, losses, gradient and parameter checksums of the first Adam steps."""
    from oracle.dgl_oracle import cvrp_train_batch, enable_grad
    g = np.load(os.path.join(GOLDEN, "train_cvrp_n40_b4.npz"))
    w0 = np.load(os.path.join(GOLDEN, "weights_cvrp_init789.npz"))
    w = {k: torch.from_numpy(w0[k]) for k in w0.files}
    orc = CVRPOracle(w, int(g["meta/knn"]), int(g["meta/depot_knn"]))
    orc.w = enable_grad(w)
    seen = {}

    def hook(t, o):
        if t <= 3:
            seen[t] = {k: (float(v.grad.double().norm()), float(v.grad.double().sum()))
                       for k, v in o.w.items() if v.grad is not None}

    probs, losses = cvrp_train_batch(orc, torch.from_numpy(g["aug_xy"]), torch.from_numpy(g["problems"][:, 1:, 2]),
                                     torch.from_numpy(g["solution"]), torch.from_numpy(g["solution_flag"]), hook=hook)
    exact = str(g["meta/torch"]) == torch.__version__
    if exact:
        assert np.array_equal(probs.numpy(), g["prob"])
    np.testing.assert_allclose(probs.numpy(), g["prob"], rtol=1e-4)
    np.testing.assert_allclose(np.array(losses), g["loss"], rtol=1e-5)
    for t in (1, 2, 3):
        for k, (nrm, sm) in seen[t].items():
            np.testing.assert_allclose(nrm, float(g[f"grad{t}/norm/{k}"]), rtol=1e-4, atol=1e-9)
    for k, v in orc.w.items():
        if f"final/norm/{k}" in g:
            np.testing.assert_allclose(float(v.detach().double().norm()), float(g[f"final/norm/{k}"]), rtol=1e-6)


def test_self_improvement_label_functions_against_reference():
    """Nearest-neighbour labels and the CVRP label augmentation of the oracle against the vectors the reference
    produced (greedy_search, data_augmentation with its recorded draws)."""
    import numpy as np
    import torch
    from oracle import dgl_oracle as O
    g = np.load(os.path.join(GOLDEN, "selfimp_cvrp_n30_b6.npz"))
    p = torch.from_numpy(g["problems0"])
    nodes, flags = O.cvrp_nearest_neighbour(p[:, :, :2].contiguous(), p[:, :, 2].contiguous())
    assert np.array_equal(nodes.numpy(), g["greedy/solution"]) and np.array_equal(flags.numpy(), g["greedy/flag"])
    assert np.array_equal(O.cvrp_route_length(p[:, :, :2].contiguous(), nodes, flags).numpy(), g["greedy/len"])
    for c in (0, 1):
        a, b = O.cvrp_label_augment(torch.from_numpy(g[f"chunk{c}/kept_solution"]), torch.from_numpy(g[f"chunk{c}/kept_flag"]),
                                    torch.from_numpy(g[f"chunk{c}/flip"]), torch.from_numpy(g[f"chunk{c}/shift"]))
        assert np.array_equal(a.numpy(), g[f"chunk{c}/aug_solution"]) and np.array_equal(b.numpy(), g[f"chunk{c}/aug_flag"])
    t = np.load(os.path.join(GOLDEN, "selfimp_tsp_n40_b6.npz"))
    sol, length = O.tsp_nearest_neighbour(torch.from_numpy(t["problems0"]))
    assert np.array_equal(sol.numpy(), t["greedy/solution"]) and np.array_equal(length.numpy(), t["greedy/len"])


def test_tsp_training_batch_of_the_oracle_against_reference():
    """TSPTrainer._train_one_batch restated in the oracle reproduces the reference's teacher probabilities."""
    import numpy as np
    import torch
    from oracle import dgl_oracle as O
    from dgl_amd.weights import load_checkpoint
    t = np.load(os.path.join(GOLDEN, "selfimp_tsp_n40_b6.npz"))
    w = load_checkpoint(os.path.join(GOLDEN, "ckpt_tsp_nn.pt"))
    orc = O.TSPOracle(w, knn=int(t["meta/knn"]))
    orc.w = O.enable_grad(w)
    probs = O.tsp_train_batch(orc, torch.from_numpy(t["train/aug_xy"]), torch.from_numpy(t["chunk1/aug_solution"]))
    assert np.array_equal(probs.numpy(), t["train/prob"])


def test_reference_tours_on_clustered_instances_hang_on_the_last_bit():
    """Why the >= 99 % identical-tours gate is asserted on uniform populations only (oracle/rounding_sensitivity.py):
    moving the reset's mean / std by one ulp leaves the uniform instances' tours alone and changes the mean length of
    the clustered ones by far less than the 0.01 % that is asserted there."""
    from oracle.rounding_sensitivity import tours_kept
    kept_u, rel_u = tours_kept("tsp_n1000_b128_nn_tours", 4)
    kept_c, rel_c = tours_kept("tsp_n1000_b128_clustered_nn_tours", 6)
    print(f"uniform: {kept_u}/4 tours kept; clustered: {kept_c}/6 kept, mean length moves by {rel_c:.1e}")
    assert kept_u == 4 and rel_u == 0.0
    assert rel_c < 1e-3              # (six instances: one changed tour moves their mean by a few 1e-5)



def test_reference_training_trajectory_hangs_on_the_last_bits():
    """The tolerance of the training-trajectory tests (tests/test_gpu_selfimprove.py) is set by how far the reference's
    own teacher probabilities move when its gradients change by what two summation orders differ by: Adam's first
    steps move every entry by +-lr, entries whose gradient is rounding noise take their sign from that noise."""
    from oracle.training_sensitivity import sensitivity
    (early_likely, early_all, all_likely, all_all), = sensitivity(seeds=1)
    print(f"reference under 3e-7 gradient noise: first 12 steps {early_likely:.1e} (likely tokens), {early_all:.1e} (all); "
          f"all steps {all_likely:.1e}, {all_all:.1e}")
    assert early_all > 3e-5 and all_all > 1e-2          # measured 1.2e-4 and 5e-2
    assert early_likely < 5e-4 and all_likely < 0.1     # and no more than the device tests allow
