"""The reference-facing classes (same names / signatures as the reference) driven the way the
reference's testers drive them, compared with the golden tours."""
import os

import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu

MODEL_PARAMS = dict(mode="test", embedding_dim=128, sqrt_embedding_dim=128 ** 0.5, decoder_layer_num=3, qkv_dim=16,
                    head_num=8, ff_hidden_dim=512)
TSP_INFO = [True, True, True, False, True, True, False, False, False, False, False]
CVRP_INFO = TSP_INFO + [False, False]


def _tsp_tester(case, weights, beam):
    from dgl_amd import TSPTester
    env_params = dict(data_path="", baseline_path="", mode="test", test_mode="pomo_test", aug_size=int(case["meta/pomo"]),
                      append_information=TSP_INFO, aggregation_nums=50, pomo_size=1)
    tester_params = dict(use_cuda=True, cuda_device_num=0, test_mode="pomo_test", knn=int(case["meta/knn"]),
                         beam_size=beam, model_load={"enable": False})
    t = TSPTester(env_params, dict(MODEL_PARAMS, append_information=TSP_INFO), None, tester_params, None)
    t.model.load_state_dict(weights)
    t.env.raw_data_nodes = torch.from_numpy(case["xy"])
    B, N = case["xy"].shape[:2]
    t.env.load_problems(0, B, N, beam)
    t.env.start_idx, t.env.step_size = 0, B
    return t


def _cvrp_tester(case, weights, beam):
    from dgl_amd import CVRPTester
    env_params = dict(mode="test", test_mode="pomo_test", aug_size=int(case["meta/pomo"]), append_information=CVRP_INFO,
                      aggregation_nums=10, pomo_size=1)
    tester_params = dict(use_cuda=True, cuda_device_num=0, test_mode="pomo_test", knn=int(case["meta/knn"]),
                         depot_knn=int(case["meta/depot_knn"]), beam_size=beam, model_load={"enable": False})
    t = CVRPTester(env_params, dict(MODEL_PARAMS, append_information=CVRP_INFO), None, tester_params, None)
    t.model.load_state_dict(weights)
    t.env.raw_depot_node_xy = torch.from_numpy(case["depot_xy"])
    t.env.raw_node_xy = torch.from_numpy(case["node_xy"])
    t.env.raw_node_demand = torch.from_numpy(case["demand"])
    B, N = case["node_xy"].shape[:2]
    t.env.beam_size, t.env.problem_size = beam, N
    t.env.load_problems(0, B, N)
    t.env.start_idx, t.env.step_size = 0, B
    return t


@pytest.mark.parametrize("name", ["tsp_n100_b64_nn", "tsp_n200_b2_beam8_nn", "tsp_n60_b2_beam4_pomo2", "tsp_n200_b4_k50_nn"])
def test_tsp_reference_protocol_and_fast_path(name, device):
    case, weights = load_golden(name)
    beam = int(case["meta/beam"])
    t = _tsp_tester(case, weights, beam)
    B, N = case["xy"].shape[:2]
    start = torch.from_numpy(case["start"]).long()
    _, _, _, score = t.beamsearch_tour_nodes_shortest(beam, B, N, start=start)
    tours = t.last_tours.cpu().numpy()
    assert tours.dtype == np.int64 and tours.shape == case["tour"].shape
    same_ref = (tours == case["tour"]).all(axis=1).mean()
    best = case["length"].reshape(B, -1).min(axis=1).mean()
    t2 = _tsp_tester(case, weights, beam)
    shortest = t2.fast_rollout(beam, start=start)
    same_fast = (t2.last_tours.cpu().numpy() == case["tour"]).all(axis=1).mean()
    print(f"{name}: identical tours reference-protocol {same_ref:.3f}, fast path {same_fast:.3f}; score {score:.5f} "
          f"fast {shortest.mean().item():.5f} reference {best:.5f}")
    if name.endswith("_nn"):
        assert same_ref >= 0.95 and same_fast >= 0.95
    assert abs(score - best) < 2e-3 * best
    assert abs(shortest.mean().item() - best) < 2e-3 * best


@pytest.mark.parametrize("name", ["cvrp_n100_b64_nn", "cvrp_n150_b2_beam8_nn", "cvrp_n200_b4_k50_nn"])
def test_cvrp_reference_protocol_and_fast_path(name, device):
    case, weights = load_golden(name)
    beam = int(case["meta/beam"])
    t = _cvrp_tester(case, weights, beam)
    B, N = case["node_xy"].shape[:2]
    start = torch.from_numpy(case["start"]).long()
    score, _, _ = t.beamsearch_tour_nodes_shortest(beam, B, N + 1, start=start)
    same_ref = ((t.last_tours.cpu().numpy() == case["nodes"]).all(axis=1)
                & (t.last_flags.cpu().numpy() == case["flags"]).all(axis=1)).mean()
    best = case["length"].reshape(B, -1).min(axis=1).mean()
    t2 = _cvrp_tester(case, weights, beam)
    shortest = t2.fast_rollout(beam, start=start)
    same_fast = (t2.last_tours.cpu().numpy() == case["nodes"]).all(axis=1).mean()
    print(f"{name}: identical routes reference-protocol {same_ref:.3f}, fast path {same_fast:.3f}; score {score:.5f} "
          f"fast {shortest.mean().item():.5f} reference {best:.5f}")
    assert same_ref >= 0.9 and same_fast >= 0.9
    assert abs(score - best) < 2e-3 * best


def test_step_state_views_match_golden(device):
    """step_state.data / avg / std / capacity / ninf_mask / selected_flag as the reference exposes them."""
    case, weights = load_golden("cvrp_n60_b4")
    t = _cvrp_tester(case, weights, 1)
    env = t.env
    env.reset("test")
    N = env.problem_size
    acts = case["nodes"].astype(np.int64) + (N + 1) * case["flags"].astype(np.int64)
    for step in range(1, 4):
        sel = torch.from_numpy(acts[:, step - 1].copy()).to(device)
        state, reward, done = env.step(sel)
        assert np.array_equal(sel.cpu().numpy(), case["nodes"][:, step - 1])       # rewritten to node ids
        g = case[f"snap{step}/data"]
        np.testing.assert_allclose(state.data.cpu().numpy(), g, rtol=2e-5, atol=2e-6)
        assert np.array_equal(state.ninf_mask.cpu().numpy(), case[f"snap{step}/ninf"])
        assert np.array_equal(state.selected_flag.cpu().numpy(), case["flags"][:, :step])
        np.testing.assert_array_equal(state.capacity.cpu().numpy(), case[f"snap{step}/capacity"])
        assert reward == (None, None) and not done


def test_tester_run_with_files(device, tmp_path):
    """TSPTester.run end to end: instance + baseline files in the reference's text formats."""
    from dgl_amd import TSPTester
    case, weights = load_golden("tsp_n100_b64_nn")
    xy = case["xy"][:16]
    data = tmp_path / "tsp100.txt"
    data.write_text(" ".join(f"{x:.9g},{y:.9g}" for x, y in xy.reshape(-1, 2)))
    base = tmp_path / "base.txt"
    base.write_text("\n".join(f"{i} {l:.9g}" for i, l in enumerate(case["length"][:16])))
    ck = os.path.join(os.path.dirname(__file__), "golden")
    env_params = dict(data_path=str(data), baseline_path=str(base), mode="test", test_mode="pomo_test", aug_size=1,
                      append_information=TSP_INFO, aggregation_nums=50, pomo_size=1)
    tester_params = dict(use_cuda=True, cuda_device_num=0, test_mode="pomo_test", knn=30, beam_size=1,
                         problem_size=100, test_episodes_all=16, test_episodes=16, test_batch_size=8,
                         model_load={"enable": True, "path": ck, "file": "ckpt_tsp_nn.pt"})
    t = TSPTester(env_params, dict(MODEL_PARAMS, append_information=TSP_INFO), None, tester_params, None)
    gap = t.run()
    assert 5.0 < t.score < 20.0 and abs(gap) < 0.2


def test_reference_training_loop_on_the_swapped_in_model(device, monkeypatch):
    """The reference's own _train_one_batch loop (CVRP/CVRPTrainer.py:232-282), statement for statement, on this
    build's CVRPEnv / CVRPModel with a torch optimizer: model(...) in mode 'train' returns (teacher, prob, 1) and
    loss.backward() fills the parameters' .grad.  Compared with the batch captured from the reference."""
    from conftest import GOLDEN
    from dgl_amd import cvrp as C
    g = np.load(os.path.join(GOLDEN, "train_cvrp_n40_b4.npz"))
    w0 = np.load(os.path.join(GOLDEN, "weights_cvrp_init789.npz"))
    knn, depot_knn_nums = int(g["meta/knn"]), int(g["meta/depot_knn"])
    monkeypatch.setattr(C, "mix_augment", lambda xy: xy)            # the golden holds the augmented coordinates
    env = C.CVRPEnv(mode="train", pomo_size=1, append_information=CVRP_INFO, knn=knn, depot_knn=depot_knn_nums,
                    device=str(device))
    B, M = g["aug_xy"].shape[:2]
    env.batch_size, env.problem_size = B, M - 1
    env.depot_node_xy = torch.from_numpy(g["aug_xy"])
    env.depot_node_demand = torch.from_numpy(g["problems"][:, :, 2])
    env.solution = torch.from_numpy(g["solution"]).to(device)
    env.solution_flag = torch.from_numpy(g["solution_flag"]).to(device)
    solution0 = env.solution.clone()
    model = C.CVRPModel(**dict(MODEL_PARAMS, mode="train", append_information=CVRP_INFO)).to(device)
    model.load_state_dict({k: torch.from_numpy(w0[k]) for k in w0.files})
    optimizer = torch.optim.Adam(model.parameters(), lr=1e-4, weight_decay=1e-6)      # CVRP/train.py:49-54

    model.train()
    env.reset("train")
    prob_list = torch.ones(size=(B, 0), device=device)
    state, reward, done = env.pre_step()
    model.mode = "train"
    model.pre_forward(env.dis_matrix, env.batch_size)
    current_step = 0
    losses = []
    while not done:
        if current_step == 0:
            selected = env.solution[:, 0]
            prob = torch.ones(env.solution.shape[0], 1, device=device)
            is_via_depot = env.solution_flag[:, current_step] == 1
            selected[is_via_depot] += (env.problem_size + 1)
        else:
            selected, prob, probs = model(state, env.selected_node_list, env.solution, env.solution_flag, knn,
                                          current_step, depot_knn_nums)
            loss_mean = -prob.type(torch.float64).log().mean()
            model.zero_grad()
            loss_mean.backward()
            optimizer.step()
            losses.append(float(loss_mean))
        current_step += 1
        state, reward, done = env.step(selected)
        prob_list = torch.cat((prob_list, prob.detach()), dim=1)
    assert torch.equal(env.solution, solution0)              # the in-place rewrites of the teacher's entries cancel
    assert model.decoder.k_1.weight.grad is None             # never used by the forward pass, as in the reference
    probs = prob_list[:, 1:].t().cpu().numpy()
    rel = np.abs(probs - g["prob"]) / g["prob"]
    print(f"reference loop on the swapped-in model: {len(losses)} Adam steps, teacher probabilities max relative "
          f"difference {rel.max():.2e}, median {np.median(rel):.2e}")
    np.testing.assert_allclose(np.array(losses), g["loss"], rtol=2e-3, atol=2e-4)
    assert np.median(rel) < 1e-4 and rel.max() < 2e-2


def test_reference_tsp_training_loop_on_the_swapped_in_model(device, monkeypatch):
    """The TSP counterpart: the reference's _train_one_batch loop (TSP/TSPTrainer.py:216-256), statement for statement,
    on this build's TSPEnv / TSPModel with a torch optimizer - model(state, selected_node_list, solution, knn, step) in
    mode 'train' returns (teacher, prob, 1) and loss.backward() fills the parameters' .grad.  Labels, augmented
    coordinates and the teacher probabilities of every step come from one training batch of the reference's
    self-improvement loop (selfimp_tsp_n40_b6.npz, started from the checkpoint fixture)."""
    from conftest import GOLDEN
    from dgl_amd import tsp as T
    from dgl_amd.weights import load_checkpoint
    g = np.load(os.path.join(GOLDEN, "selfimp_tsp_n40_b6.npz"))
    knn = int(g["meta/knn"])
    monkeypatch.setattr(T, "mix_augment", lambda xy: xy)            # the golden holds the augmented coordinates
    env = T.TSPEnv(mode="train", pomo_size=1, append_information=TSP_INFO, knn=knn, device=str(device))
    B, N = g["train/aug_xy"].shape[:2]
    env.batch_size, env.problem_size = B, N
    env.problems = torch.from_numpy(g["train/aug_xy"])
    last_chunk = max(int(k.split("/")[0][5:]) for k in g.files if k.startswith("chunk"))
    env.solution = torch.from_numpy(g[f"chunk{last_chunk}/aug_solution"]).to(device)
    model = T.TSPModel(**dict(MODEL_PARAMS, mode="train", append_information=TSP_INFO)).to(device)
    model.load_state_dict(load_checkpoint(os.path.join(GOLDEN, "ckpt_tsp_nn.pt")))
    optimizer = torch.optim.Adam(model.parameters(), lr=1e-4, weight_decay=1e-6)      # TSP/train.py: the same settings

    model.train()
    env.reset("train")
    prob_list = torch.ones(size=(B, 0), device=device)
    state, reward, done = env.pre_step()
    model.mode = "train"
    model.pre_forward(env.dis_matrix, env.batch_size)
    current_step = 0
    while not done:
        if current_step == 0:
            selected = env.solution[:, 0]
            prob = torch.ones(env.solution.shape[0], 1, device=device)
        else:
            selected, prob, probs = model(state, env.selected_node_list, env.solution, knn, current_step)
            loss_mean = -prob.type(torch.float64).log().mean()
            model.zero_grad()
            loss_mean.backward()
            optimizer.step()
        current_step += 1
        state, reward, done = env.step(selected)
        prob_list = torch.cat((prob_list, prob.detach()), dim=1)
    assert model.decoder.k_1.weight.grad is None and model.decoder.embedding_last_node_pos.weight.grad is None
    assert model.encoder.embedding.weight.grad is not None
    probs = prob_list[:, 1:].t().cpu().numpy()
    rel = np.abs(probs - g["train/prob"]) / g["train/prob"]
    likely = g["train/prob"] >= 1e-3
    print(f"reference TSP loop on the swapped-in model: {probs.shape[0]} Adam steps, teacher probabilities max relative "
          f"difference {rel.max():.2e} (likely tokens {rel[likely].max():.2e}), median {np.median(rel):.2e}; "
          f"first 12 steps {rel[:12].max():.2e}")
    # the bounds of tests/test_gpu_selfimprove.py::_check_training (the same batch through PolicyTrainer; observed 4e-2)
    assert np.median(rel) < 1e-4 and rel[:12].max() < 1e-3 and rel[likely].max() < 0.1 and rel.max() < 0.15
