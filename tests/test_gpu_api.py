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


@pytest.mark.parametrize("name", ["tsp_n100_b64_nn", "tsp_n200_b2_beam8_nn", "tsp_n60_b2_beam4_pomo2"])
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


@pytest.mark.parametrize("name", ["cvrp_n100_b64_nn", "cvrp_n150_b2_beam8_nn"])
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
