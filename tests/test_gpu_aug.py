"""aug_test protocol (coordinate augmentation, best-of-augmentation on the original coordinates)
against the vectors captured from the reference (oracle/make_aug_goldens.py)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def _golden(name="aug_tsp_n50_b3_a4"):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


# the second of each pair: one instance of 1000 nodes with the replica count the reference evaluates that size with
# (TSP/test.py:126: 32; 8 for CVRP)
TSP_CASES = [("aug_tsp_n50_b3_a4", 0.9), ("aug_tsp_n1000_b1_a32", 0.95)]
CVRP_CASES = [("aug_cvrp_n40_b3_a4", 0.9), ("aug_cvrp_n1000_b1_a8", 0.85)]


@pytest.mark.parametrize("name", [c[0] for c in TSP_CASES])
def test_mix_augment_reproduces_reference_coordinates(name, device):
    from dgl_amd.augment import mix_augment
    g = _golden(name)
    aug = int(g["meta/aug"])
    xy = torch.from_numpy(g["xy"]).repeat_interleave(aug, dim=0).to(device)
    torch.manual_seed(int(g["meta/seed"]))
    out = mix_augment(xy).cpu().numpy()
    same_torch = str(g["meta/torch"]) == torch.__version__
    if same_torch:                       # same torch.rand / cos / sin: the kernel's arithmetic is exact
        assert np.array_equal(out, g["aug_xy"])
    np.testing.assert_allclose(out, g["aug_xy"], rtol=0, atol=2e-6)
    assert np.array_equal(out[0], g["xy"][0])           # only the first row of the batch stays as it is


@pytest.mark.parametrize("name,min_same", TSP_CASES)
def test_aug_test_env_and_tester_reduction(name, min_same, device):
    """TSPEnv in aug_test mode with the reference's loop: same tours, rewards on the original coordinates,
    best-of-augmentation per instance."""
    from dgl_amd import TSPEnv, TSPModel
    from dgl_amd.tester import TSPTester
    from dgl_amd.weights import load_checkpoint
    g = _golden(name)
    aug, knn = int(g["meta/aug"]), int(g["meta/knn"])
    B, N = g["xy"].shape[0], g["xy"].shape[1]
    info = [True, True, True, False, True, True, False, False, False, False, False]
    model_params = dict(mode="test", embedding_dim=128, sqrt_embedding_dim=128 ** 0.5, decoder_layer_num=3, qkv_dim=16,
                        head_num=8, ff_hidden_dim=512, append_information=info)
    tester = TSPTester(env_params=dict(mode="test", test_mode="aug_test", aug_size=aug, append_information=info,
                                       data_path="", aggregation_nums=50, pomo_size=1),
                       model_params=model_params,
                       tester_params=dict(use_cuda=True, cuda_device_num=0, knn=knn, beam_size=1, test_batch_size=B,
                                          problem_size=N, test_episodes=B, test_episodes_all=B,
                                          model_load={"enable": False}))
    tester.model.load_state_dict(load_checkpoint(os.path.join(GOLDEN, "ckpt_tsp_nn.pt")))
    env = tester.env
    env.raw_data_nodes = torch.from_numpy(g["xy"])
    env.load_problems(0, B, N, 1)
    env.start_idx, env.step_size = 0, B
    assert env.batch_size == B * aug
    torch.manual_seed(int(g["meta/seed"]))
    shortest = tester.fast_rollout(1, start=torch.from_numpy(g["start"]))
    tours = tester.last_tours.cpu().numpy()
    same = (tours == g["tour"]).all(axis=1).mean()
    print(f"aug_test: {same:.2f} of the {B * aug} augmented rollouts identical; best-of-aug {shortest.cpu().numpy()} "
          f"vs {g['shortest']}")
    assert same >= min_same
    identical = (tours == g["tour"]).all(axis=1)
    assert np.array_equal(tester.last_lengths.cpu().numpy()[identical], g["reward"][identical])    # the reference's own floats
    np.testing.assert_allclose(shortest.cpu().numpy(), g["shortest"], rtol=2e-4)


@pytest.mark.parametrize("name,min_same", CVRP_CASES)
def test_cvrp_aug_test_env_and_tester_reduction(name, min_same, device):
    """CVRPEnv in aug_test mode: reference-protocol loop and device-resident rollout against the reference's run."""
    from dgl_amd.tester import CVRPTester
    from dgl_amd.weights import load_checkpoint
    g = _golden(name)
    aug, knn, kd = int(g["meta/aug"]), int(g["meta/knn"]), int(g["meta/depot_knn"])
    B, N = g["node_xy"].shape[0], g["node_xy"].shape[1]
    info = [True, True, True, False, True, True, False, False, False, False, False, False, False]
    model_params = dict(mode="test", embedding_dim=128, sqrt_embedding_dim=128 ** 0.5, decoder_layer_num=3, qkv_dim=16,
                        head_num=8, ff_hidden_dim=512, append_information=info)
    tester = CVRPTester(env_params=dict(mode="test", test_mode="aug_test", aug_size=aug, append_information=info,
                                        data_path="", aggregation_nums=10, pomo_size=1),
                        model_params=model_params,
                        tester_params=dict(use_cuda=True, cuda_device_num=0, knn=knn, depot_knn=kd, beam_size=1,
                                           test_batch_size=B, problem_size=N, test_episodes=B, test_episodes_all=B,
                                           model_load={"enable": False}))
    tester.model.load_state_dict(load_checkpoint(os.path.join(GOLDEN, "ckpt_cvrp_nn.pt")))
    env = tester.env
    env.raw_depot_node_xy = torch.from_numpy(g["depot_xy"])
    env.raw_node_xy = torch.from_numpy(g["node_xy"])
    env.raw_node_demand = torch.from_numpy(g["demand"])
    env.problem_size = N
    env.load_problems(0, B, N)
    env.start_idx, env.step_size = 0, B
    assert env.batch_size == B * aug
    for protocol in ("fast", "reference"):
        torch.manual_seed(int(g["meta/seed"]))
        if protocol == "fast":
            shortest = tester.fast_rollout(1, start=torch.from_numpy(g["start"]))
        else:
            tester.beamsearch_tour_nodes_shortest(1, B, N + 1, start=torch.from_numpy(g["start"]))
            shortest, _ = tester._summarise(tester.last_lengths, 1)
        np.testing.assert_allclose(env.engine.xy.cpu().numpy(), g["aug_xy"], rtol=0, atol=2e-6)
        nodes, flags = tester.last_tours.cpu().numpy(), tester.last_flags.cpu().numpy()
        same = ((nodes == g["nodes"]) & (flags == g["flags"])).all(axis=1)
        print(f"cvrp aug_test ({protocol}): {same.mean():.2f} of the {B * aug} augmented rollouts identical; best-of-aug "
              f"{shortest.cpu().numpy()} vs {g['shortest']}")
        assert same.mean() >= min_same
        assert np.array_equal(tester.last_lengths.cpu().numpy()[same], g["reward"][same])      # the reference's own floats
        np.testing.assert_allclose(shortest.cpu().numpy(), g["shortest"], rtol=2e-4)


def test_library_drivers_on_fixture_files(device, tmp_path):
    """run_tsplib / run_cvrplib over synthetic library files (written here in the libraries' formats; the committed
    parser fixtures have fewer than 26 points, which the kernels refuse): the integer score is the rounded-up length
    of the returned best tour on the original coordinates and the tour is a valid one."""
    from dgl_amd import formats as F
    from dgl_amd.libeval import run_cvrplib, run_tsplib
    from dgl_amd.weights import load_checkpoint
    root = str(tmp_path)
    os.makedirs(os.path.join(root, "tsplib"))
    os.makedirs(os.path.join(root, "vrplib"))
    rng = np.random.RandomState(5)
    pts = rng.randint(0, 2000, size=(30, 2))
    with open(os.path.join(root, "tsplib", "syn30.tsp"), "w") as f:
        f.write("NAME : syn30\nCOMMENT : synthetic fixture\nTYPE : TSP\nDIMENSION : 30\nEDGE_WEIGHT_TYPE : EUC_2D\n"
                "NODE_COORD_SECTION\n")
        for i, (x, y) in enumerate(pts):
            f.write(f"{i + 1} {x} {y}.5\n" if i % 3 == 0 else f"{i + 1}  {x}   {y}\n")
        f.write("EOF\n")
    cpts, dem = rng.randint(0, 1000, size=(30, 2)), rng.randint(1, 30, size=30)
    dem[0] = 0
    with open(os.path.join(root, "vrplib", "S-n30-k8.vrp"), "w") as f:
        f.write("NAME : \tS-n30-k8\nCOMMENT : \t\"synthetic fixture\"\nTYPE : \tCVRP\nDIMENSION : \t30\n"
                "EDGE_WEIGHT_TYPE : \tEUC_2D\nCAPACITY : \t60\nNODE_COORD_SECTION\t\t\n")
        for i, (x, y) in enumerate(cpts):
            f.write(f"{i + 1}\t{x}\t{y}\n")
        f.write("DEMAND_SECTION\t\t\n")
        for i, d in enumerate(dem):
            f.write(f"{i + 1}\t{d}\n")
        f.write("DEPOT_SECTION\n\t1\n\t-1\nEOF\n")
    mp = lambda info: dict(mode="test", embedding_dim=128, sqrt_embedding_dim=128 ** 0.5, decoder_layer_num=3, qkv_dim=16,
                           head_num=8, ff_hidden_dim=512, append_information=info)
    tinfo = [True, True, True, False, True, True, False, False, False, False, False]
    lines = []
    res = run_tsplib(root, dict(mode="test", append_information=tinfo, aggregation_nums=5, pomo_size=1), mp(tinfo),
                     dict(use_cuda=True, cuda_device_num=0, knn=12, beam_size=2, model_load={"enable": False}),
                     names=["syn30"], best_known={"syn30": 9000},
                     weights=load_checkpoint(os.path.join(GOLDEN, "ckpt_tsp_nn.pt")), log=lines.append)
    length, gap, _ = res["syn30"]
    assert gap == length / 9000 - 1 and any(l.startswith("syn30 score") for l in lines)
    # a tour of 30 points spread over 2000 x 2000 is no shorter than the instance's extent and the model's is sane
    assert 4000 < length < 20000
    cinfo = tinfo + [False, False]
    res = run_cvrplib(root, dict(mode="test", append_information=cinfo, aggregation_nums=5, pomo_size=1), mp(cinfo),
                      dict(use_cuda=True, cuda_device_num=0, knn=6, depot_knn=4, beam_size=1, model_load={"enable": False}),
                      names=["S-n30-k8"], best_known={"S-n30-k8": 6000},
                      weights=load_checkpoint(os.path.join(GOLDEN, "ckpt_cvrp_nn.pt")), log=lines.append)
    length, gap, _ = res["S-n30-k8"]
    assert gap == length / 6000 - 1 and 3000 < length < 25000
    print("\n".join(lines))
