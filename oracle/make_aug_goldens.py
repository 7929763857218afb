"""Golden vectors of the aug_test protocol, captured from the imported REFERENCE (build container
only): TSPEnv in test_mode 'aug_test' (instances repeated aug_size times, mix-augmented at reset,
TSP/TSPEnv.py:119-121,222-224), greedy decode with the trained fixture checkpoint, rewards on the
ORIGINAL coordinates (TSP/TSPEnv.py:508-509) and the best-of-augmentation reduction
(TSP/TSPTester.py:183-188).  Writes tests/golden/aug_tsp_n50_b3_a4.npz, and the same for CVRPEnv
(CVRP/CVRPEnv.py:250-257,462-467,864-865; CVRP/CVRPTester.py:197-199) as aug_cvrp_n40_b3_a4.npz.
--published-scale: one instance of 1000 nodes with the replica counts the reference evaluates that size with
(aug_tsp_n1000_b1_a32.npz, aug_cvrp_n1000_b1_a8.npz).

usage: python oracle/make_aug_goldens.py [--published-scale]
"""
from __future__ import annotations

import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_goldens import CVRP_INFO, TSP_INFO, _import_reference, model_params  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def main(B=3, N=50, aug=4, knn=30, name="aug_tsp_n50_b3_a4"):
    import torch
    TSPModel, TSPEnv, _, _, Beamsearch = _import_reference()
    ck = torch.load(os.path.join(GOLD, "ckpt_tsp_nn.pt"), map_location="cpu", weights_only=True)
    model = TSPModel(**model_params(TSP_INFO))
    model.load_state_dict(ck["model_state_dict"])
    model.eval()
    model.mode = "test"
    torch.manual_seed(99)
    xy = torch.rand(B, N, 2)
    env = TSPEnv(data_path="", baseline_path="", mode="test", test_mode="aug_test", aug_size=aug,
                 append_information=TSP_INFO, aggregation_nums=50, pomo_size=1)
    env.raw_data_nodes = xy
    env.load_problems(0, B, N, 1)
    env.start_idx = 0
    env.step_size = B
    with torch.no_grad():
        bs = Beamsearch(1, env.batch_size * env.pomo_size, N, torch.FloatTensor, torch.LongTensor, "logits", False)
        torch.manual_seed(777)                      # the RNG state mix_aug draws its angles from
        env.reset("test")
        aug_xy = env.step_state.data[:, :, :2].numpy().copy()
        state, reward, done = env.pre_step()
        model.pre_forward(env.dis_matrix, env.batch_size)
        start = torch.tensor([7])
        step = 0
        while not done:
            if step == 0:
                sel = start.repeat(env.batch_size).repeat_interleave(env.beam_size).to(torch.int64)
            else:
                _, probs, _ = model(state, env.selected_node_list, None, knn, step)
                env.selected_node_list = bs.advance(torch.log(probs.view(env.batch_size * env.pomo_size, 1, -1)), env,
                                                    knn, step)
                sel = bs.next_nodes[-1].view(-1)
            step += 1
            state, reward, done = env.step(sel)
        view = reward.view(env.batch_size // env.aug_size, env.aug_size)
        shortest, index = torch.min(view, dim=1)
    out = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(out, xy=xy.numpy(), aug_xy=aug_xy, tour=env.selected_node_list.numpy().astype(np.int32),
                        reward=reward.numpy(), shortest=shortest.numpy(), index=index.numpy(), start=start.numpy(),
                        **{"meta/knn": np.int64(knn), "meta/aug": np.int64(aug), "meta/seed": np.int64(777),
                           "meta/torch": np.array(torch.__version__)})
    print("saved", out, "shortest", shortest.numpy(), "index", index.numpy())


def main_cvrp(B=3, N=40, aug=4, knn=15, kd=15, name="aug_cvrp_n40_b3_a4"):
    import torch
    _, _, CVRPModel, CVRPEnv, Beamsearch = _import_reference()
    ck = torch.load(os.path.join(GOLD, "ckpt_cvrp_nn.pt"), map_location="cpu", weights_only=True)
    model = CVRPModel(**model_params(CVRP_INFO))
    model.load_state_dict(ck["model_state_dict"])
    model.eval()
    model.mode = "test"
    torch.manual_seed(98)
    depot_xy, node_xy = torch.rand(B, 2), torch.rand(B, N, 2)
    demand = torch.randint(1, 200, size=(B, N)) / 500.0
    env = CVRPEnv(mode="test", test_mode="aug_test", aug_size=aug, append_information=CVRP_INFO, aggregation_nums=10,
                  pomo_size=1)
    env.raw_depot_node_xy, env.raw_node_xy, env.raw_node_demand = depot_xy, node_xy, demand
    env.beam_size = 1
    env.problem_size = N
    env.load_problems(0, B, N)
    env.start_idx = 0
    env.step_size = B
    with torch.no_grad():
        bs = Beamsearch(1, env.batch_size * env.pomo_size, N + 1, torch.FloatTensor, torch.LongTensor, "logits", False,
                        "CVRP")
        torch.manual_seed(778)
        env.reset("test")
        aug_xy = env.step_state.data[:, :, :2].numpy().copy()
        state, reward, done = env.pre_step()
        model.pre_forward(env.dis_matrix, env.batch_size)
        start = torch.tensor([5])
        step = 0
        while not done:
            if step == 0:
                sel = start.repeat(env.batch_size).repeat_interleave(env.beam_size).to(torch.int64) + 2 + N
            else:
                _, probs, _ = model(state, env.selected_node_list, None, None, knn, step, kd, env.step_state.selected_flag)
                env.selected_node_list = bs.advance(torch.log(probs.view(env.batch_size * env.pomo_size, 1, -1)), env,
                                                    knn, step)
                sel = bs.next_nodes[-1].view(-1)
            step += 1
            state, reward, done = env.step(sel.clone())
        view = reward.view(env.batch_size // env.aug_size, env.aug_size)
        shortest, index = torch.min(view, dim=1)
    out = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(out, depot_xy=depot_xy.numpy(), node_xy=node_xy.numpy(), demand=demand.numpy(), aug_xy=aug_xy,
                        nodes=env.selected_node_list.numpy().astype(np.int32),
                        flags=env.step_state.selected_flag.numpy().astype(np.int8),
                        reward=reward.numpy(), shortest=shortest.numpy(), index=index.numpy(), start=start.numpy(),
                        **{"meta/knn": np.int64(knn), "meta/depot_knn": np.int64(kd), "meta/aug": np.int64(aug),
                           "meta/seed": np.int64(778), "meta/torch": np.array(torch.__version__)})
    print("saved", out, "shortest", shortest.numpy(), "index", index.numpy())


if __name__ == "__main__":
    if "--published-scale" in sys.argv:
        # the augmentation widths the reference evaluates N = 1000 with (TSP/test.py:126: 32 replicas; CVRP: 8)
        main(B=1, N=1000, aug=32, name="aug_tsp_n1000_b1_a32")
        main_cvrp(B=1, N=1000, aug=8, name="aug_cvrp_n1000_b1_a8")
    else:
        main()
        main_cvrp()
