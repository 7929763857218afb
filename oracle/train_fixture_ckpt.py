"""Train small checkpoint fixtures with the REFERENCE model/env (build container only).

The pretrained checkpoints the reference's scripts load are not shipped with it, and
random-initialised weights give near-uniform action distributions whose arg-max flips
under 1e-6 perturbations, which makes "identical greedy tours" untestable.  This
harness drives the reference's own classes through its teacher-forced training step
(TSP/TSPTrainer.py:216-256, CVRP/CVRPTrainer.py:232-282: one Adam step per decode
step, loss = -log p(teacher action) in fp64) on labels from the reference's
nearest-neighbour heuristic (TSPEnv.greedy_search / CVRPEnv.greedy_search), for a fixed
wall-clock budget, and saves the result in the reference's checkpoint layout
({'epoch', 'model_state_dict', ...}) under tests/golden/.

usage: python oracle/train_fixture_ckpt.py tsp|cvrp --minutes 10 --threads 3
"""
from __future__ import annotations

import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_goldens import CVRP_INFO, TSP_INFO, _import_reference, model_params  # noqa: E402


def train_tsp(minutes, B, N, knn, out):
    import torch
    from torch.optim import Adam
    TSPModel, TSPEnv, _, _, _ = _import_reference()
    torch.manual_seed(123)
    mp = model_params(TSP_INFO)
    mp["mode"] = "train"
    model = TSPModel(**mp)
    opt = Adam(model.parameters(), lr=1e-4, weight_decay=1e-6)
    env = TSPEnv(data_path="", mode="train", pomo_size=1, append_information=TSP_INFO, aggregation_nums=50,
                 test_mode="none")
    t0, it = time.time(), 0
    while time.time() - t0 < minutes * 60:
        env.load_problems(0, B, N, 1)
        env.greedy_search()
        model.train()
        env.reset("train")
        state, _, done = env.pre_step()
        model.mode = "train"
        model.pre_forward(env.dis_matrix, env.batch_size)
        step, losses = 0, []
        while not done:
            if step == 0:
                selected = env.solution[:, 0]
            else:
                selected, prob, _ = model(state, env.selected_node_list, env.solution, knn, step)
                loss = -prob.type(torch.float64).log().mean()
                model.zero_grad()
                loss.backward()
                opt.step()
                losses.append(loss.item())
            step += 1
            state, _, done = env.step(selected)
        it += 1
        if it % 5 == 0:
            print(f"tsp iter {it} loss {sum(losses) / len(losses):.4f} elapsed {time.time() - t0:.0f}s", flush=True)
    torch.save({"epoch": it, "model_state_dict": model.state_dict()}, out)
    print("saved", out)


def train_cvrp(minutes, B, N, knn, depot_knn, out):
    import torch
    from torch.optim import Adam
    _, _, CVRPModel, CVRPEnv, _ = _import_reference()
    torch.manual_seed(123)
    mp = model_params(CVRP_INFO)
    mp["mode"] = "train"
    model = CVRPModel(**mp)
    opt = Adam(model.parameters(), lr=1e-4, weight_decay=1e-6)
    env = CVRPEnv(mode="train", pomo_size=1, append_information=CVRP_INFO, aggregation_nums=10,
                  test_mode="none")
    t0, it = time.time(), 0
    while time.time() - t0 < minutes * 60:
        env.load_problems(0, B, N)
        env.greedy_search()
        model.train()
        env.reset("train")
        state, _, done = env.pre_step()
        model.mode = "train"
        model.pre_forward(env.dis_matrix, env.batch_size)
        step, losses = 0, []
        while not done:
            if step == 0:
                selected = env.solution[:, 0]
                via = env.solution_flag[:, 0] == 1
                selected[via] += N + 1
            else:
                selected, prob, _ = model(state, env.selected_node_list, env.solution, env.solution_flag,
                                          knn, step, depot_knn)
                loss = -prob.type(torch.float64).log().mean()
                model.zero_grad()
                loss.backward()
                opt.step()
                losses.append(loss.item())
            step += 1
            state, _, done = env.step(selected)
        it += 1
        if it % 5 == 0:
            print(f"cvrp iter {it} loss {sum(losses) / len(losses):.4f} elapsed {time.time() - t0:.0f}s", flush=True)
    torch.save({"epoch": it, "model_state_dict": model.state_dict()}, out)
    print("saved", out)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("problem", choices=["tsp", "cvrp"])
    ap.add_argument("--minutes", type=float, default=10)
    ap.add_argument("--threads", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--n", type=int, default=0)
    args = ap.parse_args()
    import torch
    torch.set_num_threads(args.threads)
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
    if args.problem == "tsp":
        train_tsp(args.minutes, args.batch, args.n or 50, 30, os.path.join(gold, "ckpt_tsp_nn.pt"))
    else:
        train_cvrp(args.minutes, args.batch, args.n or 30, args.n or 30, args.n or 30, os.path.join(gold, "ckpt_cvrp_nn.pt"))
