"""Golden vectors of the training step, captured from the imported REFERENCE (build container only):
one batch of CVRPTrainer._train_one_batch (CVRP/CVRPTrainer.py:232-282) driven exactly as the
trainer drives it - env.reset('train') (mix-augmented coordinates, distance matrix over
(x, y, demand)), teacher-forced decode, loss = -log(prob).double().mean(), backward, Adam step per
decode step - on labels from the reference's nearest-neighbour heuristic.

Writes tests/golden/train_cvrp_n40_b4.npz: instance, augmented coordinates, labels, initial weights
are NOT stored (seeded random init is rebuilt from weights_cvrp_init789.npz); per decode step the
teacher probabilities and loss; for the first three steps per-tensor gradient and parameter
checksums (L2 norm, sum) and the small tensors in full; the final loss and parameter checksums.

usage: python oracle/make_train_goldens.py
"""
from __future__ import annotations

import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_goldens import CVRP_INFO, _import_reference, model_params  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def main():
    import torch
    from torch.optim import Adam
    _, _, CVRPModel, CVRPEnv, _ = _import_reference()
    B, N, knn, kd = 4, 40, 15, 15
    w0 = np.load(os.path.join(GOLD, "weights_cvrp_init789.npz"))
    mp = model_params(CVRP_INFO)
    mp["mode"] = "train"
    model = CVRPModel(**mp)
    model.load_state_dict({k: torch.from_numpy(w0[k]) for k in w0.files})
    opt = Adam(model.parameters(), lr=1e-4, weight_decay=1e-6)                  # CVRP/train.py:49-54
    env = CVRPEnv(mode="train", pomo_size=1, append_information=CVRP_INFO, aggregation_nums=10, test_mode="none")
    torch.manual_seed(2024)
    env.load_problems(0, B, N)
    env.greedy_search()
    rec = {"problems": env.problems.numpy().copy(), "solution": env.solution.numpy().copy(),
           "solution_flag": env.solution_flag.numpy().copy(),
           "meta/knn": np.int64(knn), "meta/depot_knn": np.int64(kd)}
    model.train()
    env.reset("train")
    rec["aug_xy"] = env.step_state.data[:, :, :2].numpy().copy()
    state, _, done = env.pre_step()
    model.mode = "train"
    model.pre_forward(env.dis_matrix, env.batch_size)
    names = [k for k, _ in model.named_parameters()]
    step, probs, losses = 0, [], []
    while not done:
        if step == 0:
            selected = env.solution[:, 0]
            via = env.solution_flag[:, 0] == 1
            selected[via] += N + 1
        else:
            selected, prob, _ = model(state, env.selected_node_list, env.solution, env.solution_flag, knn, step, kd)
            loss = -prob.type(torch.float64).log().mean()
            model.zero_grad()
            loss.backward()
            if step <= 3:
                for k, p in model.named_parameters():
                    g = p.grad
                    if g is None:
                        continue
                    rec[f"grad{step}/norm/{k}"] = np.float64(g.double().norm())
                    rec[f"grad{step}/sum/{k}"] = np.float64(g.double().sum())
                    if g.numel() <= 1024:
                        rec[f"grad{step}/full/{k}"] = g.numpy().copy()
            opt.step()
            if step <= 3:
                for k, p in model.named_parameters():
                    rec[f"param{step}/sum/{k}"] = np.float64(p.detach().double().sum())
                    if p.numel() <= 1024:
                        rec[f"param{step}/full/{k}"] = p.detach().numpy().copy()
            probs.append(prob.detach().numpy().reshape(-1).copy())
            losses.append(loss.item())
        step += 1
        state, _, done = env.step(selected)
    rec["prob"] = np.stack(probs, 0)            # [N-1][B]
    rec["loss"] = np.array(losses)
    for k, p in model.named_parameters():
        rec[f"final/sum/{k}"] = np.float64(p.detach().double().sum())
        rec[f"final/norm/{k}"] = np.float64(p.detach().double().norm())
    rec["final/Linear_final.weight"] = model.state_dict()["decoder.Linear_final.weight"].numpy().copy()
    rec["meta/torch"] = np.array(torch.__version__)
    out = os.path.join(GOLD, "train_cvrp_n40_b4.npz")
    np.savez_compressed(out, **rec)
    print("saved", out, os.path.getsize(out), "bytes; loss first/last", losses[0], losses[-1], "params", len(names))


if __name__ == "__main__":
    main()
