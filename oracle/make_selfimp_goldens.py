"""Golden vectors of one loop of the self-improvement training epoch, captured from the imported
REFERENCE (build container only).  The trainer modules of the reference import a package path that
does not exist in the checkout, so Env / Model / Beamsearch are driven directly in the order of
CVRP/CVRPTrainer.py:90-109,185-218,284-398 and TSP/TSPTrainer.py:89-105,178-202,258-321:

  load_problems -> greedy_search -> random_replace -> per chunk of step_size instances: validation
  beam rollout, keep-if-shorter, label augmentation (CVRP: data_augmentation; TSP: roll) ->
  one teacher-forced training batch (Adam lr 1e-4, weight decay 1e-6).

Every random draw that shapes the result is stored beside it (instances, replaced slots, start
nodes, augmentation draws, augmented coordinates), so the GPU tests do not depend on RNG parity.
Writes tests/golden/selfimp_cvrp_n30_b6.npz and selfimp_tsp_n40_b6.npz.

usage: python oracle/make_selfimp_goldens.py
"""
from __future__ import annotations

import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_goldens import CVRP_INFO, TSP_INFO, _import_reference, model_params  # noqa: E402

TRAIN_SEED = 999
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


class RecordDraws:
    """Logs the values torch.rand / torch.randint return while active."""

    def __init__(self, torch):
        self.torch, self.log = torch, []

    def __enter__(self):
        t = self.torch
        self._rand, self._randint = t.rand, t.randint

        def rand(*a, **k):
            v = self._rand(*a, **k)
            self.log.append(("rand", v.clone()))
            return v

        def randint(*a, **k):
            v = self._randint(*a, **k)
            self.log.append(("randint", v.clone()))
            return v

        t.rand, t.randint = rand, randint
        return self

    def __exit__(self, *exc):
        self.torch.rand, self.torch.randint = self._rand, self._randint


def cvrp():
    import torch
    from torch.optim import Adam
    _, _, CVRPModel, CVRPEnv, Beamsearch = _import_reference()
    B, N, step_size, beam, knn, kd, replace = 6, 30, 3, 2, 15, 15, 2
    ck = torch.load(os.path.join(GOLD, "ckpt_cvrp_nn.pt"), map_location="cpu", weights_only=True)
    mp = model_params(CVRP_INFO)
    model = CVRPModel(**mp)
    model.load_state_dict(ck["model_state_dict"])
    opt = Adam(model.parameters(), lr=1e-4, weight_decay=1e-6)
    env = CVRPEnv(mode="train", pomo_size=1, append_information=CVRP_INFO, aggregation_nums=10, test_mode="none")
    torch.manual_seed(4321)
    env.load_problems(0, B, N)
    env.step_size, env.beam_size = step_size, beam
    rec = {"problems0": env.problems.numpy().copy()}
    env.greedy_search()
    rec["greedy/solution"] = env.solution.numpy().copy()
    rec["greedy/flag"] = env.solution_flag.numpy().copy()
    rec["greedy/len"] = env.solution_len.numpy().copy()
    env.replace_size = replace
    random_index = env.random_replace()
    rec["replace/index"] = random_index.numpy().copy()
    rec["problems1"] = env.problems.numpy().copy()
    env.start_idx = 0
    chunk = 0
    while env.start_idx < B:
        with torch.no_grad():
            bs = Beamsearch(beam, step_size * env.pomo_size, N + 1, torch.FloatTensor, torch.LongTensor, "logits", False,
                            "CVRP")
            env.reset("valid")
            state, reward, done = env.pre_step()
            model.eval()
            model.mode = "valid"
            model.pre_forward(env.dis_matrix, env.step_size)
            step = 0
            while not done:
                if step == 0:
                    start = torch.randperm(N)[:env.pomo_size]
                    rec[f"chunk{chunk}/start"] = start.numpy().copy()
                    sel = start.repeat(step_size).repeat_interleave(beam).to(torch.int64) + 2 + N
                else:
                    _, probs, _ = model(state, env.selected_node_list, None, None, knn, step, kd)
                    env.selected_node_list = bs.advance(torch.log(probs.view(step_size * env.pomo_size, beam, -1)), env,
                                                        knn, step)
                    sel = bs.next_nodes[-1].view(-1)
                step += 1
                state, reward, done = env.step(sel)
            view = reward.view(step_size, beam * env.pomo_size)
            lens, index = torch.min(view, dim=1)
            pick = index[:, None].unsqueeze(1).expand(step_size, 1, N)
            tours = torch.gather(env.selected_node_list.view(step_size, beam * env.pomo_size, -1), 1, pick).squeeze(1)
            flags = torch.gather(env.step_state.selected_flag.view(step_size, beam * env.pomo_size, -1), 1, pick).squeeze(1)
            rec[f"chunk{chunk}/reward"] = reward.numpy().copy()
            rec[f"chunk{chunk}/tours"] = tours.numpy().copy()
            rec[f"chunk{chunk}/flags"] = flags.numpy().copy()
            rec[f"chunk{chunk}/lens"] = lens.numpy().copy()
            sl = slice(env.start_idx, env.start_idx + step_size)
            better = torch.gt(env.solution_len[sl], lens)
            env.solution[sl][better] = tours[better]
            env.solution_flag[sl][better] = flags[better]
            env.solution_len[sl][better] = lens[better]
            rec[f"chunk{chunk}/kept_solution"] = env.solution.numpy().copy()
            rec[f"chunk{chunk}/kept_flag"] = env.solution_flag.numpy().copy()
            rec[f"chunk{chunk}/kept_len"] = env.solution_len.numpy().copy()
            counts = env.solution_flag.sum(dim=1).tolist()
            with RecordDraws(torch) as d:
                env.data_augmentation()
            flip = np.zeros((B, N), dtype=np.uint8)
            shift = np.zeros(B, dtype=np.int32)
            it = iter(d.log)
            for i, S in enumerate(counts):
                for j in range(1, int(S)):
                    kind, v = next(it)
                    assert kind == "rand"
                    flip[i, j] = 1 if v.item() > 0.5 else 0
                kind, v = next(it)
                assert kind == "randint"
                shift[i] = int(v.item())
            assert next(it, None) is None
            rec[f"chunk{chunk}/flip"], rec[f"chunk{chunk}/shift"] = flip, shift
            rec[f"chunk{chunk}/aug_solution"] = env.solution.numpy().copy()
            rec[f"chunk{chunk}/aug_flag"] = env.solution_flag.numpy().copy()
        env.start_idx += step_size
        chunk += 1
    # one training batch
    model.train()
    torch.manual_seed(TRAIN_SEED)               # the generator state reset('train') draws its augmentation from
    env.reset("train")
    rec["train/aug_xy"] = env.step_state.data[:, :, :2].numpy().copy()
    state, _, done = env.pre_step()
    model.mode = "train"
    model.pre_forward(env.dis_matrix, env.batch_size)
    step, probs = 0, []
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
            opt.step()
            probs.append(prob.detach().numpy().reshape(-1).copy())
        step += 1
        state, _, done = env.step(selected)
    rec["train/prob"] = np.stack(probs, 0)
    rec["train/solution_after"] = env.solution.numpy().copy()
    for k, p in model.named_parameters():
        rec[f"final/sum/{k}"] = np.float64(p.detach().double().sum())
        if p.numel() <= 1024:
            rec[f"final/full/{k}"] = p.detach().numpy().copy()
    rec.update({"meta/knn": np.int64(knn), "meta/depot_knn": np.int64(kd), "meta/beam": np.int64(beam),
                "meta/step_size": np.int64(step_size), "meta/replace": np.int64(replace), "meta/train_seed": np.int64(TRAIN_SEED),
                "meta/torch": np.array(torch.__version__)})
    out = os.path.join(GOLD, "selfimp_cvrp_n30_b6.npz")
    np.savez_compressed(out, **rec)
    print("saved", out, os.path.getsize(out), "bytes; chunks", chunk, "label lens", rec[f"chunk{chunk - 1}/kept_len"])


def tsp():
    import torch
    from torch.optim import Adam
    TSPModel, TSPEnv, _, _, Beamsearch = _import_reference()
    B, N, step_size, beam, knn, replace = 6, 40, 3, 2, 30, 2
    ck = torch.load(os.path.join(GOLD, "ckpt_tsp_nn.pt"), map_location="cpu", weights_only=True)
    model = TSPModel(**model_params(TSP_INFO))
    model.load_state_dict(ck["model_state_dict"])
    opt = Adam(model.parameters(), lr=1e-4, weight_decay=1e-6)
    env = TSPEnv(data_path="", mode="train", pomo_size=1, append_information=TSP_INFO, aggregation_nums=50,
                 test_mode="none")
    torch.manual_seed(8765)
    env.step_size = step_size
    env.load_problems(0, B, N, beam)
    rec = {"problems0": env.problems.numpy().copy()}
    env.greedy_search()
    rec["greedy/solution"] = env.solution.numpy().copy()
    rec["greedy/len"] = env.solution_len.numpy().copy()
    env.replace_size = replace
    random_index = env.random_replace()
    rec["replace/index"] = random_index.numpy().copy()
    rec["problems1"] = env.problems.numpy().copy()
    env.start_idx = 0
    chunk = 0
    while env.start_idx < B:
        with torch.no_grad():
            bs = Beamsearch(beam, step_size * env.pomo_size, N, torch.FloatTensor, torch.LongTensor, "logits", False, "TSP")
            env.reset("valid")
            state, reward, done = env.pre_step()
            model.eval()
            model.mode = "valid"
            model.pre_forward(env.dis_matrix, env.step_size)
            step = 0
            while not done:
                if step == 0:
                    start = torch.randperm(N)[:env.pomo_size]
                    rec[f"chunk{chunk}/start"] = start.numpy().copy()
                    sel = start.repeat(step_size).repeat_interleave(beam).to(torch.int64)
                else:
                    _, probs, _ = model(state, env.selected_node_list, None, knn, step)
                    env.selected_node_list = bs.advance(torch.log(probs.view(step_size * env.pomo_size, beam, -1)), env,
                                                        knn, step)
                    sel = bs.next_nodes[-1].view(-1)
                step += 1
                state, reward, done = env.step(sel)
            view = reward.view(step_size, beam * env.pomo_size)
            lens, index = torch.min(view, dim=1)
            tours = torch.gather(env.selected_node_list.view(step_size, beam * env.pomo_size, -1), 1,
                                 index[:, None].unsqueeze(1).expand(step_size, 1, N)).squeeze(1)
            rec[f"chunk{chunk}/reward"] = reward.numpy().copy()
            rec[f"chunk{chunk}/tours"] = tours.numpy().copy()
            rec[f"chunk{chunk}/lens"] = lens.numpy().copy()
            sl = slice(env.start_idx, env.start_idx + step_size)
            better = torch.gt(env.solution_len[sl], lens)
            env.solution[sl][better] = tours[better]
            env.solution_len[sl][better] = lens[better]
            rec[f"chunk{chunk}/kept_solution"] = env.solution.numpy().copy()
            rec[f"chunk{chunk}/kept_len"] = env.solution_len.numpy().copy()
            shift = torch.randint(N, size=(1,), dtype=torch.int64)
            env.solution = torch.roll(env.solution, shifts=shift.item(), dims=1)
            rec[f"chunk{chunk}/shift"] = np.int64(shift.item())
            rec[f"chunk{chunk}/aug_solution"] = env.solution.numpy().copy()
        env.start_idx += step_size
        chunk += 1
    model.train()
    torch.manual_seed(TRAIN_SEED)               # the generator state reset('train') draws its augmentation from
    env.reset("train")
    rec["train/aug_xy"] = env.step_state.data[:, :, :2].numpy().copy()
    state, _, done = env.pre_step()
    model.mode = "train"
    model.pre_forward(env.dis_matrix, env.batch_size)
    step, probs = 0, []
    while not done:
        if step == 0:
            selected = env.solution[:, 0]
        else:
            selected, prob, _ = model(state, env.selected_node_list, env.solution, knn, step)
            loss = -prob.type(torch.float64).log().mean()
            model.zero_grad()
            loss.backward()
            opt.step()
            probs.append(prob.detach().numpy().reshape(-1).copy())
        step += 1
        state, _, done = env.step(selected)
    rec["train/prob"] = np.stack(probs, 0)
    for k, p in model.named_parameters():
        rec[f"final/sum/{k}"] = np.float64(p.detach().double().sum())
        if p.numel() <= 1024:
            rec[f"final/full/{k}"] = p.detach().numpy().copy()
    rec.update({"meta/knn": np.int64(knn), "meta/beam": np.int64(beam), "meta/step_size": np.int64(step_size),
                "meta/replace": np.int64(replace), "meta/train_seed": np.int64(TRAIN_SEED), "meta/torch": np.array(torch.__version__)})
    out = os.path.join(GOLD, "selfimp_tsp_n40_b6.npz")
    np.savez_compressed(out, **rec)
    print("saved", out, os.path.getsize(out), "bytes; chunks", chunk, "label lens", rec[f"chunk{chunk - 1}/kept_len"])


if __name__ == "__main__":
    cvrp()
    tsp()
