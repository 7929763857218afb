"""Self-improvement training drivers with the reference's interface: TSPTrainer
(TSP/TSPTrainer.py:30-322) and CVRPTrainer (CVRP/CVRPTrainer.py:31-398).

One loop of an epoch = replace part of the instance pool (env.random_replace), roll the current
policy out over the pool in chunks of ``step_size`` instances and keep every rollout that beats the
stored label (beamsearch_tour_nodes_shortest), re-phrase the labels (TSP: roll; CVRP: sub-route
reversal / rotation), then one teacher-forced batch with an Adam update per decode step
(_train_one_batch).  Rollouts, labels and the training step stay on the device; the host RNG calls
(instance pool, start nodes, label draws, augmentation angles) are the reference's, in its order.

With torch.distributed initialised every rank owns its own instance pool (host generator seeded
123 + rank after the identically seeded weight initialisation) and the gradients are all-reduced
inside PolicyTrainer.optimizer_step.
"""
from __future__ import annotations

import logging
import os
import time

import numpy as np
import torch

from . import distributed as D
from ._lib import PROBLEM_CVRP, PROBLEM_TSP
from .cvrp import CVRPEnv, CVRPModel
from .engine import PackedPolicy
from .tester import AverageMeter, set_seed
from .train import PolicyTrainer, train_one_batch_cvrp, train_one_batch_tsp
from .tsp import TSPEnv, TSPModel


class LogData:
    """Named series of [x, y] pairs, the 'result_log' entry of a checkpoint (utils/utils.py:507-593)."""

    def __init__(self):
        self.keys = set()
        self.data = {}

    def get_raw_data(self):
        return self.keys, self.data

    def set_raw_data(self, r_data):
        self.keys, self.data = r_data

    def has_key(self, key):
        return key in self.keys

    def get_keys(self):
        return self.keys

    def append(self, key, *args):
        if len(args) == 1:
            v = args[0]
            if isinstance(v, (int, float)):
                value = [len(self.data[key]) if self.has_key(key) else 0, v]
            elif type(v) in (tuple, list):
                value = list(v)
            else:
                raise ValueError("Unsupported value type")
        elif len(args) == 2:
            value = [args[0], args[1]]
        else:
            raise ValueError("Unsupported value type")
        self.data.setdefault(key, []).append(value)
        self.keys.add(key)

    def append_all(self, key, *args):
        if len(args) == 1:
            xs, ys = list(range(len(args[0]))), args[0]
        elif len(args) == 2:
            xs, ys = args
        else:
            raise ValueError("Unsupported value type")
        rows = np.stack([xs, ys], axis=1).tolist()
        if key in self.keys:                   # the reference extends by the two columns here; rows are meant
            self.data[key].extend(rows)
        else:
            self.data[key] = rows
            self.keys.add(key)

    def get_last(self, key):
        return self.data[key][-1] if self.has_key(key) else None

    def getXY(self, key, start_idx=0):
        xs = [r[0] for r in self.data[key]]
        ys = [r[1] for r in self.data[key]]
        if start_idx == 0:
            return xs, ys
        if start_idx in xs:
            i = xs.index(start_idx)
            return xs[i:], ys[i:]
        raise KeyError("no start_idx value in X axis data.")

    def get(self, key):
        return self.getXY(key)[1]


class MultiStepLR:
    """torch.optim.lr_scheduler.MultiStepLR in its chainable form: a step multiplies the CURRENT learning rate by
    gamma (once per occurrence of the new epoch count among the milestones) and leaves it alone otherwise, so a
    learning rate restored from a checkpoint's optimizer state is kept on resume, as in the reference.  The trainers
    call step() once per saved checkpoint (every fifth epoch, CVRP/CVRPTrainer.py:126-136), not once per epoch, and
    set last_epoch = epoch - 1 after loading a checkpoint (TSP/TSPTrainer.py:75-84)."""

    def __init__(self, trainer, milestones, gamma):
        self.trainer, self.milestones, self.gamma = trainer, sorted(milestones), gamma
        self.base_lr = trainer.lr
        self.last_epoch = 0

    def step(self):
        self.last_epoch += 1
        hits = self.milestones.count(self.last_epoch)
        if hits:
            self.trainer.lr = self.trainer.lr * self.gamma ** hits

    def state_dict(self):
        return {"milestones": {m: 1 for m in self.milestones}, "gamma": self.gamma, "base_lrs": [self.base_lr],
                "last_epoch": self.last_epoch, "_last_lr": [self.trainer.lr]}


class _TrainerBase:
    Model = None
    Env = None
    problem = None
    problem_id = None

    def __init__(self, env_params, model_params, optimizer_params, trainer_params):
        self.env_params = dict(env_params)
        self.model_params = model_params
        self.optimizer_params = optimizer_params
        self.trainer_params = trainer_params
        self.logger = logging.getLogger("trainer")
        self.result_folder = trainer_params.get("result_folder", "./result")
        self.result_log = LogData()
        if not trainer_params.get("use_cuda", True) or not torch.cuda.is_available():
            raise RuntimeError("this build runs on an AMD GPU only: no CPU execution path exists")
        dev_num = int(os.environ.get("LOCAL_RANK", trainer_params.get("cuda_device_num", 0)))
        torch.cuda.set_device(dev_num)
        self.device = torch.device("cuda", dev_num)
        self.env_params["device"] = self.device
        self.env_params["knn"] = trainer_params["knn"]
        if "depot_knn" in trainer_params:
            self.env_params["depot_knn"] = trainer_params["depot_knn"]
        set_seed(123)                                              # TSP/TSPTrainer.py:63
        self.model = self.Model(**model_params).to(self.device)
        self.env = self.Env(**self.env_params)
        opt = dict(optimizer_params["optimizer"])
        self.trainer = PolicyTrainer(self.model.state_dict(), self.problem_id, self.device, lr=opt.get("lr", 1e-3),
                                     weight_decay=opt.get("weight_decay", 0.0), betas=opt.get("betas", (0.9, 0.999)),
                                     eps=opt.get("eps", 1e-8))
        sch = optimizer_params.get("scheduler", {"milestones": [], "gamma": 1.0})
        self.scheduler = MultiStepLR(self.trainer, sch["milestones"], sch["gamma"])
        self.start_epoch = 1
        load = trainer_params.get("model_load", {"enable": False})
        if load.get("enable"):
            self.load_checkpoint("{path}/checkpoint-{epoch}.pt".format(**load))
            self.start_epoch = 1 + load["epoch"]
            self.scheduler.last_epoch = load["epoch"] - 1
            self.logger.info("Saved Model Loaded !!")
        self._repack()
        rank, world = D.world()
        if world > 1:
            torch.manual_seed(123 + rank)

    # ---- weights ----------------------------------------------------------------------------
    def _repack(self):
        self.policy = PackedPolicy(self.trainer.state_dict(), self.problem_id, self.device)

    def checkpoint_dict(self, epoch):
        """The reference's checkpoint entries (CVRP/CVRPTrainer.py:128-135)."""
        return {"epoch": epoch,
                "model_state_dict": {k: v.cpu() for k, v in self.trainer.state_dict().items()},
                "optimizer_state_dict": self.trainer.optimizer_state_dict(),
                "scheduler_state_dict": self.scheduler.state_dict(),
                "result_log": self.result_log.get_raw_data()}

    def save_checkpoint(self, epoch):
        os.makedirs(self.result_folder, exist_ok=True)
        path = "{}/checkpoint-{}.pt".format(self.result_folder, epoch)
        torch.save(self.checkpoint_dict(epoch), path)
        return path

    def load_checkpoint(self, path):
        # a checkpoint holds tensors plus plain containers; 'result_log' is a (set, dict) pair
        ck = torch.load(path, map_location="cpu", weights_only=True)
        self.trainer.load_state_dict(ck["model_state_dict"])
        if "optimizer_state_dict" in ck:
            self.trainer.load_optimizer_state_dict(ck["optimizer_state_dict"])
        if "result_log" in ck:
            self.result_log.set_raw_data(ck["result_log"])
        self.model.load_state_dict(self.trainer.state_dict())

    # ---- epochs -----------------------------------------------------------------------------
    def run(self):
        tp = self.trainer_params
        t0 = time.time()
        self.env.step_size = tp["step_size"]
        self.env.beam_size = tp["beam_size"]
        self._load_pool()
        self.env.greedy_search()
        for epoch in range(self.start_epoch, tp["epochs"] + 1):
            self.logger.info("=================================================================")
            score, student, loss, replace = self._train_one_epoch(epoch)
            self.result_log.append("train_score", epoch, score)
            self.result_log.append("train_student_score", epoch, student)
            self.result_log.append("train_loss", epoch, loss)
            self.result_log.append("replace_score", epoch, replace)
            if epoch % 5 == 0 and epoch != 0:
                self.logger.info("Saving trained_model")
                if D.world()[0] == 0:
                    self.save_checkpoint(epoch)
                self.scheduler.step()
            done = epoch - self.start_epoch + 1
            elapsed = time.time() - t0
            remain = elapsed * (tp["epochs"] - epoch) / done
            fmt = lambda s: "{:.2f}h".format(s / 3600) if s > 3600 else "{:.2f}m".format(s / 60)
            self.logger.info("Epoch {:3d}/{:3d}: Time Est.: Elapsed[{}], Remain[{}]".format(
                epoch, tp["epochs"], fmt(elapsed), fmt(remain)))
            if epoch == tp["epochs"]:
                self.logger.info(" *** Training Done *** ")
                for key in self.result_log.get_keys():
                    self.logger.info("{} = {}".format(key + "_list", self.result_log.get(key)))
        self.model.load_state_dict(self.trainer.state_dict())

    def replace_size(self, epoch):
        """Linear schedule from max_replace_ratio to min_replace_ratio over the epochs (CVRP/CVRPTrainer.py:185-187)."""
        ep, tp = self.env_params, self.trainer_params
        return int(((ep["max_replace_ratio"] - ep["min_replace_ratio"]) * (1 - epoch / tp["epochs"])
                    + ep["min_replace_ratio"]) * tp["train_batch_size"])

    def _train_one_epoch(self, epoch, score_avg_=None, loops=None):
        tp = self.trainer_params
        score_AM, loss_AM, gap_AM, best_AM, replace_AM, result_AM = (AverageMeter() for _ in range(6))
        batch_size, beam_size = tp["train_batch_size"], tp["beam_size"]
        env = self.env
        env.replace_size = self.replace_size(epoch)
        for i in range(tp.get("loops_per_epoch", 100) if loops is None else loops):
            loss_AM.reset()
            random_index = env.random_replace()
            env.start_idx = 0
            while env.start_idx < batch_size:
                best_score, score, gap = self.beamsearch_tour_nodes_shortest(beam_size, env.step_size, self._num_nodes())
                gap_AM.update(gap, env.step_size)
                score_AM.update(score, env.step_size)
                best_AM.update(best_score, env.step_size)
                env.start_idx += env.step_size
            replace_score = torch.mean(env.solution_len[random_index.to(self.device)]).item()
            replace_AM.update(replace_score, batch_size)
            avg_loss = self._train_one_batch(i * batch_size, batch_size, epoch)
            result_AM.update(score, batch_size)
            loss_AM.update(avg_loss, batch_size)
            self.logger.info("Epoch {:3d}:  Loop:{:3d}  avg_score {:.4f} Loss: {:.4f} replace_score: {:.4f}".format(
                epoch, i, result_AM.avg, loss_AM.avg, replace_AM.avg))
            self.logger.info("best_score {:4f}:  score:{:4f}  gap {:.4f}".format(best_score, score, gap))
        return score_AM.avg, result_AM.avg, loss_AM.avg, replace_AM.avg

    def _keep_if_shorter(self, tours, lens, flags=None):
        """First rollout: adopt; later: overwrite the stored labels of this chunk where the rollout is shorter
        (CVRP/CVRPTrainer.py:375-383)."""
        env = self.env
        if env.solution is None:
            env.solution, env.solution_len = tours, lens
            if flags is not None:
                env.solution_flag = flags
            return
        sl = slice(env.start_idx, env.start_idx + env.step_size)
        better = torch.gt(env.solution_len[sl], lens)
        env.solution[sl] = torch.where(better[:, None], tours, env.solution[sl])
        if flags is not None:
            env.solution_flag[sl] = torch.where(better[:, None], flags, env.solution_flag[sl])
        env.solution_len[sl] = torch.where(better, lens, env.solution_len[sl])


class TSPTrainer(_TrainerBase):
    Model, Env, problem, problem_id = TSPModel, TSPEnv, "TSP", PROBLEM_TSP

    def _load_pool(self):
        tp = self.trainer_params
        self.env.load_problems(0, tp["train_batch_size"], tp["problem_size"], tp["beam_size"])

    def _num_nodes(self):
        return self.env.problem_size

    def _train_one_batch(self, episode, batch_size, epoch):
        """TSP/TSPTrainer.py:216-256."""
        env = self.env
        env.reset("train")
        loss = train_one_batch_tsp(self.trainer, env.engine, env.solution)
        self._repack()
        return loss

    @torch.no_grad()
    def beamsearch_tour_nodes_shortest(self, beam_size, batch_size, num_nodes, dtypeFloat=None, dtypeLong=None,
                                       probs_type="raw", random_start=False, start=None, shift=None):
        """Validation rollout of one chunk and label update (TSP/TSPTrainer.py:258-321).  ``start`` / ``shift``
        replace the random start node and the random label roll."""
        env = self.env
        env.beam_size = beam_size
        env.reset("valid")
        eng = env.engine
        idx = torch.randperm(env.problem_size)[:env.pomo_size] if start is None else start
        sel = idx.repeat(batch_size).repeat_interleave(beam_size).to(device=self.device, dtype=torch.int32)
        eng.set_knn(self.trainer_params["knn"])
        length = eng.greedy_rollout(self.policy, sel) if beam_size == 1 else eng.beam_rollout(self.policy, sel, beam_size)
        env.selected_count = eng.t
        group = beam_size * env.pomo_size
        shortest_lens, index = torch.min(length.view(batch_size, group), dim=1)
        tours = env.selected_node_list.view(batch_size, group, -1)
        shortest_tours = tours.gather(1, index[:, None, None].expand(batch_size, 1, num_nodes)).squeeze(1)
        self.last_shortest = (shortest_tours, shortest_lens.clone())
        self._keep_if_shorter(shortest_tours, shortest_lens.clone())
        best_score = env.solution_len.mean()
        score = shortest_lens.mean()
        if shift is None:
            shift = torch.randint(env.problem_size, size=(1,), dtype=torch.int64).item()
        env.solution = torch.roll(env.solution, shifts=int(shift), dims=1)       # TSP/TSPTrainer.py:315-317
        gap = (score - best_score) / best_score
        return best_score.item(), score.item(), gap.item()


class CVRPTrainer(_TrainerBase):
    Model, Env, problem, problem_id = CVRPModel, CVRPEnv, "CVRP", PROBLEM_CVRP

    def _load_pool(self):
        tp = self.trainer_params
        self.env.load_problems(0, tp["train_batch_size"], tp["problem_size"])

    def _num_nodes(self):
        return self.env.problem_size + 1

    def _train_one_batch(self, episode, batch_size, epoch):
        """CVRP/CVRPTrainer.py:232-282."""
        env = self.env
        env.reset("train")
        loss = train_one_batch_cvrp(self.trainer, env.engine, env.solution, env.solution_flag)
        self._repack()
        return loss

    @torch.no_grad()
    def beamsearch_tour_nodes_shortest(self, beam_size, batch_size, num_nodes, dtypeFloat=None, dtypeLong=None,
                                       probs_type="raw", random_start=False, start=None, draws=None):
        """Validation rollout of one chunk, label update and label augmentation (CVRP/CVRPTrainer.py:284-398).
        ``start`` replaces the random start customer, ``draws`` = (flip, shift) the augmentation's draws."""
        env = self.env
        env.beam_size = beam_size
        env.reset("valid")
        eng = env.engine
        N = env.problem_size
        idx = torch.randperm(N)[:env.pomo_size] if start is None else start
        sel = idx.repeat(batch_size).repeat_interleave(beam_size).to(device=self.device, dtype=torch.int32) + 2 + N
        eng.set_knn(self.trainer_params["knn"], self.trainer_params["depot_knn"])
        length = eng.greedy_rollout(self.policy, sel) if beam_size == 1 else eng.beam_rollout(self.policy, sel, beam_size)
        env.selected_count = eng.t
        group = beam_size * env.pomo_size
        shortest_lens, index = torch.min(length.view(batch_size, group), dim=1)
        pick = index[:, None, None].expand(batch_size, 1, num_nodes - 1)
        shortest_tours = env.selected_node_list.view(batch_size, group, -1).gather(1, pick).squeeze(1)
        shortest_flags = env.step_state.selected_flag.view(batch_size, group, -1).gather(1, pick).squeeze(1)
        self.last_shortest = (shortest_tours, shortest_flags, shortest_lens.clone())
        self._keep_if_shorter(shortest_tours, shortest_lens.clone(), shortest_flags)
        sl = slice(env.start_idx, env.start_idx + env.step_size)
        best_score = env.solution_len[sl].mean()
        score = shortest_lens.mean()
        gap = (score - best_score) / best_score
        env.data_augmentation(*(draws or (None, None)))
        return best_score.item(), score.item(), gap.item()
