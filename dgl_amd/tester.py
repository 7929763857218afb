"""Evaluation drivers with the reference's interface: TSPTester (TSP/TSPTester.py:25-199) and
CVRPTester (CVRP/CVRPTester.py).  ``beamsearch_tour_nodes_shortest`` follows the reference's loop
through model / env / Beamsearch objects; ``run`` uses the device-resident fast path
(engine.greedy_rollout / beam_rollout) unless tester_params['reference_protocol'] is set, shards
the instances over the ranks of a torch.distributed job and all-reduces the statistics."""
from __future__ import annotations

import logging
import os
import random

import numpy as np
import torch

from . import distributed as D
from .beamsearch import Beamsearch
from .cvrp import CVRPEnv, CVRPModel
from .tsp import TSPEnv, TSPModel
from .weights import load_checkpoint


class AverageMeter:
    """utils/utils.py:490-504."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.sum = 0.0
        self.count = 0

    def update(self, val, n=1):
        self.sum += float(val) * n
        self.count += n

    @property
    def avg(self):
        return self.sum / self.count if self.count else 0.0


def set_seed(seed):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)


class _TesterBase:
    Model = None
    Env = None
    problem = None

    def __init__(self, env_params, model_params, optimizer_params=None, tester_params=None, valid_params=None):
        self.env_params = dict(env_params)
        self.model_params = model_params
        self.optimizer_params = optimizer_params
        self.tester_params = tester_params
        self.valid_params = valid_params
        self.logger = logging.getLogger("tester")
        if not tester_params.get("use_cuda", True) or not torch.cuda.is_available():
            raise RuntimeError("this build runs on an AMD GPU only: no CPU execution path exists")
        dev_num = int(os.environ.get("LOCAL_RANK", tester_params.get("cuda_device_num", 0)))
        torch.cuda.set_device(dev_num)
        self.device = torch.device("cuda", dev_num)
        self.env_params["device"] = self.device
        self.env_params.setdefault("knn", tester_params.get("knn", self.env_params.get("knn", 30)))
        if "depot_knn" in tester_params:
            self.env_params.setdefault("depot_knn", tester_params["depot_knn"])
        set_seed(789)                                              # TSP/TSPTester.py:60
        self.model = self.Model(**model_params).to(self.device)
        self.env = self.Env(**self.env_params)
        load = tester_params.get("model_load", {"enable": False})
        if load.get("enable"):
            self.model.load_state_dict(load_checkpoint("{path}/{file}".format(**load)))
            self.logger.info("Saved Model Loaded !!")
        self.model.eval()
        self.model.mode = "test"

    # ---- start nodes ---------------------------------------------------------------
    def _start_nodes(self):
        return torch.randperm(self.env.problem_size)[:self.env.pomo_size]       # TSP/TSPTester.py:165

    def _rank_batches(self, episodes, batch_size, problem_size):
        """The batches of the whole run in its own order, each with its start draw, cut down to this rank's block of
        instances: yields (first episode, count, start nodes).  The reference draws one set of start nodes per batch
        (TSP/TSPTester.py:165, CVRP/CVRPTester.py:172); drawing them for every batch of the run, on every rank, and
        never letting a rank's batch straddle two of the run's batches gives every instance the start node it gets in
        a single process, so the reduced score does not depend on the number of ranks.  (The random angles of aug_test
        are drawn per batch with batch-sized shapes and do depend on it.)"""
        rank, ws = D.world()
        lo, hi = D.shard_range(episodes, rank, ws)
        for g0 in range(0, episodes, batch_size):
            start = torch.randperm(problem_size)[:self.env.pomo_size]
            a, b = max(g0, lo), min(g0 + batch_size, episodes, hi)
            if a < b:
                yield a, b - a, start

    def _summarise(self, reward, beam_size):
        """Best of the beams / multi-starts (and augmentations, TSP/TSPTester.py:183-188) of every instance."""
        if self.env_params.get("test_mode") == "aug_test":
            a = self.env.aug_size
            view = reward.view(self.env.batch_size // a, beam_size * self.env.pomo_size * a)
        else:
            view = reward.view(self.env.batch_size, beam_size * self.env.pomo_size)
        shortest, index = torch.min(view, dim=1)
        self._group = view.shape
        return shortest, index

    def _best_of_group(self, rows: torch.Tensor, index: torch.Tensor) -> torch.Tensor:
        """Row of the best rollout of every instance (TSP/TSPTester.py:190-193)."""
        B, G = self._group
        return rows.view(B, G, -1)[torch.arange(B, device=rows.device), index]


class TSPTester(_TesterBase):
    Model, Env, problem = TSPModel, TSPEnv, "TSP"

    @torch.no_grad()
    def beamsearch_tour_nodes_shortest(self, beam_size, batch_size, num_nodes, dtypeFloat=None, dtypeLong=None,
                                       probs_type="raw", random_start=False, start=None):
        """The reference's loop, object for object (TSP/TSPTester.py:145-199)."""
        env, model = self.env, self.model
        beamsearch = Beamsearch(beam_size, env.batch_size * env.pomo_size, num_nodes, dtypeFloat, dtypeLong, probs_type,
                                random_start, device=self.device)
        current_step = 0
        env.reset("test")
        state, reward, done = env.pre_step()
        model.pre_forward(env.dis_matrix, env.batch_size)
        knn = self.tester_params["knn"]
        while not done:
            if current_step == 0:
                idx = self._start_nodes() if start is None else start
                selected = idx.repeat(env.batch_size).repeat_interleave(env.beam_size).to(torch.int64)
            else:
                _, trans_probs, _ = model(state, env.selected_node_list, None, knn, current_step)
                env.selected_node_list = beamsearch.advance(
                    torch.log(trans_probs.view(env.batch_size * env.pomo_size, beam_size, -1)), env, knn, current_step)
                selected = beamsearch.next_nodes[-1].view(-1)
            current_step += 1
            state, reward, done = env.step(selected)
        shortest, index = self._summarise(reward, beam_size)
        self.last_tours = env.selected_node_list
        self.last_lengths = reward
        self.shortest_tours = self._best_of_group(self.last_tours, index)
        return 0, 0, 0, shortest.mean().item()

    @torch.no_grad()
    def fast_rollout(self, beam_size, start=None):
        """Same result through the device-resident path: no dense [R,N] probabilities, beam expansion
        over beam*k candidates, one launch sequence per step."""
        env, model = self.env, self.model
        env.reset("test")
        eng = env.engine
        idx = self._start_nodes() if start is None else start
        sel = idx.repeat(env.batch_size).repeat_interleave(env.beam_size).to(device=self.device, dtype=torch.int32)
        eng.set_knn(self.tester_params["knn"])
        policy = model.packed(self.device)
        if beam_size == 1:
            length = eng.greedy_rollout(policy, sel)
        else:
            length = eng.beam_rollout(policy, sel, beam_size)
        env.selected_count = eng.t
        if getattr(env, "_orig_xy", None) is not None:             # aug_test: lengths on the original coordinates
            length = env._get_travel_distance()
        shortest, index = self._summarise(length, beam_size)
        self.last_tours = env.selected_node_list
        self.last_lengths = length
        self.shortest_tours = self._best_of_group(self.last_tours, index)
        return shortest

    def run(self):
        tp = self.tester_params
        batch_size, beam_size = tp["test_batch_size"], tp["beam_size"]
        problem_size, episodes = tp["problem_size"], tp["test_episodes"]
        self.env.load_data(tp["test_episodes_all"], problem_size)
        have_base = len(self.env.baseline_len) > 0
        score_sum = gap_sum = 0.0
        count = 0
        for episode, nb, start in self._rank_batches(episodes, batch_size, problem_size):
            self.env.load_problems(episode, nb, problem_size, beam_size)
            self.env.start_idx, self.env.step_size = 0, nb
            if tp.get("reference_protocol"):
                self.beamsearch_tour_nodes_shortest(beam_size, nb, problem_size, start=start)
                shortest, _ = self._summarise(self.last_lengths, beam_size)
            else:
                shortest = self.fast_rollout(beam_size, start=start)
            score_sum += float(shortest.double().sum())
            if have_base:
                base = self.env.baseline_len[episode:episode + nb].to(shortest.device)
                gap_sum += float(((shortest - base) / base).double().sum())
            count += nb
        score_sum, gap_sum, count = D.reduce_stats(score_sum, gap_sum, float(count))
        self.score = score_sum / max(count, 1.0)
        avg_gap = gap_sum / max(count, 1.0) if have_base else float("nan")
        self.logger.info("avg_score {:.4f} gap {:.4f}".format(self.score, avg_gap))
        return avg_gap


class CVRPTester(_TesterBase):
    Model, Env, problem = CVRPModel, CVRPEnv, "CVRP"

    @torch.no_grad()
    def beamsearch_tour_nodes_shortest(self, beam_size, batch_size, num_nodes, dtypeFloat=None, dtypeLong=None,
                                       probs_type="raw", random_start=False, start=None):
        """CVRP/CVRPTester.py:147-227 ('pomo_test'); num_nodes = problem_size + 1."""
        env, model = self.env, self.model
        env.beam_size = beam_size
        beamsearch = Beamsearch(beam_size, env.batch_size * env.pomo_size, num_nodes, dtypeFloat, dtypeLong, probs_type,
                                random_start, "CVRP", device=self.device)
        current_step = 0
        env.reset("test")
        state, reward, done = env.pre_step()
        model.pre_forward(env.dis_matrix, env.batch_size)
        knn, depot_knn = self.tester_params["knn"], self.tester_params["depot_knn"]
        while not done:
            if current_step == 0:
                idx = self._start_nodes() if start is None else start
                selected = idx.repeat(env.batch_size).repeat_interleave(env.beam_size).to(torch.int64) + 2 + env.problem_size
            else:
                _, trans_probs, _ = model(state, env.selected_node_list, None, None, knn, current_step, depot_knn,
                                          env.step_state.selected_flag)
                env.selected_node_list = beamsearch.advance(
                    torch.log(trans_probs.view(env.batch_size * env.pomo_size, beam_size, -1)), env, knn, current_step)
                selected = beamsearch.next_nodes[-1].view(-1)
            current_step += 1
            state, reward, done = env.step(selected)
        shortest, index = self._summarise(reward, beam_size)
        self.last_tours = env.selected_node_list
        self.last_flags = env.step_state.selected_flag
        self.last_lengths = reward
        self.shortest_tours = self._best_of_group(self.last_tours, index)
        self.shortest_tours_flag = self._best_of_group(self.last_flags, index)
        return shortest.mean().item(), shortest.mean().item(), 0.0

    @torch.no_grad()
    def fast_rollout(self, beam_size, start=None):
        env, model = self.env, self.model
        env.beam_size = beam_size
        env.reset("test")
        eng = env.engine
        idx = self._start_nodes() if start is None else start
        sel = idx.repeat(env.batch_size).repeat_interleave(env.beam_size).to(device=self.device, dtype=torch.int32)
        sel = sel + 2 + env.problem_size
        eng.set_knn(self.tester_params["knn"], self.tester_params["depot_knn"])
        policy = model.packed(self.device)
        length = eng.greedy_rollout(policy, sel) if beam_size == 1 else eng.beam_rollout(policy, sel, beam_size)
        env.selected_count = eng.t
        if getattr(env, "_orig_xy", None) is not None:             # aug_test: lengths on the original coordinates
            length = env.cal_length()
        shortest, index = self._summarise(length, beam_size)
        self.last_tours = env.selected_node_list
        self.last_flags = env.step_state.selected_flag
        self.last_lengths = length
        self.shortest_tours = self._best_of_group(self.last_tours, index)
        self.shortest_tours_flag = self._best_of_group(self.last_flags, index)
        return shortest

    def run(self):
        tp = self.tester_params
        batch_size, beam_size = tp["test_batch_size"], tp["beam_size"]
        problem_size, episodes = tp["problem_size"], tp["test_episodes"]
        self.env.problem_size = problem_size
        self.env.load_data(tp["test_episodes_all"], problem_size)
        score_sum = 0.0
        count = 0
        for episode, nb, start in self._rank_batches(episodes, batch_size, problem_size):
            self.env.load_problems(episode, nb, problem_size)
            self.env.start_idx, self.env.step_size = 0, nb
            if tp.get("reference_protocol"):
                self.beamsearch_tour_nodes_shortest(beam_size, nb, problem_size + 1, start=start)
                shortest, _ = self._summarise(self.last_lengths, beam_size)
            else:
                shortest = self.fast_rollout(beam_size, start=start)
            score_sum += float(shortest.double().sum())
            count += nb
        score_sum, _, count = D.reduce_stats(score_sum, 0.0, float(count))
        self.score = score_sum / max(count, 1.0)
        self.logger.info("avg_score {:.4f}".format(self.score))
        return self.score
