"""TSP host side with the reference's interface: TSPEnv (TSP/TSPEnv.py), TSPModel
(TSP/TSPModel.py).  Same constructor dictionaries, method names, argument meaning and
checkpoint keys; the bodies call the HIP kernels through the C ABI (dgl_amd.engine).

Differences a caller can observe:
  * state lives on the GPU in compact form; ``step_state.data`` ([R,N,7]) is materialised on
    access (dgl_tsp_state_data) instead of being updated in place;
  * ``env.dis_matrix`` is never built (distances are recomputed from coordinates with the
    arithmetic of torch.cdist's matrix-product path), ``model.pre_forward`` accepts ``None``;
  * only the shipped feature switches (append_information 0,1,2,4,5) are implemented;
  * index tensors are int64 at this boundary (as in the reference) and int32 on the device.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch
import torch.nn as nn

from . import _lib
from ._lib import PROBLEM_TSP, check, ptr, stream_ptr
from .augment import mix_augment
from .engine import PackedPolicy, TspEngine, forward_range_guarded
from .formats import read_baseline_lengths, read_tsp_instances
from .nn_layout import DecoderParams, EncoderParams

SHIPPED_INFO = [True, True, True, False, True, True, False, False, False, False, False]


@dataclass
class Reset_State:
    problems: torch.Tensor


class Step_State:
    """TSP/TSPEnv.py:39-43, backed by the device engine."""

    def __init__(self, engine: TspEngine):
        self.engine = engine

    @property
    def data(self) -> torch.Tensor:
        e = self.engine
        if e.t == 0:
            return e.xy.repeat_interleave(e.rpi, 0)
        return e.data()

    @property
    def avg_unselect_distence(self) -> torch.Tensor:
        return self.engine.state["avg"]

    @property
    def std_dev_unselect_distence(self) -> torch.Tensor:
        return self.engine.state["sd"]


def get_random_problems(batch_size, problem_size):
    return torch.rand(size=(batch_size, problem_size, 2))       # TSP/TSPEnv.py:29-32


class TSPEnv:
    def __init__(self, **env_params):
        self.env_params = env_params
        self.problem_size = None
        self.data_path = env_params.get("data_path", "")
        self.mode = env_params["mode"]
        if self.mode == "test":
            self.baseline_path = env_params.get("baseline_path", "")
            if env_params["test_mode"] == "pomo_test":
                self.pomo_size = env_params["aug_size"]
            elif env_params["test_mode"] == "aug_test":        # TSP/TSPEnv.py:60-62
                self.aug_size = env_params["aug_size"]
                self.pomo_size = 1
        else:
            self.pomo_size = env_params["pomo_size"]
        info = list(env_params.get("append_information", SHIPPED_INFO))
        if info[:len(SHIPPED_INFO)] != SHIPPED_INFO:
            raise NotImplementedError("only the shipped feature set (append_information 0,1,2,4,5) is implemented")
        self.append_information = info
        self.device = torch.device(env_params.get("device", "cuda"))
        self.knn = env_params.get("knn", 30)       # neighbourhood built inside step(); the model may override it
        self.batch_size = None
        self.problems = None
        self.raw_data_nodes = []
        self.baseline_len = []
        self.selected_count = None
        self.step_state = None
        self.engine: Optional[TspEngine] = None
        self.beam_size = 1
        self.start_idx = 0
        self.step_size = None
        self.dis_matrix = None
        self.solution = None
        self.solution_len = None

    # ---- data ---------------------------------------------------------------------
    def load_data(self, test_episode, problem_size):
        """Instance text file "x,y x,y ..." and optional baseline-length file (TSP/TSPEnv.py:82-106)."""
        self.raw_data_nodes = read_tsp_instances(self.data_path, test_episode, problem_size)
        self.baseline_len = []
        if self.baseline_path != "":
            self.baseline_len = read_baseline_lengths(self.baseline_path)

    def load_problems(self, episode, batch_size, problem_size, beam_size):
        self.episode = episode
        self.batch_size = batch_size
        self.problem_size = problem_size
        if self.mode == "train":
            self.problems = get_random_problems(batch_size, problem_size)
        else:
            self.problems = self.raw_data_nodes[episode:episode + batch_size]
            if self.env_params.get("test_mode") == "aug_test":      # TSP/TSPEnv.py:119-121
                self.problems = self.problems.repeat_interleave(self.aug_size, dim=0)
                self.batch_size = self.batch_size * self.aug_size
        self.beam_size = beam_size
        self.solution = None
        self.solution_len = None

    # ---- self-improvement labels (training) -------------------------------------------
    def greedy_search(self):
        """Nearest-neighbour labels from node 0 for the whole batch (TSP/TSPEnv.py:140-169)."""
        lib = _lib.load()
        xy = self.problems.to(self.device, torch.float32).contiguous()
        B, N = self.batch_size, self.problem_size
        sol = torch.empty(B, N, dtype=torch.int32, device=self.device)
        length = torch.empty(B, dtype=torch.float32, device=self.device)
        check(lib.dgl_tsp_nearest_neighbour(ptr(xy), B, N, ptr(sol), ptr(length), stream_ptr()),
              "dgl_tsp_nearest_neighbour")
        self.solution = sol.to(torch.int64)
        self.solution_len = length
        return self.solution_len.mean().item()

    def random_replace(self, random_index=None, replacement=None):
        """Fresh instances for ``replace_size`` random slots (TSP/TSPEnv.py:130-137); same host RNG calls as
        the reference unless the slots and their new instances (``replacement`` [r][N][2]) are given."""
        if random_index is None:
            random_index = torch.randperm(self.batch_size)[:self.replace_size]
            replacement = get_random_problems(self.replace_size, self.problem_size)
        self.problems[random_index] = replacement
        self.solution_len[random_index.to(self.device)] = float("inf")
        return random_index

    # ---- episode --------------------------------------------------------------------
    def reset(self, mode):
        if mode not in ("test", "valid", "train"):
            raise ValueError(f"unknown mode {mode!r}")
        self.mode = mode
        self.selected_count = 0
        self._orig_xy = None
        if mode == "train":                 # TSP/TSPEnv.py:255-266: whole batch, one row per instance, augmented
            xy = mix_augment(self.problems.to(self.device, torch.float32).contiguous())
            self.engine = TspEngine(xy, 1, self.knn)
        else:
            nb = self.batch_size if mode == "test" else self.step_size
            xy = self.problems[self.start_idx:self.start_idx + nb].to(self.device, torch.float32).contiguous()
            if mode == "test" and self.env_params.get("test_mode") == "aug_test":
                # run_aug("mix", ...) on the repeated instances (TSP/TSPEnv.py:222-224); rewards use the originals
                self._orig_xy = xy
                xy = mix_augment(xy)
            self.engine = TspEngine(xy, self.pomo_size * self.beam_size, self.knn)
        self.engine.reset()
        self.step_state = Step_State(self.engine)
        self._permuted_at = -1
        return Reset_State(self.problems), None, False

    def pre_step(self):
        return self.step_state, None, False

    @property
    def selected_node_list(self) -> torch.Tensor:
        e = self.engine
        return e.state["tour"][:, :e.t].to(torch.int64)

    @selected_node_list.setter
    def selected_node_list(self, value):
        # Beamsearch.advance returns the re-indexed list and the tester assigns it back
        # (TSP/TSPTester.py:174); the device copy was already permuted by perm_attr.
        pass

    def perm_attr(self, attr_name, prev_k):
        """TSP/TSPEnv.py:278-308.  The whole compact state is re-indexed once per beam step, by the
        first attribute asked for."""
        e = self.engine
        if self._permuted_at == e.t or self.beam_size == 1:     # beam 1: the back-pointer is always 0
            return
        e.permute(prev_k.to(torch.int32).contiguous(), self.beam_size)
        self._permuted_at = e.t

    def step(self, selected):
        e = self.engine
        self.selected_count += 1
        e.step(selected.to(device=self.device, dtype=torch.int32).contiguous())
        done = self.selected_count == self.problem_size
        reward = self._get_travel_distance() if done else (None, None)
        return self.step_state, reward, done

    def _get_travel_distance(self, instances=None):
        return self.engine.tour_length(self._orig_xy).clone()


class TSPModel(nn.Module):
    """Parameter container with the reference's state_dict keys; forward runs the HIP kernels."""

    def __init__(self, **model_params):
        super().__init__()
        self.model_params = model_params
        self.mode = model_params["mode"]
        if (model_params["embedding_dim"], model_params["head_num"], model_params["qkv_dim"],
                model_params["ff_hidden_dim"], model_params["decoder_layer_num"]) != (128, 8, 16, 512, 3):
            raise NotImplementedError("kernels are specialised for embedding 128, 8 heads x 16, FF 512, 3 decoder layers")
        self.encoder = EncoderParams(7)
        self.decoder = DecoderParams("TSP")
        self._packed = None
        self._packed_key = None

    def pre_forward(self, dis_matrix, batch_size):
        self.dis_matrix = dis_matrix
        self.batch_size = batch_size

    def packed(self, device) -> PackedPolicy:
        key = (str(device),) + tuple((p.data_ptr(), p._version) for p in self.parameters())
        if self._packed is None or key != self._packed_key:
            self._packed = PackedPolicy(self.state_dict(), PROBLEM_TSP, device)
            self._packed_key = key
        return self._packed

    def forward(self, state, selected_node_list, solution, knn, current_step):
        eng: TspEngine = state.engine
        eng.set_knn(knn)
        eng.ensure_neighbourhood()
        if self.mode == "train":                    # TSP/TSPModel.py:206-217
            from .train import teacher_probability
            selected_teacher = solution[:, current_step]
            return selected_teacher, teacher_probability(self, eng, PROBLEM_TSP, selected_teacher), 1
        forward_range_guarded(eng, self.packed(eng.device))
        return None, eng.dense_probs(), 1
