"""CVRP host side with the reference's interface: CVRPEnv (CVRP/CVRPEnv.py), CVRPModel
(CVRP/CVRPModel.py).  See dgl_amd.tsp for what differs from the reference's objects."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch
import torch.nn as nn

from . import _lib
from ._lib import PROBLEM_CVRP, check, ptr, stream_ptr
from .augment import mix_augment
from .engine import CvrpEngine, PackedPolicy, forward_range_guarded
from .formats import read_cvrp_instances_from_file
from .nn_layout import DecoderParams, EncoderParams

SHIPPED_INFO = [True, True, True, False, True, True, False, False, False, False, False, False, False]


def get_random_problems(batch_size, problem_size):
    """CVRP/CVRPEnv.py:30-51 (demand scaler 30 / 40 for 20 / 50 customers, 500 otherwise)."""
    depot_xy = torch.rand(size=(batch_size, 1, 2))
    node_xy = torch.rand(size=(batch_size, problem_size, 2))
    scaler = {20: 30, 50: 40}.get(problem_size, 500)
    node_demand = torch.randint(1, 200, size=(batch_size, problem_size)) / float(scaler)
    return depot_xy, node_xy, node_demand


@dataclass
class Reset_State:
    problems: torch.Tensor


class Step_State:
    """CVRP/CVRPEnv.py:170-180, backed by the device engine."""

    def __init__(self, engine: CvrpEngine):
        self.engine = engine

    @property
    def data(self) -> torch.Tensor:
        e = self.engine
        if e.t == 0:
            return torch.cat((e.xy, e.demand[:, :, None]), 2).repeat_interleave(e.rpi, 0)
        return e.data()

    @property
    def avg_unselect_distence(self):
        return self.engine.state["avg"]

    @property
    def std_dev_unselect_distence(self):
        return self.engine.state["sd"]

    @property
    def capacity(self):
        return self.engine.state["capacity"]

    @property
    def ninf_mask(self) -> torch.Tensor:
        """0 / -inf feasibility mask [R,N+1] (CVRP/CVRPEnv.py:633-637), derived on access."""
        e = self.engine
        dem = e.demand.repeat_interleave(e.rpi, 0)
        mask = torch.zeros_like(dem)
        mask[(e.state["capacity"][:, None] + 0.000001) < dem] = float("-inf")
        return mask

    @property
    def selected_flag(self) -> torch.Tensor:
        e = self.engine
        return e.state["flags"][:, :e.t].to(torch.int64)


class CVRPEnv:
    def __init__(self, **env_params):
        self.env_params = env_params
        self.problem_size = None
        self.mode = env_params["mode"]
        if self.mode == "test":
            if env_params["test_mode"] == "pomo_test":
                self.pomo_size = env_params["aug_size"]
            elif env_params["test_mode"] == "aug_test":
                self.aug_size = env_params["aug_size"]
                self.pomo_size = 1
        else:
            self.pomo_size = env_params["pomo_size"]
        info = list(env_params.get("append_information", SHIPPED_INFO))
        if info[:len(SHIPPED_INFO)] != SHIPPED_INFO:
            raise NotImplementedError("only the shipped feature set (append_information 0,1,2,4,5) is implemented")
        self.append_information = info
        self.device = torch.device(env_params.get("device", "cuda"))
        self.knn = env_params.get("knn", 15)
        self.depot_knn = env_params.get("depot_knn", 15)
        self.batch_size = None
        self.problems = None
        self.step_state = None
        self.engine: Optional[CvrpEngine] = None
        self.beam_size = 1
        self.start_idx = 0
        self.step_size = None
        self.dis_matrix = None
        self.solution = None
        self.solution_flag = None
        self.solution_len = None
        self.selected_count = None

    def load_data(self, test_episode, problem_size):
        self.raw_depot_node_xy, self.raw_node_xy, self.raw_node_demand = \
            read_cvrp_instances_from_file(self.env_params["data_path"])

    def load_problems(self, episode, batch_size, problem_size):
        """CVRP/CVRPEnv.py:222-264."""
        self.episode = episode
        self.batch_size = batch_size
        self.problem_size = problem_size
        if self.mode == "train":
            depot_xy, node_xy, node_demand = get_random_problems(batch_size, problem_size)
            self.depot_node_xy = torch.cat((depot_xy, node_xy), dim=1)
        else:
            depot_xy = self.raw_depot_node_xy[episode:episode + batch_size]
            node_xy = self.raw_node_xy[episode:episode + batch_size]
            node_demand = self.raw_node_demand[episode:episode + batch_size]
            self.depot_node_xy = torch.cat((depot_xy[:, None, :], node_xy), dim=1)
        self.depot_node_demand = torch.cat((torch.zeros(size=(batch_size, 1)), node_demand), dim=1)
        if self._aug_test():
            self.depot_node_xy = self.depot_node_xy.repeat_interleave(self.aug_size, dim=0)
            self.depot_node_demand = self.depot_node_demand.repeat_interleave(self.aug_size, dim=0)
            self.batch_size = batch_size * self.aug_size
        self.problems = torch.cat((self.depot_node_xy, self.depot_node_demand[:, :, None]), dim=2)
        self.solution = None
        self.solution_len = None

    # ---- self-improvement labels (training) ------------------------------------------------
    def _device_instances(self, sl=slice(None)):
        xy = self.depot_node_xy[sl].to(self.device, torch.float32).contiguous()
        dem = self.depot_node_demand[sl].to(self.device, torch.float32).contiguous()
        return xy, dem

    def _label_length(self, xy, dem, nodes, flags) -> torch.Tensor:
        """cal_length (CVRP/CVRPEnv.py:951-993) of labels held outside an engine."""
        eng = CvrpEngine(xy, dem, 1, self.knn, self.depot_knn)
        eng.state["nodes"].copy_(nodes.to(torch.int32))
        eng.state["flags"].copy_(flags.to(torch.int8))
        return eng.route_length().clone()

    def greedy_search(self):
        """Nearest-neighbour labels for the whole batch (CVRP/CVRPEnv.py:340-409)."""
        lib = _lib.load()
        xy, dem = self._device_instances()
        B, N = self.batch_size, self.problem_size
        nodes = torch.empty(B, N, dtype=torch.int32, device=self.device)
        flags = torch.empty(B, N, dtype=torch.int8, device=self.device)
        check(lib.dgl_cvrp_nearest_neighbour(ptr(xy), ptr(dem), B, N, ptr(nodes), ptr(flags), stream_ptr()),
              "dgl_cvrp_nearest_neighbour")
        self.solution = nodes.to(torch.int64)
        self.solution_flag = flags.to(torch.int64)
        self.solution_len = self._label_length(xy, dem, nodes, flags)
        return self.solution_len.mean().item()

    def random_replace(self, random_index=None, replacement=None):
        """Fresh instances for ``replace_size`` random slots; their labels are outdated until the next
        rollout beats the infinite length (CVRP/CVRPEnv.py:298-323).  Same host RNG calls as the reference
        unless the slots and their new instances (``replacement`` [r][N+1][3]) are given."""
        if random_index is None:
            random_index = torch.randperm(self.batch_size)[:self.replace_size]
            depot_xy, node_xy, node_demand = get_random_problems(self.replace_size, self.problem_size)
            new_xy = torch.cat((depot_xy, node_xy), dim=1)
            new_demand = torch.cat((torch.zeros(size=(self.replace_size, 1)), node_demand), dim=1)
        else:
            new_xy, new_demand = replacement[:, :, :2], replacement[:, :, 2]
        self.depot_node_xy[random_index] = new_xy
        self.depot_node_demand[random_index] = new_demand
        self.problems = torch.cat((self.depot_node_xy, self.depot_node_demand[:, :, None]), dim=2)
        self.solution_len[random_index.to(self.device)] = float("inf")
        return random_index

    def draw_label_augmentation(self):
        """The reference's draws for data_augmentation, in its order on the host generator: per instance
        one torch.rand(1) per sub-route but the last, then randint(0, S) (CVRP/CVRPEnv.py:329-335)."""
        counts = self.solution_flag.sum(dim=1).tolist()
        B, N = self.solution_flag.shape
        flip = torch.zeros(B, N, dtype=torch.uint8)
        shift = torch.zeros(B, dtype=torch.int32)
        for i, S in enumerate(counts):
            for j in range(1, int(S)):
                if torch.rand(1) > 0.5:
                    flip[i, j] = 1
            shift[i] = torch.randint(0, int(S), (1,))[0].item()
        return flip, shift

    def data_augmentation(self, flip=None, shift=None):
        """Sub-route reversals and rotation of every label (CVRP/CVRPEnv.py:325-338); ``flip`` [B][N] and
        ``shift`` [B] default to draw_label_augmentation()."""
        if flip is None or shift is None:
            flip, shift = self.draw_label_augmentation()
        lib = _lib.load()
        B, N = self.solution.shape
        nodes = self.solution.to(torch.int32).contiguous()
        flags = self.solution_flag.to(torch.int8).contiguous()
        flip = flip.to(device=self.device, dtype=torch.uint8).contiguous()
        shift = shift.to(device=self.device, dtype=torch.int32).contiguous()
        assert flip.shape == (B, N) and shift.shape == (B,) and int(shift.min()) >= 0
        out_n, out_f = torch.empty_like(nodes), torch.empty_like(flags)
        check(lib.dgl_cvrp_solution_augment(ptr(nodes), ptr(flags), B, N, ptr(flip), ptr(shift), ptr(out_n), ptr(out_f),
                                            stream_ptr()), "dgl_cvrp_solution_augment")
        self.solution = out_n.to(torch.int64)
        self.solution_flag = out_f.to(torch.int64)

    def _aug_test(self) -> bool:
        return self.mode == "test" and self.env_params.get("test_mode") == "aug_test"

    def reset(self, mode):
        if mode not in ("test", "valid", "train"):
            raise ValueError(f"unknown mode {mode!r}")
        self.mode = mode
        self.selected_count = 0
        self._orig_xy = None
        if mode == "train":
            # CVRP/CVRPEnv.py:508-524: whole batch, one row per instance, mix-augmented coordinates,
            # distance matrix over (x, y, demand)
            xy, dem = self._device_instances()
            self.engine = CvrpEngine(mix_augment(xy), dem, 1, self.knn, self.depot_knn, demand_in_distance=True)
        else:
            nb = self.batch_size if mode == "test" else self.step_size
            xy, dem = self._device_instances(slice(self.start_idx, self.start_idx + nb))
            if self._aug_test():                      # CVRP/CVRPEnv.py:462-467: decide on augmented coordinates
                self._orig_xy = xy
                xy = mix_augment(xy)
            self.engine = CvrpEngine(xy, dem, self.pomo_size * self.beam_size, self.knn, self.depot_knn)
        self.engine.reset()
        self.step_state = Step_State(self.engine)
        self._permuted_at = -1
        return Reset_State(self.problems), None, False

    def pre_step(self):
        return self.step_state, None, False

    @property
    def selected_node_list(self) -> torch.Tensor:
        e = self.engine
        return e.state["nodes"][:, :e.t].to(torch.int64)

    @selected_node_list.setter
    def selected_node_list(self, value):
        pass        # see TSPEnv.selected_node_list

    def perm_attr(self, attr_name, prev_k):
        """CVRP/CVRPEnv.py:536-581; the compact state is re-indexed once per beam step."""
        e = self.engine
        if self._permuted_at == e.t or self.beam_size == 1:     # beam 1: the back-pointer is always 0
            return
        e.permute(prev_k.to(torch.int32).contiguous(), self.beam_size)
        self._permuted_at = e.t

    def step(self, selected):
        """``selected`` holds action ids; like the reference (CVRP/CVRPEnv.py:619) the caller's tensor
        is rewritten to node ids for via-depot moves."""
        e = self.engine
        act = selected.to(device=self.device, dtype=torch.int32).contiguous()
        via = selected > self.problem_size
        selected[via] = selected[via] - (self.problem_size + 1)
        self.selected_count += 1
        e.step(act)
        done = self.selected_count == self.problem_size
        reward = self.cal_length() if done else (None, None)
        return self.step_state, reward, done

    def cal_length(self, problems=None, order_node=None, order_flag=None):
        """Route length of the engine's routes; in aug_test on the original coordinates (CVRP/CVRPEnv.py:864-865)."""
        return self.engine.route_length(self._orig_xy).clone()


class CVRPModel(nn.Module):
    def __init__(self, **model_params):
        super().__init__()
        self.model_params = model_params
        self.mode = model_params["mode"]
        if (model_params["embedding_dim"], model_params["head_num"], model_params["qkv_dim"],
                model_params["ff_hidden_dim"], model_params["decoder_layer_num"]) != (128, 8, 16, 512, 3):
            raise NotImplementedError("kernels are specialised for embedding 128, 8 heads x 16, FF 512, 3 decoder layers")
        self.encoder = EncoderParams(9)
        self.decoder = DecoderParams("CVRP")
        self._packed = None
        self._packed_key = None

    def pre_forward(self, dis_matrix, batch_size):
        self.dis_matrix = dis_matrix
        self.batch_size = batch_size

    def packed(self, device) -> PackedPolicy:
        key = (str(device),) + tuple((p.data_ptr(), p._version) for p in self.parameters())
        if self._packed is None or key != self._packed_key:
            self._packed = PackedPolicy(self.state_dict(), PROBLEM_CVRP, device)
            self._packed_key = key
        return self._packed

    def forward(self, state, selected_node_list, solution, solution_flag, knn, current_step, depot_knn_nums,
                selected_flag=None):
        eng: CvrpEngine = state.engine
        eng.set_knn(knn, depot_knn_nums)
        eng.ensure_neighbourhood()
        if self.mode == "train":                    # CVRP/CVRPModel.py:311-345
            from .train import teacher_probability
            # like the reference, the teacher's entry of ``solution`` (a view) is rewritten to the action id in
            # place for a move via the depot; CVRPEnv.step rewrites it back (CVRP/CVRPModel.py:329, CVRPEnv.py:619)
            selected_teacher = solution[:, current_step]
            is_via_depot = solution_flag[:, current_step] == 1
            selected_teacher[is_via_depot] += eng.N + 1
            return selected_teacher, teacher_probability(self, eng, PROBLEM_CVRP, selected_teacher), 1
        forward_range_guarded(eng, self.packed(eng.device))
        return None, eng.dense_probs(), 1
