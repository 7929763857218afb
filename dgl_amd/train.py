"""Teacher-forced training step of the reference's trainers on the device
(CVRP/CVRPTrainer.py:232-282, TSP/TSPTrainer.py): forward in mode 'train' -> probability of the
teacher's action, loss = -log(prob).mean(), backward, gradient all-reduce over the ranks of a
torch.distributed job, Adam.  The arithmetic runs in libdgl_amd (dgl_policy_general,
dgl_adam_step); this module owns the flat parameter / gradient / moment vectors.

Only the tensors the forward pass uses take part: ``decoder.embedding_last_node_pos`` and
``decoder.k_1`` exist in the checkpoint but never receive a gradient, and torch.optim.Adam skips
parameters without one (no weight decay either), so they are carried through unchanged.
"""
from __future__ import annotations

import ctypes
from typing import Dict, Optional

import torch

from . import _lib
from ._lib import PROBLEM_TSP, check, ptr, stream_ptr
from .distributed import all_reduce_gradients
from .engine import general_workspace, weights_struct

UNUSED = ("decoder.embedding_last_node_pos.", "decoder.k_1.")


class PolicyTrainer:
    def __init__(self, state_dict: Dict[str, torch.Tensor], problem: int, device: torch.device, lr: float = 1e-4,
                 weight_decay: float = 1e-6, betas=(0.9, 0.999), eps: float = 1e-8):
        self.lib = _lib.load()
        self.problem, self.device = problem, device
        self.lr, self.weight_decay, self.betas, self.eps = lr, weight_decay, betas, eps
        self.keys = [k for k in state_dict if not k.startswith(UNUSED)]
        self.unused = {k: v.detach().clone() for k, v in state_dict.items() if k.startswith(UNUSED)}
        self.order = list(state_dict.keys())
        sizes = [state_dict[k].numel() for k in self.keys]
        self.numel = sum(sizes)
        self.params = torch.empty(self.numel, dtype=torch.float32, device=device)
        self.grads = torch.zeros_like(self.params)
        self.exp_avg = torch.zeros_like(self.params)
        self.exp_avg_sq = torch.zeros_like(self.params)
        self.views, self.grad_views = {}, {}
        off = 0
        for k, n in zip(self.keys, sizes):
            shape = tuple(state_dict[k].shape)
            self.views[k] = self.params[off:off + n].view(shape)
            self.grad_views[k] = self.grads[off:off + n].view(shape)
            self.views[k].copy_(state_dict[k].to(device=device, dtype=torch.float32))
            off += n
        self.w = weights_struct(self.views, problem)
        self.g = weights_struct(self.grad_views, problem)
        self.step_count = 0

    # ---- checkpoint interchange -----------------------------------------------------------
    def state_dict(self) -> Dict[str, torch.Tensor]:
        out = {}
        for k in self.order:
            out[k] = (self.views[k] if k in self.views else self.unused[k]).detach().clone()
        return out

    def load_state_dict(self, state_dict: Dict[str, torch.Tensor]) -> None:
        for k in self.order:
            v = state_dict[k].to(device=self.device, dtype=torch.float32)
            (self.views[k] if k in self.views else self.unused[k]).copy_(v)

    def optimizer_state_dict(self) -> dict:
        """torch.optim.Adam's state_dict layout over the parameters in checkpoint order (the trainers save
        it as 'optimizer_state_dict', CVRP/CVRPTrainer.py:128-135): parameters that never received a
        gradient have no state entry."""
        state, off = {}, 0
        index = {k: i for i, k in enumerate(self.order)}
        for k in self.keys:
            n = self.views[k].numel()
            if self.step_count > 0:
                state[index[k]] = {"step": torch.tensor(float(self.step_count)),
                                   "exp_avg": self.exp_avg[off:off + n].view_as(self.views[k]).clone(),
                                   "exp_avg_sq": self.exp_avg_sq[off:off + n].view_as(self.views[k]).clone()}
            off += n
        group = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.weight_decay,
                 "amsgrad": False, "maximize": False, "foreach": None, "capturable": False, "differentiable": False,
                 "fused": None, "decoupled_weight_decay": False, "params": list(range(len(self.order)))}
        return {"state": state, "param_groups": [group]}

    def load_optimizer_state_dict(self, sd: dict) -> None:
        group = sd["param_groups"][0]
        self.lr, self.betas, self.eps = group["lr"], tuple(group["betas"]), group["eps"]
        self.weight_decay = group["weight_decay"]
        index = {k: i for i, k in enumerate(self.order)}
        off, steps = 0, set()
        for k in self.keys:
            n = self.views[k].numel()
            st = sd["state"].get(index[k])
            if st is not None:
                self.exp_avg[off:off + n].copy_(st["exp_avg"].reshape(-1))
                self.exp_avg_sq[off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
                steps.add(int(st["step"]))
            off += n
        if len(steps) > 1:
            raise ValueError("parameters with different Adam step counts are not supported")
        self.step_count = steps.pop() if steps else 0

    def gradient_dict(self) -> Dict[str, torch.Tensor]:
        return {k: v.detach().clone() for k, v in self.grad_views.items()}

    # ---- one decode step ------------------------------------------------------------------
    def zero_grad(self) -> None:
        self.grads.zero_()

    def forward_backward(self, engine, teacher: torch.Tensor, global_rows: Optional[int] = None) -> torch.Tensor:
        """model(state, ..., mode='train') + loss.backward() for the engine's current tile: returns the
        teacher's probability per row [R]; gradients of -log(prob).mean() (mean over ``global_rows``,
        default the engine's rows times the world size) are added to ``self.grads``."""
        io = engine.policy_io(want_logits=False, want_prob=False, want_selected=False)
        if global_rows is None:
            global_rows = io.R * world_size()
        ws = general_workspace(engine, io.R, io.n)
        prob = torch.empty(io.R, dtype=torch.float32, device=self.device)
        teacher = teacher.to(device=self.device, dtype=torch.int32).contiguous()
        check(self.lib.dgl_policy_general(ctypes.byref(self.w), ctypes.byref(self.g), ctypes.byref(io), ptr(teacher),
                                          1.0 / float(global_rows), ptr(ws), ptr(prob), stream_ptr()),
              "dgl_policy_general")
        return prob

    def optimizer_step(self) -> None:
        """Gradient all-reduce (sum over ranks; the per-rank gradients already carry 1/global_rows) and Adam."""
        all_reduce_gradients(self.grads)
        self.step_count += 1
        b1, b2 = self.betas
        check(self.lib.dgl_adam_step(ptr(self.params), ptr(self.grads), ptr(self.exp_avg), ptr(self.exp_avg_sq),
                                     self.numel, self.step_count, self.lr, b1, b2, self.eps, self.weight_decay, 1.0,
                                     stream_ptr()), "dgl_adam_step")


class _TeacherProbability(torch.autograd.Function):
    """The reference's ``model(...)`` in mode 'train' as one differentiable node: forward returns the teacher's
    probability per row [R, 1] (TSP/TSPModel.py:206-217, CVRP/CVRPModel.py:311-345), backward hands the
    parameters' gradients to autograd, so that the reference's own loop
        loss = -prob.type(torch.float64).log().mean(); model.zero_grad(); loss.backward(); optimizer.step()
    (CVRP/CVRPTrainer.py:268-272) runs unchanged on a torch optimizer over ``model.parameters()``.

    The device kernel evaluates forward and backward in one call and differentiates exactly that loss
    (-mean log p over the rows); backward() checks that the incoming gradient is c * d(-mean log p)/dp for one
    constant c (c = 1 for the reference's loss) and scales by it; any other loss is refused."""

    @staticmethod
    def forward(ctx, engine, problem, teacher, keys, *params):
        lib = _lib.load()
        dev = engine.device
        for k, p in zip(keys, params):
            if p.device != dev or p.dtype != torch.float32 or not p.is_contiguous():
                raise ValueError(f"parameter {k} must be a contiguous float32 tensor on {dev} (model.to(device))")
        flat = torch.zeros(sum(p.numel() for p in params), dtype=torch.float32, device=dev)
        grads, off = {}, 0
        for k, p in zip(keys, params):
            grads[k] = flat[off:off + p.numel()].view(p.shape)
            off += p.numel()
        w = weights_struct({k: p.detach() for k, p in zip(keys, params)}, problem)
        g = weights_struct(grads, problem)
        io = engine.policy_io(want_logits=False, want_prob=False, want_selected=False)
        ws = general_workspace(engine, io.R, io.n)
        prob = torch.empty(io.R, dtype=torch.float32, device=dev)
        teacher = teacher.to(device=dev, dtype=torch.int32).contiguous()
        with torch.cuda.device(dev):
            check(lib.dgl_policy_general(ctypes.byref(w), ctypes.byref(g), ctypes.byref(io), ptr(teacher), 1.0 / io.R,
                                         ptr(ws), ptr(prob), stream_ptr()), "dgl_policy_general")
        ctx.grads = [grads[k] for k in keys]
        ctx.rows = io.R
        ctx.save_for_backward(prob)
        return prob.view(-1, 1)

    @staticmethod
    def backward(ctx, grad_out):
        (prob,) = ctx.saved_tensors
        # the kernel's gradients are those of sum_r (1/R)(-log p_r): they answer d loss / d p_r = -1 / (R p_r)
        c = grad_out.reshape(-1).double() * prob.double() * (-float(ctx.rows))
        # one reduction, one trip to the host: [mean, largest deviation from it]
        mean = c.mean()
        scale, spread = torch.stack((mean, (c - mean).abs().max())).tolist()
        if not spread <= 1e-4 * abs(scale):
            raise NotImplementedError("the device backward pass differentiates -log(prob).mean() (times a constant) only")
        # grad_out went through float32 (1/p rounded, then times p): a constant within 1e-5 of 1 IS the reference's loss,
        # and the kernel's gradients are handed over as they are, bit-equal to PolicyTrainer.forward_backward
        out = ctx.grads if abs(scale - 1.0) < 1e-5 else [g * scale for g in ctx.grads]
        return (None, None, None, None, *out)


def teacher_probability(model: torch.nn.Module, engine, problem: int, teacher: torch.Tensor) -> torch.Tensor:
    """Probability of the teacher's action per row, [R, 1], differentiable with respect to ``model``'s parameters
    (the two tensors the forward pass never uses get no gradient, as in the reference)."""
    named = [(k, p) for k, p in model.named_parameters() if not k.startswith(UNUSED)]
    keys = [k for k, _ in named]
    return _TeacherProbability.apply(engine, problem, teacher, keys, *[p for _, p in named])


def world_size() -> int:
    d = torch.distributed
    return d.get_world_size() if d.is_available() and d.is_initialized() else 1


def train_one_batch_cvrp(trainer: PolicyTrainer, engine, solution: torch.Tensor, solution_flag: torch.Tensor) -> float:
    """CVRPTrainer._train_one_batch (CVRP/CVRPTrainer.py:232-282) on a reset engine: ``solution`` [R][N]
    customer ids (1..N) in visiting order, ``solution_flag`` [R][N] 1 where the move goes via the depot.
    One Adam update per decode step.  Returns -log(prob_list).mean() over all steps after the first."""
    N = engine.N
    dev = engine.device
    act = (solution + solution_flag * (N + 1)).to(device=dev, dtype=torch.int32)
    engine.reset()
    engine.step(act[:, 0].contiguous())
    logs = []
    for t in range(1, N):
        trainer.zero_grad()
        prob = trainer.forward_backward(engine, act[:, t].contiguous())
        trainer.optimizer_step()
        logs.append(prob.double().log())
        engine.step(act[:, t].contiguous())
    lp = torch.stack(logs, 1)
    trainer.last_log_probs = lp
    # the reference's prob_list also holds the ones of the first step (log 1 = 0)
    return float(-(lp.sum() / (lp.numel() + lp.shape[0])))


def train_one_batch_tsp(trainer: PolicyTrainer, engine, solution: torch.Tensor) -> float:
    """TSPTrainer._train_one_batch (TSP/TSPTrainer.py:216-256): ``solution`` [R][N] node ids in visiting
    order; the first node is given, every later one is a teacher-forced decision with its own Adam update."""
    N = engine.N
    act = solution.to(device=engine.device, dtype=torch.int32)
    engine.reset()
    engine.step(act[:, 0].contiguous())
    logs = []
    for t in range(1, N):
        trainer.zero_grad()
        prob = trainer.forward_backward(engine, act[:, t].contiguous())
        trainer.optimizer_step()
        logs.append(prob.double().log())
        engine.step(act[:, t].contiguous())
    lp = torch.stack(logs, 1)
    trainer.last_log_probs = lp
    return float(-(lp.sum() / (lp.numel() + lp.shape[0])))
