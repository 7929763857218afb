"""Device-side engines: thin owners of the HBM buffers that the C-ABI kernels work on.

PyTorch is used only to allocate device memory and to name the HIP stream; every
computation on the decode path is a call into libdgl_amd.so.  The reference-facing
classes (dgl_amd.tsp / dgl_amd.cvrp / dgl_amd.beamsearch) sit on top of these.
"""
from __future__ import annotations

import ctypes
import functools
import os
from typing import Dict, Optional

import torch

from . import _lib
from ._lib import PROBLEM_CVRP, PROBLEM_TSP, MAX_TOKENS, TILE_STRIDE, check, ptr, stream_ptr

def on_own_device(method):
    """Run an engine method with the engine's device current (kernels, their attributes and the stream handed to
    the library then all belong to that device, whatever device the caller had selected)."""
    @functools.wraps(method)
    def wrapper(self, *args, **kw):
        if torch.cuda.current_device() == self.device.index:
            return method(self, *args, **kw)
        with torch.cuda.device(self.device):
            return method(self, *args, **kw)
    return wrapper


_BLOCKS = ["encoder.layers.0.", "decoder.layers.0.", "decoder.layers.1.", "decoder.layers.2."]


def default_arithmetic() -> int:
    """Arithmetic of the dense contractions (include/dgl_amd.h, DGL_ARITH_*): the f16 split-operand
    matrix cores unless DGL_ARITHMETIC=f32 asks for the f32-input matrix instructions."""
    name = os.environ.get("DGL_ARITHMETIC", "split_f16").lower()
    if name in ("f32", "fp32"):
        return _lib.ARITH_F32
    if name in ("split_f16", "f16x3"):
        return _lib.ARITH_SPLIT_F16
    raise ValueError(f"DGL_ARITHMETIC={name!r}: expected 'split_f16' or 'f32'")


def weights_struct(sd: Dict[str, torch.Tensor], problem: int) -> _lib.PolicyWeights:
    """dgl_policy_weights_t over device tensors keyed like the reference's state_dict
    (TSP/TSPModel.py:238-339, CVRP/CVRPModel.py:378-427)."""
    w = _lib.PolicyWeights()
    w.problem = problem
    w.emb_w, w.emb_b = ptr(sd["encoder.embedding.weight"]), ptr(sd["encoder.embedding.bias"])
    for i, pre in enumerate(_BLOCKS):
        b = w.block[i]
        b.wq, b.wk, b.wv = ptr(sd[pre + "Wq.weight"]), ptr(sd[pre + "Wk.weight"]), ptr(sd[pre + "Wv.weight"])
        b.wc, b.bc = ptr(sd[pre + "multi_head_combine.weight"]), ptr(sd[pre + "multi_head_combine.bias"])
        b.w1, b.b1 = ptr(sd[pre + "feedForward.W1.weight"]), ptr(sd[pre + "feedForward.W1.bias"])
        b.w2, b.b2 = ptr(sd[pre + "feedForward.W2.weight"]), ptr(sd[pre + "feedForward.W2.bias"])
    for a, nm in enumerate(anchor_names(problem)):
        w.anchor_w[a], w.anchor_b[a] = ptr(sd[nm + ".weight"]), ptr(sd[nm + ".bias"])
    w.final_w, w.final_b = ptr(sd["decoder.Linear_final.weight"]), ptr(sd["decoder.Linear_final.bias"])
    return w


def anchor_names(problem: int):
    if problem == PROBLEM_TSP:
        return ["decoder.embedding_last_node", "decoder.embedding_last_node"]
    return ["decoder.embedding_last_node_not_via_depot", "decoder.embedding_last_node_via_depot"]


class PackedPolicy:
    """Weights of TSPModel / CVRPModel re-laid out for the MFMA kernel.

    ``state_dict`` uses the reference's checkpoint keys (TSP/TSPModel.py:238-339,
    CVRP/CVRPModel.py:378-427)."""

    def __init__(self, state_dict: Dict[str, torch.Tensor], problem: int, device: torch.device):
        self.problem = problem
        self.device = device
        lib = _lib.load()
        sd = {k: v.detach().to(device=device, dtype=torch.float32).contiguous() for k, v in state_dict.items()}
        feat = 7 if problem == PROBLEM_TSP else 9
        if tuple(sd["encoder.embedding.weight"].shape) != (128, feat):
            raise ValueError(f"encoder.embedding.weight has shape {tuple(sd['encoder.embedding.weight'].shape)}, "
                             f"expected (128, {feat})")
        self.sd = sd                                 # raw weights stay resident: the general (n > 32 tokens) path reads them
        self.w = w = weights_struct(sd, problem)
        self.packed = torch.empty(lib.dgl_policy_packed_floats(problem), dtype=torch.float32, device=device)
        with torch.cuda.device(device):
            check(lib.dgl_policy_pack(ctypes.byref(w), ptr(self.packed), stream_ptr()), "dgl_policy_pack")
            torch.cuda.current_stream().synchronize()      # sd may be freed after this returns


def general_workspace(engine, R: int, n: int) -> torch.Tensor:
    """Scratch of dgl_policy_general for R rows of n tokens, cached on the engine."""
    need = engine.lib.dgl_policy_general_workspace_floats(R, n)
    ws = getattr(engine, "_general_ws", None)
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.float32, device=engine.device)
        engine._general_ws = ws
    return ws


def wide_forward_enabled() -> bool:
    """Rows of 33..128 tokens in one launch (dgl_policy_forward's wide kernels); DGL_WIDE_FORWARD=0 sends them through
    the general path's sequence of launches instead (A/B measurements)."""
    return os.environ.get("DGL_WIDE_FORWARD", "1") != "0"


def policy_forward(engine, policy: PackedPolicy, io: _lib.PolicyIO) -> None:
    """The one-launch kernels for rows of at most 32 tokens, and of 33..128 tokens under the split-f16 arithmetic; the
    general path otherwise."""
    if engine.out_w == MAX_TOKENS:
        check(engine.lib.dgl_policy_forward(ptr(policy.packed), ctypes.byref(io), stream_ptr()), "dgl_policy_forward")
    elif (MAX_TOKENS < io.n <= _lib.MAX_WIDE_TOKENS and engine.arithmetic == _lib.ARITH_SPLIT_F16 and wide_forward_enabled()):
        check(engine.lib.dgl_policy_forward(ptr(policy.packed), ctypes.byref(io), stream_ptr()), "dgl_policy_forward")
    else:
        ws = general_workspace(engine, io.R, io.n)
        check(engine.lib.dgl_policy_general(ctypes.byref(policy.w), None, ctypes.byref(io), 0, 0.0, ptr(ws), 0,
                                            stream_ptr()), "dgl_policy_general")


def range_guarded(rollout):
    """Rollout methods under the split-f16 arithmetic: if the kernel reported an operand at the edge of the binary16
    range (|v| >= 65504; a network without LayerNorm does not bound its activations), the whole rollout is repeated
    with the binary32 kernel.  One 4-byte read per rollout."""
    @functools.wraps(rollout)
    def wrapper(self, *args, **kw):
        self.overflow.zero_()
        out = rollout(self, *args, **kw)
        self.reran_f32 = False
        if self.arithmetic == _lib.ARITH_SPLIT_F16 and int(self.overflow.item()) != 0:
            self.arithmetic = _lib.ARITH_F32
            try:
                out = rollout(self, *args, **kw)
            finally:
                self.arithmetic = _lib.ARITH_SPLIT_F16
            self.reran_f32 = True
        return out
    return wrapper


def forward_range_guarded(engine, policy: "PackedPolicy") -> None:
    """One policy evaluation for the reference-protocol path (model.forward): repeated with the binary32 kernel
    when the split-f16 kernel reports an operand at the edge of the binary16 range."""
    engine.overflow.zero_()
    engine.forward(policy)
    if engine.arithmetic == _lib.ARITH_SPLIT_F16 and int(engine.overflow.item()) != 0:
        engine.arithmetic = _lib.ARITH_F32
        try:
            engine.forward(policy)
        finally:
            engine.arithmetic = _lib.ARITH_SPLIT_F16


class TspEngine:
    """Compact TSP state for R = B * rows_per_instance rows plus per-step scratch."""

    def __init__(self, xy: torch.Tensor, rows_per_instance: int, knn: int):
        assert xy.is_cuda and xy.dtype == torch.float32 and xy.dim() == 3 and xy.shape[2] == 2
        self.lib = _lib.load()
        self.arithmetic = default_arithmetic()
        self.xy = xy.contiguous()
        self.B, self.N = xy.shape[0], xy.shape[1]
        self.rpi = rows_per_instance
        self.R = self.B * rows_per_instance
        self.knn = knn
        dev = torch.device("cuda", xy.device.index if xy.device.index is not None else torch.cuda.current_device())
        self.device = dev
        from . import sqrt_model
        sqrt_model.apply_environment(dev)
        R, N = self.R, self.N
        self.W = (N + 63) // 64
        self.bufs = [self._alloc_state() for _ in range(2 if rows_per_instance > 1 else 1)]   # 2nd copy: beam permutation target
        self.cur = 0
        self.idx = torch.empty(R, knn, dtype=torch.int32, device=dev)
        self.tile = torch.zeros(R, (knn + 1) * TILE_STRIDE, dtype=torch.float32, device=dev)
        self._alloc_outputs()
        self.selected = torch.zeros(R, dtype=torch.int32, device=dev)
        self.length = torch.zeros(R, dtype=torch.float32, device=dev)
        self.overflow = torch.zeros(1, dtype=torch.int32, device=dev)      # set by the split-f16 kernel (range guard)
        self.t = 0
        self.tile_t = -1                 # step count the neighbourhood buffers (idx, tile) belong to
        self._beam_bufs = None

    def _alloc_outputs(self) -> None:
        """logits / prob rows hold one float per token: 32 for the fused kernels, more for larger neighbourhoods."""
        w = max(MAX_TOKENS, self.knn + 1)
        if getattr(self, "out_w", None) != w:
            self.out_w = w
            self.out_stride = 0 if w == MAX_TOKENS else w
            self.logits = torch.zeros(self.R, w, dtype=torch.float32, device=self.device)
            self.prob = torch.zeros(self.R, w, dtype=torch.float32, device=self.device)

    def set_knn(self, knn: int) -> None:
        """Change the neighbourhood size (re-allocates the per-step scratch)."""
        if knn == self.knn:
            return
        self.knn = knn
        self.idx = torch.empty(self.R, knn, dtype=torch.int32, device=self.device)
        self.tile = torch.zeros(self.R, (knn + 1) * TILE_STRIDE, dtype=torch.float32, device=self.device)
        self.tile_t = -1
        self._alloc_outputs()

    @on_own_device
    def ensure_neighbourhood(self) -> None:
        """Build idx/tile for the current state if the last step did not (dgl_tsp_knn)."""
        if self.tile_t == self.t:
            return
        check(self.lib.dgl_tsp_knn(ctypes.byref(self.state["struct"]), self.t, self.knn, ptr(self.idx), ptr(self.tile),
                                   stream_ptr()), "dgl_tsp_knn")
        self.tile_t = self.t

    def _alloc_state(self):
        R, N, dev = self.R, self.N, self.device
        t = dict(avg=torch.empty(R, N, dtype=torch.float32, device=dev),
                 sd=torch.empty(R, N, dtype=torch.float32, device=dev),
                 visited=torch.zeros(R, self.W, dtype=torch.int64, device=dev),
                 tour=torch.zeros(R, N, dtype=torch.int32, device=dev))
        s = _lib.TspState(self.B, N, self.rpi, R, ptr(self.xy), ptr(t["avg"]), ptr(t["sd"]), ptr(t["visited"]),
                          ptr(t["tour"]))
        t["struct"] = s
        return t

    @property
    def state(self):
        return self.bufs[self.cur]

    @on_own_device
    def reset(self) -> None:
        check(self.lib.dgl_tsp_reset(ctypes.byref(self.state["struct"]), stream_ptr()), "dgl_tsp_reset")
        self.t = 0
        self.tile_t = -1

    @on_own_device
    def step(self, selected: torch.Tensor, want_knn: bool = True) -> None:
        """Apply ``selected`` (int32 [R]) and, unless the tour is complete, build the next neighbourhood."""
        assert selected.dtype == torch.int32 and selected.is_cuda and selected.numel() == self.R
        self.t += 1
        more = want_knn and self.t < self.N
        check(self.lib.dgl_tsp_step(ctypes.byref(self.state["struct"]), ptr(selected), self.t, self.knn,
                                    ptr(self.idx) if more else 0, ptr(self.tile) if more else 0, stream_ptr()),
              "dgl_tsp_step")
        if more:
            self.tile_t = self.t

    @property
    def k(self) -> int:
        return min(self.knn, self.N - self.t)

    def policy_io(self, want_logits=True, want_prob=True, want_selected=True) -> _lib.PolicyIO:
        k = self.k
        return _lib.PolicyIO(PROBLEM_TSP, self.R, k + 1, k, self.N, self.knn, 0, (self.knn + 1) * TILE_STRIDE,
                             self.arithmetic, self.out_stride, ptr(self.tile), ptr(self.idx), 0, 0,
                             ptr(self.logits) if want_logits else 0, ptr(self.prob) if want_prob else 0,
                             ptr(self.selected) if want_selected else 0, ptr(self.overflow))

    @on_own_device
    def forward(self, policy: PackedPolicy, **kw) -> None:
        policy_forward(self, policy, self.policy_io(**kw))

    @on_own_device
    def dense_probs(self, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        if out is None:
            out = torch.empty(self.R, self.N, dtype=torch.float32, device=self.device)
        io = self.policy_io()
        check(self.lib.dgl_policy_scatter_probs(ctypes.byref(io), ptr(self.prob), ptr(out), self.N, stream_ptr()),
              "dgl_policy_scatter_probs")
        return out

    @on_own_device
    def permute(self, prev_k: torch.Tensor, beam: int) -> None:
        assert len(self.bufs) == 2 and prev_k.dtype == torch.int32
        src, dst = self.bufs[self.cur], self.bufs[1 - self.cur]
        check(self.lib.dgl_tsp_permute(ctypes.byref(src["struct"]), ctypes.byref(dst["struct"]), ptr(prev_k), beam,
                                       self.t, stream_ptr()), "dgl_tsp_permute")
        self.cur = 1 - self.cur

    @on_own_device
    def data(self) -> torch.Tensor:
        out = torch.empty(self.R, self.N, 7, dtype=torch.float32, device=self.device)
        check(self.lib.dgl_tsp_state_data(ctypes.byref(self.state["struct"]), self.t, ptr(out), stream_ptr()),
              "dgl_tsp_state_data")
        return out

    @on_own_device
    def tour_length(self, xy: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Closed-tour length of every row; ``xy`` [B][N][2]: measure on these coordinates instead of the
        engine's own (aug_test evaluates on the original instance, TSP/TSPEnv.py:508-509)."""
        st = self.state["struct"]
        if xy is not None:
            assert xy.shape == self.xy.shape and xy.is_cuda and xy.dtype == torch.float32
            self._len_xy = xy.contiguous()                      # keep alive until the kernel has run
            t = self.state
            st = _lib.TspState(self.B, self.N, self.rpi, self.R, ptr(self._len_xy), ptr(t["avg"]), ptr(t["sd"]),
                               ptr(t["visited"]), ptr(t["tour"]))
        check(self.lib.dgl_tsp_tour_length(ctypes.byref(st), ptr(self.length), stream_ptr()), "dgl_tsp_tour_length")
        return self.length

    def fused_decode_available(self) -> bool:
        """dgl_tsp_greedy_decode takes rows of at most 32 tokens under the split-f16 arithmetic whose step arrays fit
        two rows to a workgroup's LDS.  Opt-in (DGL_FUSED_DECODE=1): measured 2-4 % SLOWER than two launches per
        decision at 512 rows x 1000 nodes - inside one long-running kernel the workgroups drift apart, the weight
        stream loses the lock-step sharing of the per-XCD L2 and every matrix phase takes ~35 % more cycles, which
        outweighs the two kernel boundaries saved per decision."""
        if os.environ.get("DGL_FUSED_DECODE", "0") != "1":
            return False
        from . import sqrt_model
        if sqrt_model.current(self.device) != "ieee":          # the fused kernel's step phase takes correctly rounded roots
            return False
        if self.N <= 25:                                       # ... and the matrix-product distances of more than 25 points
            return False
        per_row = 4 * (self.N + (self.knn + 1) * TILE_STRIDE + 128 + 2 * 192 + 8 + 2 * self.N)
        return self.out_w == MAX_TOKENS and self.arithmetic == _lib.ARITH_SPLIT_F16 and 2 * per_row <= 150 * 1024

    @on_own_device
    @range_guarded
    def greedy_rollout(self, policy: PackedPolicy, start: torch.Tensor) -> torch.Tensor:
        """Whole greedy decode (beam 1): N-1 policy decisions, everything stays on the device.  One launch for the
        whole loop (dgl_tsp_greedy_decode) where that kernel applies, otherwise two launches per decision."""
        self.reset()
        self.step(start)
        if self.t < self.N and self.fused_decode_available():
            io = self.policy_io(want_logits=False, want_prob=False)
            check(self.lib.dgl_tsp_greedy_decode(ctypes.byref(self.state["struct"]), ptr(policy.packed), ctypes.byref(io),
                                                 self.t, self.N, self.knn, stream_ptr()), "dgl_tsp_greedy_decode")
            self.t = self.N
            self.tile_t = -1
            return self.tour_length()
        while self.t < self.N:
            self.forward(policy, want_logits=False, want_prob=False)
            self.step(self.selected)
        return self.tour_length()

    @on_own_device
    def beam_advance(self, scores: torch.Tensor, beam: int, step: int, prev_k: torch.Tensor, action: torch.Tensor):
        """Beam expansion from the compact per-token probabilities of the last forward()."""
        io = self.policy_io()
        check(self.lib.dgl_beam_advance_compact(ctypes.byref(io), ptr(self.prob), ptr(scores), self.R // beam, beam,
                                                self.N, step, ptr(prev_k), ptr(action), stream_ptr()),
              "dgl_beam_advance_compact")

    @on_own_device
    def step_beam(self, prev_k: torch.Tensor, selected: torch.Tensor, beam: int) -> None:
        """Back-pointer gather and step in one pass over the state (other buffer <- this buffer); the tours are
        rebuilt from the history by ``backtrack``."""
        assert len(self.bufs) == 2 and prev_k.dtype == torch.int32 and selected.dtype == torch.int32
        if getattr(self, "hist_node", None) is None:
            self.hist_node = torch.zeros(self.N, self.R, dtype=torch.int32, device=self.device)
            self.hist_parent = torch.zeros(self.N, self.R, dtype=torch.int32, device=self.device)
        src, dst = self.bufs[self.cur], self.bufs[1 - self.cur]
        self.t += 1
        more = self.t < self.N
        check(self.lib.dgl_tsp_step_beam(ctypes.byref(src["struct"]), ctypes.byref(dst["struct"]), ptr(prev_k), beam,
                                         ptr(selected), self.t, self.knn, ptr(self.idx) if more else 0,
                                         ptr(self.tile) if more else 0, ptr(self.hist_node), ptr(self.hist_parent),
                                         stream_ptr()), "dgl_tsp_step_beam")
        self.cur = 1 - self.cur
        if more:
            self.tile_t = self.t

    @on_own_device
    def backtrack(self, t_first: int) -> None:
        check(self.lib.dgl_tsp_beam_backtrack(ctypes.byref(self.state["struct"]), ptr(self.hist_node),
                                              ptr(self.hist_parent), t_first, self.t, stream_ptr()),
              "dgl_tsp_beam_backtrack")

    @on_own_device
    @range_guarded
    def beam_rollout(self, policy: PackedPolicy, start: torch.Tensor, beam: int) -> torch.Tensor:
        """Beam-search decode (TSP/TSPTester.py:145-199) on the compact state: per step the policy, the beam
        expansion over beam*k candidates and one pass over the state that follows the back-pointers while it
        updates; the tours are rebuilt from the back-pointer history at the end.  Nothing returns to the host."""
        assert self.rpi % beam == 0 and len(self.bufs) == 2
        G = self.R // beam
        dev = self.device
        scores = torch.zeros(G, beam, dtype=torch.float32, device=dev)
        prev_k = torch.zeros(G, beam, dtype=torch.int32, device=dev)
        action = torch.zeros(G, beam, dtype=torch.int32, device=dev)
        self.reset()
        self.step(start)
        step = 1
        while self.t < self.N:
            self.forward(policy, want_logits=False, want_selected=False)
            self.beam_advance(scores, beam, step, prev_k, action)
            self.step_beam(prev_k.view(-1), action.view(-1), beam)
            step += 1
        if self.N >= 2:
            self.backtrack(2)
        self.beam_scores = scores
        return self.tour_length()


class CvrpEngine:
    """Compact CVRP state.  ``xy`` [B][N+1][2] (depot first), ``demand`` [B][N+1] (depot 0,
    already divided by the vehicle capacity)."""

    def __init__(self, xy: torch.Tensor, demand: torch.Tensor, rows_per_instance: int, knn: int, depot_knn: int,
                 demand_in_distance: bool = False):
        """``demand_in_distance``: distances over (x, y, demand), as the reference's training reset builds its
        matrix (CVRP/CVRPEnv.py:519); the test / validation reset measures (x, y) only."""
        self.demand_in_distance = bool(demand_in_distance)
        assert xy.is_cuda and xy.dtype == torch.float32 and xy.dim() == 3 and xy.shape[2] == 2
        assert demand.is_cuda and demand.dtype == torch.float32 and demand.shape == xy.shape[:2]
        self.lib = _lib.load()
        self.arithmetic = default_arithmetic()
        self.xy, self.demand = xy.contiguous(), demand.contiguous()
        self.B, self.M = xy.shape[0], xy.shape[1]
        self.N = self.M - 1
        self.rpi = rows_per_instance
        self.R = self.B * rows_per_instance
        self.knn, self.depot_knn = knn, depot_knn
        dev = torch.device("cuda", xy.device.index if xy.device.index is not None else torch.cuda.current_device())
        self.device = dev
        from . import sqrt_model
        sqrt_model.apply_environment(dev)
        R = self.R
        self.W = (self.M + 63) // 64
        self.ntok_max = knn + depot_knn + 2
        self.bufs = [self._alloc_state() for _ in range(2 if rows_per_instance > 1 else 1)]   # 2nd copy: beam permutation target
        self.cur = 0
        self.idx_last = torch.empty(R, knn, dtype=torch.int32, device=dev)
        self.idx_dep = torch.empty(R, depot_knn, dtype=torch.int32, device=dev)
        self.infeasible = torch.zeros(R, knn, dtype=torch.uint8, device=dev)
        self.tile = torch.zeros(R, self.ntok_max * TILE_STRIDE, dtype=torch.float32, device=dev)
        self._alloc_outputs()
        self.selected = torch.zeros(R, dtype=torch.int32, device=dev)
        self.length = torch.zeros(R, dtype=torch.float32, device=dev)
        self.overflow = torch.zeros(1, dtype=torch.int32, device=dev)      # set by the split-f16 kernel (range guard)
        self.t = 0
        self.tile_t = -1

    def _alloc_outputs(self) -> None:
        w = max(MAX_TOKENS, self.ntok_max)
        if getattr(self, "out_w", None) != w:
            self.out_w = w
            self.out_stride = 0 if w == MAX_TOKENS else w
            self.logits = torch.zeros(self.R, w, dtype=torch.float32, device=self.device)
            self.prob = torch.zeros(self.R, w, dtype=torch.float32, device=self.device)

    def set_knn(self, knn: int, depot_knn: int) -> None:
        if knn == self.knn and depot_knn == self.depot_knn:
            return
        self.knn, self.depot_knn = knn, depot_knn
        R, dev = self.R, self.device
        self.ntok_max = knn + depot_knn + 2
        self.idx_last = torch.empty(R, knn, dtype=torch.int32, device=dev)
        self.idx_dep = torch.empty(R, depot_knn, dtype=torch.int32, device=dev)
        self.infeasible = torch.zeros(R, knn, dtype=torch.uint8, device=dev)
        self.tile = torch.zeros(R, self.ntok_max * TILE_STRIDE, dtype=torch.float32, device=dev)
        self.tile_t = -1
        self._alloc_outputs()

    @on_own_device
    def ensure_neighbourhood(self) -> None:
        """Build both neighbourhoods and the tile for the current state if the last step did not
        (dgl_cvrp_knn; nudges near-zero capacities exactly once per step, like the reference)."""
        if self.tile_t == self.t:
            return
        check(self.lib.dgl_cvrp_knn(ctypes.byref(self.state["struct"]), self.t, self.knn, self.depot_knn,
                                    ptr(self.idx_last), ptr(self.idx_dep), ptr(self.infeasible), ptr(self.tile),
                                    stream_ptr()), "dgl_cvrp_knn")
        self.tile_t = self.t

    def _alloc_state(self):
        R, N, M, dev = self.R, self.N, self.M, self.device
        t = dict(avg=torch.empty(R, M, dtype=torch.float32, device=dev),
                 sd=torch.empty(R, M, dtype=torch.float32, device=dev),
                 capacity=torch.ones(R, dtype=torch.float32, device=dev),
                 visited=torch.zeros(R, self.W, dtype=torch.int64, device=dev),
                 nodes=torch.zeros(R, N, dtype=torch.int32, device=dev),
                 flags=torch.zeros(R, N, dtype=torch.int8, device=dev))
        t["struct"] = _lib.CvrpState(self.B, N, self.rpi, R, ptr(self.xy), ptr(self.demand), ptr(t["avg"]),
                                     ptr(t["sd"]), ptr(t["capacity"]), ptr(t["visited"]), ptr(t["nodes"]),
                                     ptr(t["flags"]), 1 if self.demand_in_distance else 0)
        return t

    @property
    def state(self):
        return self.bufs[self.cur]

    @on_own_device
    def reset(self) -> None:
        check(self.lib.dgl_cvrp_reset(ctypes.byref(self.state["struct"]), stream_ptr()), "dgl_cvrp_reset")
        self.t = 0
        self.tile_t = -1

    @on_own_device
    def step(self, action: torch.Tensor, want_knn: bool = True) -> None:
        assert action.dtype == torch.int32 and action.is_cuda and action.numel() == self.R
        self.t += 1
        more = want_knn and self.t < self.N
        check(self.lib.dgl_cvrp_step(ctypes.byref(self.state["struct"]), ptr(action), self.t, self.knn, self.depot_knn,
                                     ptr(self.idx_last) if more else 0, ptr(self.idx_dep) if more else 0,
                                     ptr(self.infeasible) if more else 0, ptr(self.tile) if more else 0,
                                     stream_ptr()), "dgl_cvrp_step")
        if more:
            self.tile_t = self.t

    @property
    def kd(self) -> int:
        return min(self.depot_knn, self.N - self.t)

    @property
    def kl(self) -> int:
        return min(self.knn, self.kd)

    def policy_io(self, want_logits=True, want_prob=True, want_selected=True) -> _lib.PolicyIO:
        kl, kd = self.kl, self.kd
        return _lib.PolicyIO(PROBLEM_CVRP, self.R, kl + kd + 2, kl, self.N, self.knn, self.depot_knn,
                             self.ntok_max * TILE_STRIDE, self.arithmetic, self.out_stride, ptr(self.tile), ptr(self.idx_last), ptr(self.idx_dep),
                             ptr(self.infeasible),
                             ptr(self.logits) if want_logits else 0, ptr(self.prob) if want_prob else 0,
                             ptr(self.selected) if want_selected else 0, ptr(self.overflow))

    @on_own_device
    def forward(self, policy: PackedPolicy, **kw) -> None:
        policy_forward(self, policy, self.policy_io(**kw))

    @on_own_device
    def dense_probs(self, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        A = 2 * (self.N + 1)
        if out is None:
            out = torch.empty(self.R, A, dtype=torch.float32, device=self.device)
        io = self.policy_io()
        check(self.lib.dgl_policy_scatter_probs(ctypes.byref(io), ptr(self.prob), ptr(out), A, stream_ptr()),
              "dgl_policy_scatter_probs")
        return out

    @on_own_device
    def permute(self, prev_k: torch.Tensor, beam: int) -> None:
        assert len(self.bufs) == 2 and prev_k.dtype == torch.int32
        src, dst = self.bufs[self.cur], self.bufs[1 - self.cur]
        check(self.lib.dgl_cvrp_permute(ctypes.byref(src["struct"]), ctypes.byref(dst["struct"]), ptr(prev_k), beam,
                                        self.t, stream_ptr()), "dgl_cvrp_permute")
        self.cur = 1 - self.cur

    @on_own_device
    def data(self) -> torch.Tensor:
        out = torch.empty(self.R, self.M, 8, dtype=torch.float32, device=self.device)
        check(self.lib.dgl_cvrp_state_data(ctypes.byref(self.state["struct"]), self.t, ptr(out), stream_ptr()),
              "dgl_cvrp_state_data")
        return out

    @on_own_device
    def route_length(self, xy: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Route length of every row; ``xy`` [B][N+1][2]: measure on these coordinates instead of the engine's
        own (aug_test evaluates on the original instance, CVRP/CVRPEnv.py:864-865)."""
        st = self.state["struct"]
        if xy is not None:
            assert xy.shape == self.xy.shape and xy.is_cuda and xy.dtype == torch.float32
            self._len_xy = xy.contiguous()                      # keep alive until the kernel has run
            t = self.state
            st = _lib.CvrpState(self.B, self.N, self.rpi, self.R, ptr(self._len_xy), ptr(self.demand), ptr(t["avg"]),
                                ptr(t["sd"]), ptr(t["capacity"]), ptr(t["visited"]), ptr(t["nodes"]),
                                ptr(t["flags"]), 0)
        check(self.lib.dgl_cvrp_route_length(ctypes.byref(st), ptr(self.length), stream_ptr()),
              "dgl_cvrp_route_length")
        return self.length

    tour_length = route_length

    @on_own_device
    @range_guarded
    def greedy_rollout(self, policy: PackedPolicy, start_action: torch.Tensor) -> torch.Tensor:
        self.reset()
        self.step(start_action)
        while self.t < self.N:
            self.forward(policy, want_logits=False, want_prob=False)
            self.step(self.selected)
        return self.route_length()

    @on_own_device
    def beam_advance(self, scores: torch.Tensor, beam: int, step: int, prev_k: torch.Tensor, action: torch.Tensor):
        io = self.policy_io()
        check(self.lib.dgl_beam_advance_compact(ctypes.byref(io), ptr(self.prob), ptr(scores), self.R // beam, beam,
                                                2 * (self.N + 1), step, ptr(prev_k), ptr(action), stream_ptr()),
              "dgl_beam_advance_compact")

    @on_own_device
    def step_beam(self, prev_k: torch.Tensor, action: torch.Tensor, beam: int) -> None:
        """Back-pointer gather and step in one pass over the state (other buffer <- this buffer); nodes and flags
        are rebuilt from the history by ``backtrack``."""
        assert len(self.bufs) == 2 and prev_k.dtype == torch.int32 and action.dtype == torch.int32
        if getattr(self, "hist_action", None) is None:
            self.hist_action = torch.zeros(self.N, self.R, dtype=torch.int32, device=self.device)
            self.hist_parent = torch.zeros(self.N, self.R, dtype=torch.int32, device=self.device)
        src, dst = self.bufs[self.cur], self.bufs[1 - self.cur]
        self.t += 1
        more = self.t < self.N
        check(self.lib.dgl_cvrp_step_beam(ctypes.byref(src["struct"]), ctypes.byref(dst["struct"]), ptr(prev_k), beam,
                                          ptr(action), self.t, self.knn, self.depot_knn,
                                          ptr(self.idx_last) if more else 0, ptr(self.idx_dep) if more else 0,
                                          ptr(self.infeasible) if more else 0, ptr(self.tile) if more else 0,
                                          ptr(self.hist_action), ptr(self.hist_parent), stream_ptr()), "dgl_cvrp_step_beam")
        self.cur = 1 - self.cur
        if more:
            self.tile_t = self.t

    @on_own_device
    def backtrack(self, t_first: int) -> None:
        check(self.lib.dgl_cvrp_beam_backtrack(ctypes.byref(self.state["struct"]), ptr(self.hist_action),
                                               ptr(self.hist_parent), t_first, self.t, stream_ptr()),
              "dgl_cvrp_beam_backtrack")

    @on_own_device
    @range_guarded
    def beam_rollout(self, policy: PackedPolicy, start_action: torch.Tensor, beam: int) -> torch.Tensor:
        """Beam-search decode (CVRP/CVRPTester.py:147-227) on the compact state; the step follows the
        back-pointers while it updates, nodes and flags are rebuilt from the history at the end."""
        assert self.rpi % beam == 0 and len(self.bufs) == 2
        G = self.R // beam
        dev = self.device
        scores = torch.zeros(G, beam, dtype=torch.float32, device=dev)
        prev_k = torch.zeros(G, beam, dtype=torch.int32, device=dev)
        action = torch.zeros(G, beam, dtype=torch.int32, device=dev)
        self.reset()
        self.step(start_action)
        step = 1
        while self.t < self.N:
            self.forward(policy, want_logits=False, want_selected=False)
            self.beam_advance(scores, beam, step, prev_k, action)
            self.step_beam(prev_k.view(-1), action.view(-1), beam)
            step += 1
        if self.N >= 2:
            self.backtrack(2)
        self.beam_scores = scores
        return self.route_length()
