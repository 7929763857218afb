"""CPU oracle for the DGL decode step (TEST INFRASTRUCTURE, not product code).

A restatement, in PyTorch-CPU ops, of the arithmetic the reference performs on
its per-step path.  Only tests, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this module; the product
package never does.

Every number on the reference's path is produced by PyTorch ATen kernels
(cdist, argsort, gather, addmm/matmul, softmax, topk); the reference pins no
torch version, so the oracle of record is torch 2.10.0 (CPU) — the build in
this image.  The restatement keeps the reference's operand shapes so that the
same ATen/BLAS kernels are hit; it is checked bit-for-bit against vectors
captured from the imported reference (``oracle/make_goldens.py`` ->
``tests/golden/*.npz``).

Reference lines restated (paths relative to the reference checkout):
  TSP/TSPEnv.py:216-235   reset           -> TSPOracle.reset
  TSP/TSPEnv.py:318-516   step            -> TSPOracle.step
  TSP/TSPEnv.py:554-564   tour length     -> tsp_tour_length
  TSP/TSPModel.py:50-100  kNN state       -> TSPOracle.knn_indices / forward
  TSP/TSPModel.py:158-172 min-max norm    -> minmax_norm
  TSP/TSPModel.py:238-457 encoder/decoder -> transformer_block / TSPOracle.forward
  CVRP/CVRPEnv.py:456-482,591-873,951-993 -> CVRPOracle.reset/step, cvrp_route_length
  CVRP/CVRPModel.py:51-187,258-371,407-523 -> CVRPOracle.forward
  utils/beamsearch.py:52-131              -> beam_advance
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor

VISITED_PENALTY = 1e2      # TSP/TSPModel.py:88, CVRP/CVRPModel.py:114-116
PROB_SMALL = 1e-5          # TSP/TSPModel.py:323
PROB_BUMP = 1e-7           # TSP/TSPModel.py:325
PROB_ABSENT = 1e-20        # TSP/TSPModel.py:337
HEADS = 8
QKV = 16


# --------------------------------------------------------------------------
# shared pieces
# --------------------------------------------------------------------------
def minmax_norm(tile: Tensor) -> Tensor:
    """Per row, per feature min-max over the token axis; 0 where the range is 0.

    TSP/TSPModel.py:158-172 and CVRP/CVRPModel.py:258-276.  The CVRP version
    saves ``tmp = tile[:, :, 2]`` and writes it back afterwards, but ``tmp`` is a
    view of the tensor that the masked assignment rewrites in place, so the
    write-back is a no-op: the demand column is normalised like every other
    column (confirmed by the captured token tiles).
    """
    lo = tile.min(dim=1)[0]
    rng = tile.max(dim=1)[0] - lo
    lo = lo.unsqueeze(1).expand_as(tile)
    rng = rng.unsqueeze(1).expand_as(tile)
    nz = rng != 0
    out = torch.zeros_like(tile)
    out[nz] = (tile[nz] - lo[nz]) / rng[nz]
    return out


def _heads(x: Tensor) -> Tensor:
    r, n, _ = x.shape
    return x.reshape(r, n, HEADS, -1).transpose(1, 2)


def transformer_block(x: Tensor, w: Dict[str, Tensor], prefix: str) -> Tensor:
    """One encoder/decoder block (identical arithmetic, TSP/TSPModel.py:341-457):
    q,k,v without bias, 8 heads x 16, softmax(q k^T / 4) v, combine + residual,
    W2 relu(W1 x + b1) + b2 + residual.  No LayerNorm."""
    q = _heads(F.linear(x, w[prefix + "Wq.weight"]))
    k = _heads(F.linear(x, w[prefix + "Wk.weight"]))
    v = _heads(F.linear(x, w[prefix + "Wv.weight"]))
    score = torch.matmul(q, k.transpose(2, 3))
    score = score / torch.sqrt(torch.tensor(float(QKV)))
    att = torch.softmax(score, dim=3)
    out = torch.matmul(att, v).transpose(1, 2)
    out = out.reshape(x.shape[0], x.shape[1], HEADS * QKV)
    o1 = x + F.linear(out, w[prefix + "multi_head_combine.weight"], w[prefix + "multi_head_combine.bias"])
    hid = F.relu(F.linear(o1, w[prefix + "feedForward.W1.weight"], w[prefix + "feedForward.W1.bias"]))
    o2 = F.linear(hid, w[prefix + "feedForward.W2.weight"], w[prefix + "feedForward.W2.bias"])
    return o1 + o2


def _floor_small(p: Tensor) -> Tensor:
    small = p <= PROB_SMALL
    out = p.clone()
    out[small] = out[small] + torch.tensor(PROB_BUMP, dtype=p.dtype)
    return out


def beam_advance(log_probs: Tensor, scores: Tensor, step: int, beam: int, action_size: int, stable: bool = False
                 ) -> Tuple[Tensor, Tensor, Tensor]:
    """utils/beamsearch.py:65-99.  log_probs [G, beam, A]; scores [G, beam].
    Returns (new scores [G,beam], prev_k [G,beam], action [G,beam]).  ``stable``: equal values are taken in
    order of their flat index (the reference's torch.topk leaves the order of ties to its implementation)."""
    if step >= 2:
        lk = log_probs + scores.unsqueeze(2).expand_as(log_probs)
    else:
        lk = log_probs.clone()
        lk[:, 1:] = -1e20
    lk = lk.reshape(lk.shape[0], -1)
    if stable:
        best, ids = lk.sort(dim=1, descending=True, stable=True)
        best, ids = best[:, :beam], ids[:, :beam]
    else:
        best, ids = lk.topk(beam, 1, True, True)
    prev_k = torch.div(ids, action_size, rounding_mode="trunc")
    action = ids - prev_k * action_size
    return best, prev_k, action


def _perm(x: Tensor, prev_k: Tensor, beam: int) -> Tensor:
    """Gather along the beam axis (TSP/TSPEnv.py:278-308)."""
    g = prev_k.shape[0]
    shp = x.shape
    y = x.reshape(g, beam, -1)
    y = y.gather(1, prev_k.unsqueeze(2).expand_as(y))
    return y.reshape(shp)


def tsp_tour_length(xy: Tensor, tour: Tensor) -> Tensor:
    """Closed-tour length with the direct formula (TSP/TSPEnv.py:554-564)."""
    idx = tour.unsqueeze(2).expand(-1, tour.shape[1], 2)
    seq = xy.gather(1, idx)
    nxt = seq.roll(dims=1, shifts=-1)
    return ((seq - nxt) ** 2).sum(2).sqrt().sum(1)


def cvrp_route_length(xy: Tensor, nodes: Tensor, flags: Tensor) -> Tensor:
    """Route length with depot detours (CVRP/CVRPEnv.py:951-993).  xy [R,N+1,2];
    nodes/flags [R,N].  Per step: prev->cur if flag 0, else prev->depot->cur;
    plus the closing return to the depot."""
    n = nodes.shape[1]
    via = flags > 0
    mid = torch.where(via, torch.zeros_like(nodes), nodes)      # depot where via, else the node itself
    prev = nodes.roll(dims=1, shifts=1)

    def loc(ix):
        return xy.gather(1, ix.unsqueeze(2).expand(-1, n, 2))

    first = ((loc(nodes) - loc(mid)) ** 2)
    mid2 = mid.clone()
    mid2[:, 0] = 0
    second = ((loc(prev) - loc(mid2)) ** 2)
    return (first.sum(2).sqrt() + second.sum(2).sqrt()).sum(1)


# --------------------------------------------------------------------------
# TSP
# --------------------------------------------------------------------------
def chain_cdist(p: Tensor) -> Tensor:
    """torch.cdist(p, p) restated for more than 25 points per instance with elementwise IEEE operations so that the
    result does not depend on which matrix-product kernel the BLAS library picks for this CPU and this size:
    [-2 p_i, |p_i|^2, 1] . [p_j, 1, |p_j|^2] accumulated left to right, first product rounded, every later term one
    fused multiply-add, then sqrt(max(., 0)).  binary32 FMAs are evaluated in binary64 (the product is exact there)
    and rounded once; the square root is the correctly rounded one.  p [B, M, C], C = 2 or 3 (training reset:
    x, y, demand; CVRP/CVRPEnv.py:519).  In the container that generated the golden vectors this equals torch.cdist
    except where that build's vectorised float sqrt is 1 ulp off (about 0.6 % of the entries; tests/test_oracle_c.py)."""
    p = p.to(torch.float32)
    if p.shape[1] <= 25:
        # at most 25 points: ATen evaluates every pair directly, sqrt(fma(dz, dz, fma(dy, dy, dx dx))) with the scalar
        # (correctly rounded) sqrt: symmetric, zero diagonal
        d = p[:, :, None, :] - p[:, None, :, :]
        acc = d[..., 0] * d[..., 0]
        for c in range(1, p.shape[-1]):
            acc = (d[..., c].double() * d[..., c].double() + acc.double()).float()
        return acc.double().sqrt().float()
    sq = p * p
    n = sq[..., 0] + sq[..., 1]
    for c in range(2, p.shape[-1]):
        n = n + sq[..., c]
    a = (-2.0 * p).double()
    b = p.double()
    acc = (a[:, :, None, 0] * b[:, None, :, 0]).float()
    for c in range(1, p.shape[-1]):
        acc = (a[:, :, None, c] * b[:, None, :, c] + acc.double()).float()
    acc = acc + n[:, :, None]
    acc = acc + n[:, None, :]
    return acc.clamp_min(0).double().sqrt().float()


def _sqrt(t: Tensor, ieee: bool) -> Tensor:
    """torch.sqrt, or the correctly rounded square root.  On CPU, torch evaluates a contiguous float sqrt with the math
    library it was built against (MKL's vector sqrt in its "high accuracy" mode in the container that generated the
    golden vectors): on the AVX-512 path that is a 14-bit reciprocal-square-root estimate y refined once,
    s = x y, s + (x - s s)(y / 2), which comes out one ulp below the correctly rounded root on 0.6 % of all inputs (every
    time the root lies less than 0.06 ulp above the midpoint of two floats) - and other CPUs take other paths.  The
    root in binary64 rounded to binary32 is the correctly rounded binary32 root (53 >= 2 * 24 + 2 bits)."""
    return t.double().sqrt().float() if ieee else torch.sqrt(t)


class TSPOracle:
    """State + policy of the TSP path.  ``w`` is a reference-format state_dict."""

    F = 7  # x, y, d_to_current, avg_unvisited_d, std_unvisited_d, sin_to_start, cos_to_start

    def __init__(self, weights: Dict[str, Tensor], knn: int = 30):
        self.w = {k: v.detach().to(torch.float32) for k, v in weights.items()}
        self.knn = knn

    # -- environment ------------------------------------------------------
    def reset(self, xy: Tensor, rows_per_instance: int = 1, chain: bool = False) -> None:
        """``chain``: the same arithmetic on every CPU - the matrix from chain_cdist instead of torch.cdist and every
        square root correctly rounded (torch's float sqrt on CPU is the math library's, see ``_sqrt``)."""
        self.ieee_sqrt = chain
        self.B, self.N = xy.shape[0], xy.shape[1]
        self.rpi = rows_per_instance
        self.R = self.B * rows_per_instance
        self.xy_inst = xy
        self.dis = chain_cdist(xy) if chain else torch.cdist(xy, xy, p=2)
        self.xy = xy.repeat_interleave(rows_per_instance, 0)
        self.avg = self.dis.mean(dim=-1).repeat_interleave(rows_per_instance, dim=0)
        self.std = self.dis.std(unbiased=False, dim=-1).repeat_interleave(rows_per_instance, dim=0)
        self.tour = torch.zeros((self.R, 0), dtype=torch.long)
        self.t = 0
        self.d_cur = None
        self.sin = None
        self.cos = None

    def _col(self, node: Tensor) -> Tensor:
        """dis[b, :, node] for every row (column ``node`` of the row's instance)."""
        inst = torch.arange(self.B).repeat_interleave(self.rpi)
        return self.dis[inst[:, None], torch.arange(self.N)[None, :], node[:, None]]

    def step(self, selected: Tensor) -> bool:
        self.t += 1
        self.tour = torch.cat((self.tour, selected[:, None]), dim=1)
        u = self.N - self.t
        d = self._col(selected)
        self.d_cur = d
        avg_before = self.avg
        self.avg = (self.avg * (u + 1) - d) / u
        self.std = _sqrt((torch.square(self.std) * (u + 1) - torch.square(d - avg_before)) / u, self.ieee_sqrt)
        if self.t == 1:
            r = torch.arange(self.R)
            start = self.xy[r, selected]
            dy = start[:, 1].unsqueeze(-1).repeat(1, self.N) - self.xy[:, :, 1]
            dx = start[:, 0].unsqueeze(-1).repeat(1, self.N) - self.xy[:, :, 0]
            self.sin = dy / (d + 1e-7)
            self.cos = dx / (d + 1e-7)
        return self.t == self.N

    def data(self) -> Tensor:
        """The reference's ``step_state.data`` [R,N,7] (after at least one step)."""
        return torch.stack((self.xy[:, :, 0], self.xy[:, :, 1], self.d_cur, self.avg, self.std,
                            self.sin, self.cos), dim=-1)

    def permute_beams(self, prev_k: Tensor, beam: int) -> None:
        for name in ("xy", "avg", "std", "d_cur", "sin", "cos", "tour"):
            setattr(self, name, _perm(getattr(self, name), prev_k, beam))

    def length(self) -> Tensor:
        return tsp_tour_length(self.xy, self.tour)

    # -- policy -----------------------------------------------------------
    def knn_indices(self) -> Tensor:
        last = self.tour[:, -1]
        inst = torch.arange(self.B).repeat_interleave(self.rpi)
        row = self.dis[inst, last]                                   # row `last` (TSPModel.py:83-84)
        mask = torch.zeros_like(row)
        mask[torch.arange(self.R)[:, None], self.tour] = VISITED_PENALTY
        k = min(self.knn, self.N - self.t)
        idx = torch.argsort(row + mask, dim=1, stable=True)[:, :k]   # lowest index wins ties
        return torch.sort(idx, dim=-1)[0]

    def tile(self, idx: Tensor) -> Tensor:
        data = self.data()
        knn_node = data.gather(1, idx.unsqueeze(-1).repeat(1, 1, self.F))
        last = data.gather(1, self.tour[:, [-1]][:, :, None].expand(self.R, 1, self.F))
        return minmax_norm(torch.cat((knn_node, last), dim=1))

    def logits_from_tile(self, tile: Tensor) -> Tensor:
        w = self.w
        h = F.linear(tile, w["encoder.embedding.weight"], w["encoder.embedding.bias"])
        h = transformer_block(h, w, "encoder.layers.0.")
        last = F.linear(h[:, -1], w["decoder.embedding_last_node.weight"], w["decoder.embedding_last_node.bias"])
        out = torch.cat((h[:, 0:-1], last.unsqueeze(1)), dim=1)
        for l in range(3):
            out = transformer_block(out, w, f"decoder.layers.{l}.")
        return F.linear(out, w["decoder.Linear_final.weight"], w["decoder.Linear_final.bias"]).squeeze(-1)

    def forward(self) -> Dict[str, Tensor]:
        idx = self.knn_indices()
        tile = self.tile(idx)
        logits = self.logits_from_tile(tile)
        masked = logits.clone()
        masked[:, [-1]] = masked[:, [-1]] + float("-inf")
        p = _floor_small(torch.softmax(masked, dim=-1)[:, 0:-1])
        probs = torch.full((self.R, self.N), PROB_ABSENT)
        probs[torch.arange(self.R)[:, None], idx] = p
        return {"idx": idx, "tile": tile, "logits": logits, "p": p, "probs": probs}


def tsp_rollout(oracle: TSPOracle, xy: Tensor, start: Tensor, beam: int = 1, pomo: int = 1,
                record=None, chain: bool = False) -> Dict[str, Tensor]:
    """Rollout loop of TSP/TSPTester.py:145-199 (test_mode 'pomo_test', no aug).
    ``start`` holds the pomo start nodes ([pomo] int64); ``chain``: see TSPOracle.reset."""
    B, N = xy.shape[0], xy.shape[1]
    oracle.reset(xy, rows_per_instance=pomo * beam, chain=chain)
    G = B * pomo
    scores = torch.zeros(G, beam)
    step = 0
    sel = start.repeat(B).repeat_interleave(beam).to(torch.int64)
    done = oracle.step(sel)
    step = 1
    while not done:
        out = oracle.forward()
        if record is not None:
            record(step, oracle, out)
        logp = torch.log(out["probs"].view(G, beam, -1))
        scores, prev_k, action = beam_advance(logp, scores, step, beam, N)
        oracle.permute_beams(prev_k, beam)
        sel = action.reshape(-1)
        step += 1
        done = oracle.step(sel)
    length = oracle.length()
    best, which = length.view(B, beam * pomo).min(dim=1)
    return {"tour": oracle.tour, "length": length, "best_length": best, "best_row": which,
            "score": best.mean()}


# --------------------------------------------------------------------------
# CVRP
# --------------------------------------------------------------------------
class CVRPOracle:
    """State + policy of the CVRP path.  Nodes are 0 (depot) .. N; an action id a
    in [0, 2(N+1)) means "go to node a" (a <= N) or "go to node a-(N+1) via the
    depot" (CVRP/CVRPEnv.py:612-622)."""

    F = 8  # x, y, demand, d_to_current, avg, std, sin_to_depot, cos_to_depot

    def __init__(self, weights: Dict[str, Tensor], knn: int = 15, depot_knn: int = 15):
        self.w = {k: v.detach().to(torch.float32) for k, v in weights.items()}
        self.knn = knn
        self.depot_knn = depot_knn

    def reset(self, depot_xy: Tensor, node_xy: Tensor, demand: Tensor, rows_per_instance: int = 1,
              train: bool = False, chain: bool = False) -> None:
        """depot_xy [B,2]; node_xy [B,N,2]; demand [B,N] (already / capacity).  ``train``: the distance
        matrix of reset('train') (CVRP/CVRPEnv.py:519), cdist over (x, y, demand).  ``chain``: the same arithmetic on
        every CPU (chain_cdist instead of torch.cdist, correctly rounded square roots: see TSPOracle.reset)."""
        B, N = node_xy.shape[0], node_xy.shape[1]
        self.ieee_sqrt = chain
        self.B, self.N, self.rpi = B, N, rows_per_instance
        self.R = B * rows_per_instance
        xy = torch.cat((depot_xy[:, None, :], node_xy), dim=1)
        dem = torch.cat((torch.zeros(B, 1), demand), dim=1)
        self.problems = torch.cat((xy, dem[:, :, None]), dim=2)
        pts = self.problems if train else xy
        self.dis = (chain_cdist(pts) if chain else torch.cdist(pts, pts, p=2)).to(torch.float32)
        self.xy = xy.repeat_interleave(rows_per_instance, 0)
        self.demand = dem.repeat_interleave(rows_per_instance, 0)
        self.capacity = torch.ones(self.R)
        self.ninf = torch.zeros(self.R, N + 1)
        self.avg = self.dis.mean(dim=-1).repeat_interleave(rows_per_instance, dim=0)
        self.std = self.dis.std(unbiased=False, dim=-1).repeat_interleave(rows_per_instance, dim=0)
        self.nodes = torch.zeros((self.R, 0), dtype=torch.long)
        self.flags = torch.zeros((self.R, 0), dtype=torch.long)
        self.t = 0
        self.d_cur = self.d_depot = self.sin = self.cos = None

    def _col(self, node: Tensor) -> Tensor:
        inst = torch.arange(self.B).repeat_interleave(self.rpi)
        return self.dis[inst[:, None], torch.arange(self.N + 1)[None, :], node[:, None]]

    def step(self, action: Tensor) -> bool:
        N = self.N
        via = action > N
        node = torch.where(via, action - (N + 1), action)
        flag = via.to(torch.long)
        dem = self.demand[torch.arange(self.R), node]
        cap = self.capacity.clone()
        cap[via] = 1 - dem[via]
        cap[~via] = self.capacity[~via] - dem[~via]
        self.capacity = cap
        too_large = (self.capacity[:, None] + 0.000001) < self.demand
        self.ninf = torch.zeros(self.R, N + 1)
        self.ninf[too_large] = float("-inf")
        self.t += 1
        self.nodes = torch.cat((self.nodes, node[:, None]), dim=1)
        self.flags = torch.cat((self.flags, flag[:, None]), dim=1)
        u = N + 1 - self.t
        d = self._col(node)
        self.d_cur = d
        if self.t == 1:
            self.d_depot = self._col(torch.zeros(self.R, dtype=torch.long))
        avg_before = self.avg
        self.avg = (self.avg * (u + 1) - d) / u
        item = (torch.square(self.std) * (u + 1) - torch.square(d - avg_before)) / u
        item[item < 0] = 0
        self.std = _sqrt(item, self.ieee_sqrt)
        if self.t == 1:
            dep = self.xy[:, 0, :]
            dy = dep[:, 1].unsqueeze(-1).repeat(1, N + 1) - self.xy[:, :, 1]
            dx = dep[:, 0].unsqueeze(-1).repeat(1, N + 1) - self.xy[:, :, 0]
            self.sin = dy / (self.d_depot + 1e-7)
            self.cos = dx / (self.d_depot + 1e-7)
        return self.t == N

    def data(self) -> Tensor:
        return torch.stack((self.xy[:, :, 0], self.xy[:, :, 1], self.demand, self.d_cur, self.avg,
                            self.std, self.sin, self.cos), dim=-1)

    def permute_beams(self, prev_k: Tensor, beam: int) -> None:
        for name in ("xy", "demand", "capacity", "ninf", "avg", "std", "d_cur", "d_depot", "sin", "cos",
                     "nodes", "flags"):
            setattr(self, name, _perm(getattr(self, name), prev_k, beam))

    def length(self) -> Tensor:
        return cvrp_route_length(self.xy, self.nodes, self.flags)

    def knn_indices(self) -> Tuple[Tensor, Tensor]:
        N, R = self.N, self.R
        last = self.nodes[:, -1]
        inst = torch.arange(self.B).repeat_interleave(self.rpi)
        d_last = self.dis[inst, last]
        d_dep = self.dis[inst, torch.zeros_like(last)]
        mask = torch.zeros_like(d_last)
        mask[torch.arange(R)[:, None], self.nodes] = VISITED_PENALTY
        mask[:, 0] = VISITED_PENALTY
        remaining = N - self.t
        kd = min(self.depot_knn, remaining)
        order = torch.argsort(torch.cat((d_last, d_dep), dim=0) + mask.repeat(2, 1), dim=1, stable=True)[:, :kd]
        idx_last = order[:R, :min(self.knn, kd)].sort(dim=-1)[0]
        idx_dep = order[R:, :].sort(dim=-1)[0]
        return idx_last, idx_dep

    def tiles(self, idx_last: Tensor, idx_dep: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
        R = self.R
        data = self.data()
        g_last = data.gather(1, idx_last.unsqueeze(-1).repeat(1, 1, self.F)).contiguous()
        g_dep = data.gather(1, idx_dep.unsqueeze(-1).repeat(1, 1, self.F)).contiguous()
        knn_ninf = self.ninf.gather(1, idx_last).contiguous()
        # in-place nudge of near-zero capacities, visible to later steps (CVRP/CVRPModel.py:164-166)
        low = self.capacity < 0.00001
        self.capacity[low] = self.capacity[low] + 0.0000001
        low = self.capacity < 0.00001
        capc = self.capacity.unsqueeze(-1)
        g_last[~low, :, 2] = g_last[~low, :, 2] / capc[~low]
        g_last[low, :, 2] = 1.1
        g_dep[:, :, 3] = self.d_depot.gather(1, idx_dep)
        last_node = data.gather(1, self.nodes[:, [-1]][:, :, None].expand(R, 1, self.F))
        depot_node = data[:, 0:1, :]
        t_last = minmax_norm(torch.cat((g_last, last_node), dim=1))
        t_dep = minmax_norm(torch.cat((g_dep, depot_node), dim=1))
        t_last = torch.cat((t_last, torch.zeros(R, t_last.shape[1], 1)), dim=-1)
        t_dep = torch.cat((t_dep, torch.ones(R, t_dep.shape[1], 1)), dim=-1)
        return torch.cat((t_last, t_dep), dim=1), knn_ninf, last_node

    def logits_from_tile(self, tile: Tensor, k_last: int) -> Tensor:
        """Returns logits in DECODER token order [last', knn.., depot', depot-knn..]."""
        w = self.w
        h = F.linear(tile, w["encoder.embedding.weight"], w["encoder.embedding.bias"])
        h = transformer_block(h, w, "encoder.layers.0.")
        e_nv = F.linear(h[:, k_last], w["decoder.embedding_last_node_not_via_depot.weight"],
                        w["decoder.embedding_last_node_not_via_depot.bias"])
        e_v = F.linear(h[:, -1], w["decoder.embedding_last_node_via_depot.weight"],
                       w["decoder.embedding_last_node_via_depot.bias"])
        out = torch.cat((e_nv.unsqueeze(1), h[:, 0:k_last], e_v.unsqueeze(1), h[:, k_last + 1:-1]), dim=1)
        for l in range(3):
            out = transformer_block(out, w, f"decoder.layers.{l}.")
        return F.linear(out, w["decoder.Linear_final.weight"], w["decoder.Linear_final.bias"]).squeeze(-1)

    def forward(self) -> Dict[str, Tensor]:
        N, R = self.N, self.R
        idx_last, idx_dep = self.knn_indices()
        tile, knn_ninf, _ = self.tiles(idx_last, idx_dep)
        kl = idx_last.shape[1]
        logits = self.logits_from_tile(tile, kl)
        m = logits.clone()
        m[:, [0, kl + 1]] = m[:, [0, kl + 1]] + float("-inf")
        m[:, 1:kl + 1] = m[:, 1:kl + 1] + knn_ninf
        p = torch.softmax(m, dim=-1)
        p = _floor_small(torch.cat((p[:, 1:kl + 1], p[:, kl + 2:]), dim=1))
        probs = torch.full((R, 2 * (N + 1)), PROB_ABSENT)
        rows = torch.arange(R)[:, None]
        probs[rows, idx_last] = p[:, :kl]
        probs[rows, idx_dep + N + 1] = p[:, kl:]
        return {"idx_last": idx_last, "idx_dep": idx_dep, "tile": tile, "logits": logits, "p": p,
                "probs": probs, "knn_ninf": knn_ninf}


def cvrp_rollout(oracle: CVRPOracle, depot_xy: Tensor, node_xy: Tensor, demand: Tensor, start: Tensor,
                 beam: int = 1, pomo: int = 1, record=None, chain: bool = False) -> Dict[str, Tensor]:
    """Rollout loop of CVRP/CVRPTester.py:147-227 ('pomo_test').  ``start`` holds the
    pomo start customers as 0-based indices r; the first action is r + 2 + N.  ``chain``: see TSPOracle.reset."""
    B, N = node_xy.shape[0], node_xy.shape[1]
    oracle.reset(depot_xy, node_xy, demand, rows_per_instance=pomo * beam, chain=chain)
    G = B * pomo
    A = 2 * (N + 1)
    scores = torch.zeros(G, beam)
    sel = start.repeat(B).repeat_interleave(beam).to(torch.int64) + 2 + N
    done = oracle.step(sel)
    step = 1
    while not done:
        out = oracle.forward()
        if record is not None:
            record(step, oracle, out)
        logp = torch.log(out["probs"].view(G, beam, -1))
        scores, prev_k, action = beam_advance(logp, scores, step, beam, A)
        oracle.permute_beams(prev_k, beam)
        sel = action.reshape(-1)
        step += 1
        done = oracle.step(sel)
    length = oracle.length()
    best, which = length.view(B, beam * pomo).min(dim=1)
    return {"nodes": oracle.nodes, "flags": oracle.flags, "length": length, "best_length": best,
            "best_row": which, "score": best.mean()}


# --------------------------------------------------------------------------
# weights in the reference's checkpoint layout
# --------------------------------------------------------------------------
def _block_shapes(prefix: str):
    return {
        prefix + "Wq.weight": (128, 128), prefix + "Wk.weight": (128, 128), prefix + "Wv.weight": (128, 128),
        prefix + "multi_head_combine.weight": (128, 128), prefix + "multi_head_combine.bias": (128,),
        prefix + "feedForward.W1.weight": (512, 128), prefix + "feedForward.W1.bias": (512,),
        prefix + "feedForward.W2.weight": (128, 512), prefix + "feedForward.W2.bias": (128,),
    }


def state_dict_shapes(problem: str) -> Dict[str, tuple]:
    """Key -> shape of the reference's ``model_state_dict`` (46 tensors TSP, 48 CVRP;
    TSP/TSPModel.py:238-339, CVRP/CVRPModel.py:378-427), including the two
    never-used layers ``decoder.embedding_last_node_pos`` and ``decoder.k_1``."""
    feat = 7 if problem == "TSP" else 9
    s = {"encoder.embedding.weight": (128, feat), "encoder.embedding.bias": (128,)}
    s.update(_block_shapes("encoder.layers.0."))
    if problem == "TSP":
        s["decoder.embedding_last_node.weight"] = (128, 128)
        s["decoder.embedding_last_node.bias"] = (128,)
    else:
        for nm in ("not_via_depot", "via_depot"):
            s[f"decoder.embedding_last_node_{nm}.weight"] = (128, 128)
            s[f"decoder.embedding_last_node_{nm}.bias"] = (128,)
    s["decoder.embedding_last_node_pos.weight"] = (128, 128)
    s["decoder.embedding_last_node_pos.bias"] = (128,)
    for l in range(3):
        s.update(_block_shapes(f"decoder.layers.{l}."))
    s["decoder.k_1.weight"] = (128, 128)
    s["decoder.k_1.bias"] = (128,)
    s["decoder.Linear_final.weight"] = (1, 128)
    s["decoder.Linear_final.bias"] = (1,)
    return s


def random_weights(problem: str, seed: int, scale: float = 1.0) -> Dict[str, Tensor]:
    """nn.Linear-style uniform(-1/sqrt(fan_in), 1/sqrt(fan_in)) weights, own RNG stream
    (NOT the reference's init order; used where only shapes matter)."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k, shp in state_dict_shapes(problem).items():
        fan_in = shp[1] if len(shp) == 2 else None
        if fan_in is None:
            wkey = k[:-4] + "weight"
            fan_in = state_dict_shapes(problem)[wkey][1]
        bound = scale / math.sqrt(fan_in)
        out[k] = (torch.rand(shp, generator=g) * 2 - 1) * bound
    return out


# --------------------------------------------------------------------------- training step
def enable_grad(weights: Dict[str, Tensor]) -> Dict[str, Tensor]:
    """Leaf copies of the weights that record gradients (the reference's nn.Parameters)."""
    return {k: v.detach().clone().requires_grad_(True) for k, v in weights.items()}


def train_forward_backward(oracle, teacher: Tensor):
    """Model side of CVRPTrainer._train_one_batch / TSPTrainer (CVRP/CVRPTrainer.py:268-271,
    CVRP/CVRPModel.py:311-345, TSP/TSPModel.py:206-217): probability of the teacher's action from the
    dense tensor, loss = -log(prob.double()).mean(), backward through ``oracle.w`` (which must hold
    gradient-recording leaves).  Returns (prob [R], loss)."""
    for v in oracle.w.values():
        v.grad = None
    out = oracle.forward()
    prob = out["probs"][torch.arange(oracle.R)[:, None], teacher.long()[:, None]].reshape(-1)
    loss = -prob.type(torch.float64).log().mean()
    loss.backward()
    return prob.detach(), float(loss.detach())


def cvrp_train_batch(oracle: "CVRPOracle", xy: Tensor, demand: Tensor, solution: Tensor, solution_flag: Tensor,
                     lr: float = 1e-4, weight_decay: float = 1e-6, hook=None):
    """CVRPTrainer._train_one_batch (CVRP/CVRPTrainer.py:232-282): xy [B,N+1,2] (depot first; the
    coordinates reset('train') produced), demand [B,N], labels solution / solution_flag [B,N].
    ``oracle.w`` must hold gradient-recording leaves; one torch Adam step per decode step.
    Returns (probs [N-1][B], losses [N-1])."""
    N = xy.shape[1] - 1
    params = list(oracle.w.values())
    opt = torch.optim.Adam(params, lr=lr, weight_decay=weight_decay)
    act = solution + solution_flag * (N + 1)
    oracle.reset(xy[:, 0], xy[:, 1:], demand, train=True)
    oracle.step(act[:, 0].clone())
    probs, losses = [], []
    for t in range(1, N):
        prob, loss = train_forward_backward(oracle, act[:, t])
        if hook is not None:
            hook(t, oracle)
        opt.step()
        probs.append(prob)
        losses.append(loss)
        with torch.no_grad():
            oracle.step(act[:, t].clone())
    return torch.stack(probs, 0), losses


def tsp_train_batch(oracle: "TSPOracle", xy: Tensor, solution: Tensor, lr: float = 1e-4, weight_decay: float = 1e-6):
    """TSPTrainer._train_one_batch (TSP/TSPTrainer.py:216-256): xy [B,N,2] (the coordinates reset('train')
    produced), labels solution [B,N]; one torch Adam step per decode step.  Returns probs [N-1][B]."""
    N = xy.shape[1]
    opt = torch.optim.Adam(list(oracle.w.values()), lr=lr, weight_decay=weight_decay)
    oracle.reset(xy)
    oracle.step(solution[:, 0].clone())
    probs = []
    for t in range(1, N):
        prob, _ = train_forward_backward(oracle, solution[:, t])
        opt.step()
        probs.append(prob)
        with torch.no_grad():
            oracle.step(solution[:, t].clone())
    return torch.stack(probs, 0)


# --------------------------------------------------------------------------- self-improvement labels
def tsp_nearest_neighbour(xy: Tensor) -> Tuple[Tensor, Tensor]:
    """TSPEnv.greedy_search (TSP/TSPEnv.py:140-169): from node 0 to the nearest unvisited node by the cdist
    matrix; length = sum of the matrix entries around the tour.  Returns (solution [B,N], length [B])."""
    B, N, _ = xy.shape
    dis = torch.cdist(xy, xy, p=2)
    rows = torch.arange(B)
    sol = torch.zeros(B, 1, dtype=torch.int64)
    for _ in range(1, N):
        d = dis[rows, sol[:, -1]].clone()
        d[rows[:, None], sol] += 1e5
        sol = torch.cat((sol, torch.argmin(d, dim=1, keepdim=True)), dim=1)
    length = torch.sum(dis[rows[:, None], sol, torch.roll(sol, -1, dims=1)], dim=1)
    return sol, length


def cvrp_nearest_neighbour(xy: Tensor, demand: Tensor) -> Tuple[Tensor, Tensor]:
    """CVRPEnv.greedy_search (CVRP/CVRPEnv.py:340-409): xy [B,N+1,2] (depot first), demand [B,N+1]; nearest
    unvisited customer, via the depot when capacity + 1e-6 < demand (always for the first move).
    Returns (nodes [B,N], flags [B,N])."""
    B, M, _ = xy.shape
    dis = torch.cdist(xy, xy, p=2)
    rows = torch.arange(B)
    nodes = torch.zeros(B, 0, dtype=torch.int64)
    flags = torch.zeros(B, 0, dtype=torch.int64)
    cap = torch.ones(B)
    last = torch.zeros(B, dtype=torch.int64)
    for t in range(M - 1):
        d = dis[rows, last].clone()
        d[:, 0] += 1e5
        d[rows[:, None], nodes] += 1e5
        nxt = torch.argmin(d, dim=1)
        dem = demand[rows, nxt]
        via = torch.ones(B, dtype=torch.bool) if t == 0 else (cap + 0.000001) < dem
        cap = torch.where(via, 1 - dem, cap - dem)
        nodes = torch.cat((nodes, nxt[:, None]), dim=1)
        flags = torch.cat((flags, via.long()[:, None]), dim=1)
        last = nxt
    return nodes, flags


def cvrp_label_augment(nodes: Tensor, flags: Tensor, flip: Tensor, shift: Tensor) -> Tuple[Tensor, Tensor]:
    """CVRPEnv.data_augmentation (CVRP/CVRPEnv.py:325-338) with the draws given: flip[i][j] stands for the
    j-th ``torch.rand(1) > 0.5`` of instance i (j = 1..S-1), shift[i] for its ``randint(0, S)``."""
    nodes, flags = nodes.clone(), flags.clone()
    for i in range(nodes.shape[0]):
        idx = torch.nonzero(flags[i], as_tuple=True)[0]
        for j in range(1, len(idx)):
            if flip[i][j]:
                nodes[i][idx[j - 1]:idx[j]] = torch.flip(nodes[i][idx[j - 1]:idx[j]], [0])
        k = int(idx[int(shift[i]) % len(idx)])
        nodes[i] = torch.cat((nodes[i][k:], nodes[i][:k]))
        flags[i] = torch.cat((flags[i][k:], flags[i][:k]))
    return nodes, flags
