"""Second opinion for the tours-only population fixtures (runs in the build container only; test infrastructure).

The reference's greedy tours are not a function of the instance alone: they also depend on
  (1) how its torch build rounds float square roots - ``torch.sqrt`` and ``torch.cdist`` on CPU hand contiguous float
      tensors to the math library (see ``oracle.dgl_oracle._sqrt``): one ulp below the correctly rounded root on 0.6 % of
      inputs in the container that made the fixtures, something else on another CPU.  The running standard deviations
      (TSP/TSPEnv.py:408-409, CVRP/CVRPEnv.py:733-738) amplify a last-bit difference by N / #unvisited, and the min-max
      normalisation over a neighbourhood by another two orders of magnitude, so the last decisions of a tour can hinge
      on it;
  (2) how its unstable ``torch.argsort`` (TSP/TSPModel.py:89, CVRP/CVRPModel.py:121) orders two equal distances at the
      edge of a neighbourhood.
This script adds to a fixture, next to the reference's tours:
  ieee/tour | ieee/nodes, ieee/flags, ieee/length : the tours of the oracle run with correctly rounded square roots
      (``chain=True``) and its stable, lowest-index-first argsort - the reference's algorithm with the two
      library-dependent choices fixed the way the device kernels fix them;
  ref_tie : per instance, whether the reference's own trajectory passes a decision at which the k-th and (k+1)-th
      key of a neighbourhood are equal (the distances of the reference's own torch.cdist, which this container's torch
      reproduces bit for bit).

usage: python oracle/make_ieee_tours.py tests/golden/cvrp_n1000_b512_nn_tours.npz [...]
"""
from __future__ import annotations

import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle.dgl_oracle import CVRPOracle, TSPOracle, VISITED_PENALTY, cvrp_rollout, tsp_rollout   # noqa: E402


def _weights(case, golden_dir):
    wname = str(case["meta/weights"])
    path = os.path.join(golden_dir, wname)
    if wname.endswith(".pt"):
        return {k: v.float() for k, v in torch.load(path, map_location="cpu", weights_only=True)["model_state_dict"].items()}
    wz = np.load(path + ".npz")
    return {k: torch.from_numpy(wz[k]) for k in wz.files}


def _boundary_tie(rows: torch.Tensor, visited_before: torch.Tensor, k_of_step) -> bool:
    """rows [T, M]: the distance row each decision sorts; visited_before [T, M] bool."""
    key = rows + VISITED_PENALTY * visited_before.to(rows.dtype)
    srt = torch.sort(key, dim=1)[0]
    for t in range(rows.shape[0]):
        k = k_of_step(t)
        if k < rows.shape[1] and srt[t, k - 1] == srt[t, k] and srt[t, k] < VISITED_PENALTY:
            return True
    return False


def ref_ties_tsp(xy: np.ndarray, tour: np.ndarray, knn: int) -> np.ndarray:
    B, N = tour.shape
    out = np.zeros(B, bool)
    for b in range(B):
        p = torch.from_numpy(xy[b:b + 1])
        dis = torch.cdist(p, p, p=2)[0]
        tr = torch.from_numpy(tour[b].astype(np.int64))
        vis = torch.zeros(N - 1, N, dtype=torch.bool)
        for t in range(1, N):                      # decision t is taken with tour[:t] visited, from tour[t-1]
            vis[t - 1] = vis[t - 2] if t > 1 else False
            vis[t - 1, tr[t - 1]] = True
        out[b] = _boundary_tie(dis[tr[:N - 1]], vis, lambda i: min(knn, N - (i + 1)))
    return out


def ref_ties_cvrp(depot_xy, node_xy, nodes: np.ndarray, knn: int, depot_knn: int) -> np.ndarray:
    B, N = nodes.shape
    out = np.zeros(B, bool)
    for b in range(B):
        p = torch.from_numpy(np.concatenate([depot_xy[b:b + 1, None, :], node_xy[b:b + 1]], axis=1))
        dis = torch.cdist(p, p, p=2)[0]
        tr = torch.from_numpy(nodes[b].astype(np.int64))
        vis = torch.zeros(N - 1, N + 1, dtype=torch.bool)
        for t in range(1, N):
            vis[t - 1] = vis[t - 2] if t > 1 else False
            vis[t - 1, tr[t - 1]] = True
        vis[:, 0] = True                            # the depot column is masked like a visited node
        kd = lambda i: min(depot_knn, N - (i + 1))
        out[b] = (_boundary_tie(dis[tr[:N - 1]], vis, lambda i: min(knn, kd(i))) or
                  _boundary_tie(dis[0:1].expand(N - 1, N + 1), vis, kd))
    return out


def main():
    for path in sys.argv[1:]:
        z = np.load(path)
        case = {k: z[k] for k in z.files}
        w = _weights(case, os.path.dirname(os.path.abspath(path)))
        t0 = time.time()
        start = torch.from_numpy(case["start"]).long()
        if str(case["meta/problem"]) == "TSP":
            res = tsp_rollout(TSPOracle(w, int(case["meta/knn"])), torch.from_numpy(case["xy"]), start, chain=True)
            case["ieee/tour"] = res["tour"].numpy().astype(case["tour"].dtype)
            case["ref_tie"] = ref_ties_tsp(case["xy"], case["tour"], int(case["meta/knn"]))
            same = (case["ieee/tour"] == case["tour"]).all(axis=1)
        else:
            knn, kd = int(case["meta/knn"]), int(case["meta/depot_knn"])
            res = cvrp_rollout(CVRPOracle(w, knn, kd), torch.from_numpy(case["depot_xy"]),
                               torch.from_numpy(case["node_xy"]), torch.from_numpy(case["demand"]), start, chain=True)
            case["ieee/nodes"] = res["nodes"].numpy().astype(case["nodes"].dtype)
            case["ieee/flags"] = res["flags"].numpy().astype(np.int8)
            case["ref_tie"] = ref_ties_cvrp(case["depot_xy"], case["node_xy"], case["nodes"], knn, kd)
            same = (case["ieee/nodes"] == case["nodes"]).all(axis=1) & (case["ieee/flags"] == case["flags"]).all(axis=1)
        case["ieee/length"] = res["length"].numpy()
        np.savez_compressed(path, **case)
        print(f"{os.path.basename(path)}: the reference's tours and the correctly-rounded-sqrt oracle's agree on "
              f"{int(same.sum())}/{len(same)} instances ({int((same | case['ref_tie']).sum())} counting the "
              f"{int(case['ref_tie'].sum())} instances with a neighbourhood tie on the reference's trajectory as agreed); "
              f"{time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
