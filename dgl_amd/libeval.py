"""Benchmark-library evaluation the way the reference's drivers run it (TSP/test_tsplib.py:162-261,
CVRP/test_cvrplib.py:166-275): every instance on its own, scaled to the unit square, decided in
aug_test mode with the augmentation factor choose_bsz(size), scored on the original integer-rounded
scale against the best-known value.  The library files are not shipped; ``root`` must hold
``tsplib/<name>.tsp`` / ``vrplib/<name>.vrp``."""
from __future__ import annotations

import copy
import os
import tempfile
import time
from typing import Dict, Iterable, List, Optional

import torch

from . import formats as F
from .tester import CVRPTester, TSPTester


def _tester(cls, env_params, model_params, tester_params, size, aug, data_path):
    env_params, tester_params = copy.deepcopy(env_params), copy.deepcopy(tester_params)
    tester_params.update(problem_size=size, test_episodes_all=1, test_episodes=1, test_batch_size=1, test_mode="aug_test")
    env_params.update(aug_size=aug, test_mode="aug_test", data_path=data_path, baseline_path="", mode="test")
    return cls(env_params=env_params, model_params=model_params, tester_params=tester_params)


def _bucket_report(gaps: Dict[str, float], sizes: Dict[str, int], edges) -> List[str]:
    lines, lo = [], 1
    for hi in edges + [None]:
        sel = [gaps[n] for n in gaps if sizes[n] >= lo and (hi is None or sizes[n] <= hi)]
        label = f"{lo}~{hi}" if hi is not None else f">{lo - 1}"
        lines.append(f"{label:12s}: {len(sel)} instances, gap {F.avg_list(sel) * 100:.3f}%")
        lo = (hi or 0) + 1
    return lines


def run_tsplib(root, env_params, model_params, tester_params, names: Optional[Iterable[str]] = None,
               best_known: Optional[Dict[str, int]] = None, weights=None, log=print):
    """-> {name: (integer tour length, gap, seconds)}; ``weights``: state_dict loaded into every tester."""
    best_known = F.tsplib_collections if best_known is None else best_known
    names = sorted(best_known if names is None else names, key=lambda n: F.parse_tsplib_name(n)[1])
    out, gaps, sizes = {}, {}, {}
    with tempfile.TemporaryDirectory() as tmp:
        for name in names:
            instance, _ = F.load_tsplib_file(root, name)
            size = instance.size(0)
            path = os.path.join(tmp, "tmp_lib.txt")
            F.write_tsp_instances(path, F.normalize_nodes_to_unit_board(instance)[None])
            tester = _tester(TSPTester, env_params, model_params, tester_params, size, F.choose_bsz(size), path)
            if weights is not None:
                tester.model.load_state_dict(weights)
            t0 = time.time()
            tester.run()
            torch.cuda.synchronize()
            seconds = time.time() - t0
            length, gap = F.tsplib_gap(instance, tester.shortest_tours[0], best_known[name])
            out[name], gaps[name], sizes[name] = (length, gap, seconds), gap, size
            log(f"{name} score {length} opt_len {best_known[name]} gap {gap} time {seconds} s")
    for line in _bucket_report(gaps, sizes, [100, 1000, 10000]):
        log("TSP " + line)
    return out


def run_cvrplib(root, env_params, model_params, tester_params, names: Optional[Iterable[str]] = None,
                best_known: Optional[Dict[str, int]] = None, weights=None, log=print):
    """-> {name: (integer route length, gap, seconds)}."""
    best_known = F.cvrplib_collections if best_known is None else best_known
    names = sorted(best_known if names is None else names, key=lambda n: F.parse_cvrplib_name(n)[1])
    out, gaps, sizes = {}, {}, {}
    with tempfile.TemporaryDirectory() as tmp:
        for name in names:
            depot, nodes, demands, capacity, _ = F.load_cvrplib_file(root, name)
            size = nodes.size(0)
            assert size == F.parse_cvrplib_name(name)[1]
            # the customers and the depot share one scaling (CVRP/test_cvrplib.py:180-186)
            scaled = F.normalize_nodes_to_unit_board(torch.cat((nodes, depot.unsqueeze(0)), dim=0)).float()
            path = os.path.join(tmp, "tmp_lib.txt")
            F.write_cvrp_instances(path, scaled[-1:], scaled[None, :-1], demands[None], capacity)
            tester = _tester(CVRPTester, env_params, model_params, tester_params, size, F.choose_bsz(size), path)
            if weights is not None:
                tester.model.load_state_dict(weights)
            t0 = time.time()
            tester.run()
            torch.cuda.synchronize()
            seconds = time.time() - t0
            length, gap = F.cvrplib_gap(depot, nodes, tester.shortest_tours[0], tester.shortest_tours_flag[0],
                                        best_known[name])
            out[name], gaps[name], sizes[name] = (length, gap, seconds), gap, size
            log(f"{name} score {length} opt_len {best_known[name]} gap {gap} time {seconds} s")
    for line in _bucket_report(gaps, sizes, [200, 500]):
        log("CVRP " + line)
    return out
