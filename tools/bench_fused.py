"""Greedy rollout: one fused launch (dgl_tsp_greedy_decode) vs two launches per decision; same tours?"""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from dgl_amd._lib import PROBLEM_TSP
from dgl_amd.engine import PackedPolicy, TspEngine
from dgl_amd.weights import random_state_dict
R, N = int(sys.argv[1]) if len(sys.argv) > 1 else 512, int(sys.argv[2]) if len(sys.argv) > 2 else 1000
dev = torch.device('cuda:0')
xy = torch.rand(R, N, 2, device=dev)
eng = TspEngine(xy, 1, 30)
pol = PackedPolicy(random_state_dict('TSP', 1), PROBLEM_TSP, dev)
start = torch.full((R,), 5, dtype=torch.int32, device=dev)
res = {}
for mode in ("0", "1", "0", "1"):
    os.environ["DGL_FUSED_DECODE"] = mode
    eng.greedy_rollout(pol, start); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): length = eng.greedy_rollout(pol, start)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    res[mode] = (eng.state["tour"].clone(), length.clone())
    print(f"fused={mode}: {dt * 1e3:.2f} ms per rollout, {R / dt:.0f} instances/s", flush=True)
print("same tours:", torch.equal(res["0"][0], res["1"][0]), "same lengths:", torch.equal(res["0"][1], res["1"][1]))
