"""Is the greedy rollout bound by the host's launch rate?  Time to enqueue one rollout vs time until it has run."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from dgl_amd._lib import PROBLEM_TSP
from dgl_amd.engine import PackedPolicy, TspEngine
from dgl_amd.weights import random_state_dict
B, N = 512, 1000
dev = torch.device('cuda:0')
xy = torch.rand(B, N, 2, device=dev)
eng = TspEngine(xy, 1, 30)
pol = PackedPolicy(random_state_dict('TSP', 1), PROBLEM_TSP, dev)
start = torch.zeros(B, dtype=torch.int32, device=dev)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.greedy_rollout(pol, start)
    t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"enqueue {1e3*(t1-t0):.1f} ms, complete {1e3*(t2-t0):.1f} ms ({1e6*(t1-t0)/(N-1):.1f} us of host time per decode step)")
