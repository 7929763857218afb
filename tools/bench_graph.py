"""Greedy rollout enqueued launch by launch vs replayed from one captured HIP graph (TSP-1000, 512 rows)."""
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

def body():
    eng.reset(); eng.step(start)
    while eng.t < eng.N:
        eng.forward(pol, want_logits=False, want_prob=False)
        eng.step(eng.selected)
    return eng.tour_length()

for _ in range(2): body()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3): l0 = body().clone()
torch.cuda.synchronize()
print(f"launch by launch: {(time.perf_counter() - t0) / 3 * 1e3:.2f} ms per rollout")
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    body()
torch.cuda.current_stream().wait_stream(s)
t0 = time.perf_counter()
with torch.cuda.graph(g):
    out = body()
print(f"capture: {time.perf_counter() - t0:.2f} s")
g.replay(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3): g.replay()
torch.cuda.synchronize()
print(f"graph replay: {(time.perf_counter() - t0) / 3 * 1e3:.2f} ms per rollout; same lengths: {torch.equal(out, l0)}")
