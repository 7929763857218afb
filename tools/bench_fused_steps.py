"""Greedy rollout with ONE launch per decision (policy phase + step phase fused, dgl_tsp_greedy_decode over one step)
against two launches per decision."""
import os, sys, time, ctypes, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from dgl_amd._lib import PROBLEM_TSP, check, ptr, stream_ptr
from dgl_amd.engine import PackedPolicy, TspEngine
from dgl_amd.weights import random_state_dict
R, N = 512, 1000
dev = torch.device('cuda:0')
eng = TspEngine(torch.rand(R, N, 2, device=dev), 1, 30)
pol = PackedPolicy(random_state_dict('TSP', 1), PROBLEM_TSP, dev)
start = torch.full((R,), 5, dtype=torch.int32, device=dev)
def fused_per_step():
    eng.reset(); eng.step(start)
    io = eng.policy_io(want_logits=False, want_prob=False)
    st = ctypes.byref(eng.state["struct"]); iop = ctypes.byref(io); P = ptr(pol.packed); s = stream_ptr()
    for t in range(1, N):
        check(eng.lib.dgl_tsp_greedy_decode(st, P, iop, t, t + 1, 30, s), "decode")
    eng.t = N
    return eng.tour_length()
os.environ["DGL_FUSED_DECODE"] = "0"
for name, fn in (("two launches per decision", lambda: eng.greedy_rollout(pol, start)), ("one fused launch per decision", fused_per_step)) * 2:
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): l = fn().clone()
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / 3 * 1e3:.2f} ms per rollout, mean length {l.mean().item():.5f}", flush=True)
