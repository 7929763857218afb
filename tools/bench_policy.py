"""Micro-benchmark of dgl_policy_forward alone (for profiling): R rows of random tiles."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from dgl_amd._lib import PROBLEM_TSP
from dgl_amd.engine import PackedPolicy, TspEngine
from dgl_amd.weights import random_state_dict
R = int(sys.argv[1]) if len(sys.argv) > 1 else 512
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50
dev = torch.device('cuda:0')
xy = torch.rand(R, 1000, 2, device=dev)
eng = TspEngine(xy, 1, 30)
pol = PackedPolicy(random_state_dict('TSP', 1), PROBLEM_TSP, dev)
eng.reset(); eng.step(torch.zeros(R, dtype=torch.int32, device=dev))
for _ in range(5): eng.forward(pol, want_logits=False, want_prob=False)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(iters): eng.forward(pol, want_logits=False, want_prob=False)
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / iters
fl = (4 * (31 * 393216 + 512 * 31 * 31) + 31 * 2 * 7 * 128 + 32768 + 31 * 256) * R
print(f"R={R} policy {ms*1e3:.1f} us  {fl/ms/1e9:.1f} TFLOP/s  ({fl/ms/1e9/157.3*100:.1f} % of fp32 MFMA peak)")
