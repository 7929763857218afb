"""Forward pass of the general path (rows of more than 32 tokens, inference): CVRP with knn = depot_knn = k."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from dgl_amd._lib import PROBLEM_CVRP
from dgl_amd.engine import CvrpEngine, PackedPolicy
from dgl_amd.weights import random_state_dict
R = int(sys.argv[1]) if len(sys.argv) > 1 else 512
k = int(sys.argv[2]) if len(sys.argv) > 2 else 50
N = 1000
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
xy = torch.rand(R, N + 1, 2, generator=g).to(dev)
dem = torch.cat((torch.zeros(R, 1), torch.randint(1, 200, (R, N), generator=g) / 500.0), 1).to(dev)
eng = CvrpEngine(xy, dem, 1, k, k)
pol = PackedPolicy(random_state_dict('CVRP', 1), PROBLEM_CVRP, dev)
eng.reset(); eng.step(torch.full((R,), 5 + 2 + N, dtype=torch.int32, device=dev))
for _ in range(3): eng.forward(pol, want_logits=False, want_prob=False)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20): eng.forward(pol, want_logits=False, want_prob=False)
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / 20
n = 2 * k + 2
fl = R * (4 * (n * 393216 + 512 * n * n) + n * 2 * 9 * 128 + 2 * 32768 + n * 256)
print(f"R={R} tokens per row {n}: {ms * 1e3:.0f} us per forward, {fl / ms / 1e9:.0f} TFLOP/s algorithmic")
