"""Reset kernel (row mean / std of the pairwise distances, O(N^2) per instance): time per launch."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from dgl_amd.engine import TspEngine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
dev = torch.device('cuda:0')
eng = TspEngine(torch.rand(B, N, 2, device=dev), 1, 30)
for _ in range(2): eng.reset()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(5): eng.reset()
b.record(); torch.cuda.synchronize()
print(f"reset B={B} N={N}: {a.elapsed_time(b)/5*1e3:.0f} us per launch")
