"""Micro-benchmark of the fused state-update + kNN + tile kernel alone (for profiling)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from dgl_amd.engine import TspEngine
R = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
knn = 30
dev = torch.device('cuda:0')
xy = torch.rand(R, N, 2, device=dev)
eng = TspEngine(xy, 1, knn)
eng.reset()
sels = [torch.randint(0, N, (R,), dtype=torch.int32, device=dev) for _ in range(iters + 3)]
for i in range(3): eng.step(sels[i])
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for i in range(iters): eng.step(sels[3 + i])
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / iters
by = (24 * N + N / 8 + 8 * knn) * R
print(f"R={R} N={N} step {ms*1e3:.1f} us  {by/ms/1e6:.0f} GB/s algorithmic ({by/ms/1e6/8000*100:.1f} % of 8 TB/s)")
