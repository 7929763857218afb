"""The fused step kernel alone at a saturating batch (16384 rows x 1000 nodes), TSP or CVRP: 20 launches (for counter passes)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from dgl_amd.engine import CvrpEngine, TspEngine
prob = sys.argv[1] if len(sys.argv) > 1 else "tsp"
R, N = 16384, 1000
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(1)
if prob == "cvrp":
    xy = torch.rand(R, N + 1, 2, generator=g).to(dev)
    dem = torch.cat((torch.zeros(R, 1), torch.randint(1, 200, (R, N), generator=g) / 500.0), 1).to(dev)
    eng = CvrpEngine(xy, dem, 1, 15, 15)
    first = torch.full((R,), 7 + 2 + N, dtype=torch.int32, device=dev)
else:
    eng = TspEngine(torch.rand(R, N, 2, generator=g).to(dev), 1, 30)
    first = torch.full((R,), 7, dtype=torch.int32, device=dev)
eng.reset(); eng.step(first)
sels = [torch.full((R,), 100 + i, dtype=torch.int32, device=dev) for i in range(20)]
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for s in sels: eng.step(s)
b.record(); torch.cuda.synchronize()
print(f"{prob} step at {R} rows: {a.elapsed_time(b) / 20 * 1e3:.1f} us per launch")
