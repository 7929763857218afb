"""Time of one training decode step (forward in mode 'train', backward, Adam) of the CVRP policy at the
token count of the reference's training configuration (knn = depot_knn = 100: 202 tokens per row)."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from dgl_amd._lib import PROBLEM_CVRP
from dgl_amd.engine import CvrpEngine
from dgl_amd.train import PolicyTrainer
from dgl_amd.weights import random_state_dict
R = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
k = int(sys.argv[3]) if len(sys.argv) > 3 else 100
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
xy = torch.rand(R, N + 1, 2, generator=g).to(dev)
dem = torch.cat((torch.zeros(R, 1), torch.randint(1, 200, (R, N), generator=g) / 500.0), 1).to(dev)
eng = CvrpEngine(xy, dem, 1, k, k, demand_in_distance=True)
tr = PolicyTrainer(random_state_dict('CVRP', 1), PROBLEM_CVRP, dev)
eng.reset()
eng.step(torch.full((R,), 5 + N + 1, dtype=torch.int32, device=dev))
teacher = eng.idx_last[:, 0].clone()
def one():
    tr.zero_grad(); tr.forward_backward(eng, teacher); tr.optimizer_step()
for _ in range(3): one()
torch.cuda.synchronize()
t0 = time.perf_counter()
iters = 20
for _ in range(iters): one()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / iters * 1e3
n = 2 * k + 2
flops = 3 * R * (4 * (n * 393216 + 512 * n * n))
print(f"R={R} N={N} n={n}: {ms:.2f} ms per training decode step (forward + backward + Adam), "
      f"{flops / ms / 1e9:.1f} TFLOP/s of dense work (3 x forward)")
