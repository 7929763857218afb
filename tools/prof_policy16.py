"""Per-phase cycle stamps of the f16 policy kernel (library built with -DPROFILE_STAMPS)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from dgl_amd._lib import PROBLEM_TSP
from dgl_amd.engine import PackedPolicy, TspEngine
from dgl_amd.weights import random_state_dict
R = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device('cuda:0')
xy = torch.rand(R, 1000, 2, device=dev)
eng = TspEngine(xy, 1, 30)
pol = PackedPolicy(random_state_dict('TSP', 1), PROBLEM_TSP, dev)
eng.reset(); eng.step(torch.zeros(R, dtype=torch.int32, device=dev))
for _ in range(3): eng.forward(pol)
torch.cuda.synchronize()
st = eng.logits.view(R, 32)[0::2].cpu()           # one record per workgroup (2 rows)
names = ["start", "embed"] + [f"L{l}.{p}" for l in range(4) for p in ("qkv", "attn", "bar", "comb", "bar", "ff", "bar")]
d = st[:, 1:] - st[:, :-1]
med = d.median(0).values
for i, nm in enumerate(names[1:]):
    print(f"{nm:10s} {med[i].item():9.0f}")
print("total median", st[:, 29].median().item(), "max", st[:, 29].max().item())
