"""Fine cycle stamps inside block 2 of the f16 policy kernel (library built with -DPROFILE_STAMPS=2)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from dgl_amd._lib import PROBLEM_TSP
from dgl_amd.engine import PackedPolicy, TspEngine
from dgl_amd.weights import random_state_dict
R = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device('cuda:0')
eng = TspEngine(torch.rand(R, 1000, 2, device=dev), 1, 30)
pol = PackedPolicy(random_state_dict('TSP', 1), PROBLEM_TSP, dev)
eng.reset(); eng.step(torch.zeros(R, dtype=torch.int32, device=dev))
for _ in range(3): eng.forward(pol)
torch.cuda.synchronize()
st = eng.logits.view(R, 32)[0::2].cpu()
names = ["qkv", "attn", "bar", "comb", "bar", "G1(0)", "G1(1)|E(0)", "bar", "G2(0)|E(1)a", "G1(2)|E(1)b", "bar",
         "G2(1)|E(2)a", "G1(3)|E(2)b", "bar", "G2(2)|E(3)", "bar", "G2(3)", "resid+store", "bar"]
d = (st[:, 1:] - st[:, :-1]).median(0).values
for i, nm in enumerate(names):
    print(f"{nm:14s} {d[i].item():8.0f}")
print("block total", (st[:, len(names)] - st[:, 0]).median().item())
