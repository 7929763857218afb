"""Phase timing of the wide policy kernel (rows of 33..128 tokens): needs the profiling build
  python -m dgl_amd.build --variant widestamps -DWIDE_STAMPS
run as  DGL_AMD_LIB=dgl_amd/lib/libdgl_amd_widestamps.so python tools/bench_wide_stamps.py [R] [k]
Thread 0 of every row's workgroup stores the cycle counter at each phase boundary (policy16_wide.h, WSTAMP) where the
probabilities would go; medians over the rows are printed per phase and per block."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from dgl_amd._lib import PROBLEM_CVRP
from dgl_amd.engine import CvrpEngine, PackedPolicy
from dgl_amd.weights import random_state_dict
R = int(sys.argv[1]) if len(sys.argv) > 1 else 256
k = int(sys.argv[2]) if len(sys.argv) > 2 else 50
N = 1000
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
xy = torch.rand(R, N + 1, 2, generator=g).to(dev)
dem = torch.cat((torch.zeros(R, 1), torch.randint(1, 200, (R, N), generator=g) / 500.0), 1).to(dev)
eng = CvrpEngine(xy, dem, 1, k, k)
pol = PackedPolicy(random_state_dict('CVRP', 1), PROBLEM_CVRP, dev)
eng.reset(); eng.step(torch.full((R,), 5 + 2 + N, dtype=torch.int32, device=dev))
for _ in range(3): eng.forward(pol)
eng.prob.zero_()
eng.forward(pol)
torch.cuda.synchronize()
st = eng.prob.cpu().numpy()
nst = int((st[0] > 0).sum()) + 1
st = st[:, :nst]
d = np.diff(st, axis=1)
med = np.median(d, axis=0)
names = ["embed"]
per_block = ["pre", "park+qkv", "bar", "attn", "bar", "comb"] + [f"ff{i}" for i in range(2 if k > 31 else 1)] + ["bar"]
for b in range(4):
    names += [f"b{b}.{p}" for p in per_block]
names += ["final"]
print(f"R={R} tokens {2 * k + 2}: {nst} stamps, total median {np.median(st[:, -1]):.0f} ticks of s_memtime")
for i, m in enumerate(med):
    print(f"{names[i] if i < len(names) else str(i):>12s} {m:8.0f}")
