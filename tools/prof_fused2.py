"""Per-phase cycles of the policy phase INSIDE the fused decode kernel (-DFUSED_STAMPS -DPROFILE_STAMPS), summed over steps."""
import os, sys, ctypes, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from dgl_amd._lib import PROBLEM_TSP, check, ptr, stream_ptr
from dgl_amd.engine import PackedPolicy, TspEngine
from dgl_amd.weights import random_state_dict
R, N = 512, 1000
dev = torch.device('cuda:0')
eng = TspEngine(torch.rand(R, N, 2, device=dev), 1, 30)
pol = PackedPolicy(random_state_dict('TSP', 1), PROBLEM_TSP, dev)
start = torch.full((R,), 5, dtype=torch.int32, device=dev)
eng.reset(); eng.step(start)
io = eng.policy_io(want_logits=True, want_prob=True)
eng.logits.zero_(); eng.prob.zero_()
steps = 400
check(eng.lib.dgl_tsp_greedy_decode(ctypes.byref(eng.state["struct"]), ptr(pol.packed), ctypes.byref(io), 1, 1 + steps, 30, stream_ptr()), "decode")
torch.cuda.synchronize()
st = eng.prob.view(R, 32)[0::2].cpu() / steps
names = ["start", "embed"] + [f"L{l}.{p}" for l in range(4) for p in ("qkv", "attn", "bar", "comb", "bar", "ff", "bar")]
d = st[:, 1:] - st[:, :-1]
med = d.median(0).values
for i, nm in enumerate(names[1:]):
    print(f"{nm:10s} {med[i].item():9.0f}")
print("total median", st[:, 29].median().item())
