"""Per-phase cycles of the fused decode kernel (library built with -DFUSED_STAMPS; DGL_AMD_LIB points at it)."""
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
for rep in range(2):
    eng.reset(); eng.step(start)
    io = eng.policy_io(want_logits=True, want_prob=False)
    eng.logits.zero_()
    check(eng.lib.dgl_tsp_greedy_decode(ctypes.byref(eng.state["struct"]), ptr(pol.packed), ctypes.byref(io), 1, N, 30, stream_ptr()), "decode")
    torch.cuda.synchronize()
st = eng.logits.view(-1)[: (R // 2) * 16].view(R // 2, 4, 4).cpu()
steps = st[0, 0, 3].item()
for w in range(4):
    m = st[:, w, :3].median(0).values / steps
    print(f"wave {w}: per decode step: policy {m[0]:.0f} cycles, step {m[1]:.0f}, barriers {m[2]:.0f}; total {m.sum():.0f}")
