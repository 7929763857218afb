"""Compare the two arithmetic modes of dgl_policy_forward on the same tiles."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from dgl_amd import _lib
from dgl_amd._lib import PROBLEM_TSP
from dgl_amd.engine import PackedPolicy, TspEngine
from dgl_amd.weights import random_state_dict
R = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device('cuda:0')
torch.manual_seed(0)
xy = torch.rand(R, 200, 2, device=dev)
eng = TspEngine(xy, 1, 30)
pol = PackedPolicy(random_state_dict('TSP', 1), PROBLEM_TSP, dev)
eng.reset(); eng.step(torch.zeros(R, dtype=torch.int32, device=dev))
out = {}
for name, mode in (("f32", _lib.ARITH_F32), ("f16x3", _lib.ARITH_SPLIT_F16)):
    eng.arithmetic = mode
    eng.forward(pol)
    torch.cuda.synchronize()
    out[name] = (eng.logits.clone(), eng.prob.clone(), eng.selected.clone())
a, b = out["f32"][0][:, :31], out["f16x3"][0][:, :31]
print("logits f32  :", a[0, :8].tolist())
print("logits f16x3:", b[0, :8].tolist())
rel = ((a - b).abs().max() / a.abs().max()).item()
print(f"max |diff| / max |logit| = {rel:.3g}; selected equal: {(out['f32'][2] == out['f16x3'][2]).float().mean().item():.3f}")
