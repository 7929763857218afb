"""Does the policy kernel's duration depend on the operand values (power-limited clocks)?  Times
dgl_policy_forward on random weights / tiles and on all-zero weights (same instruction stream)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from dgl_amd._lib import PROBLEM_TSP
from dgl_amd.engine import PackedPolicy, TspEngine
from dgl_amd.weights import random_state_dict
R, iters = 512, 2000
dev = torch.device('cuda:0')
xy = torch.rand(R, 1000, 2, device=dev)
eng = TspEngine(xy, 1, 30)
eng.reset(); eng.step(torch.zeros(R, dtype=torch.int32, device=dev))
sd = random_state_dict('TSP', 1)
tile0 = eng.tile.clone()
for tag, scale, bias_scale, zero_tile in (("random weights, random tiles", 1.0, 1.0, False), ("zero weights", 0.0, 0.0, False),
                                          ("random matrices, zero biases, zero tiles (all activations 0)", 1.0, 0.0, True),
                                          ("random weights, random tiles", 1.0, 1.0, False)):
    pol = PackedPolicy({k: v * (bias_scale if k.endswith("bias") else scale) for k, v in sd.items()}, PROBLEM_TSP, dev)
    eng.tile.copy_(tile0)
    if zero_tile:
        eng.tile.zero_()
    for _ in range(20): eng.forward(pol, want_logits=False, want_prob=False)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): eng.forward(pol, want_logits=False, want_prob=False)
    b.record(); torch.cuda.synchronize()
    print(f"{tag:64s} {a.elapsed_time(b) / iters * 1e3:.1f} us per launch")
