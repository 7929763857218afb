"""Decode the f16 weight stream written by dgl_policy_pack and compare with the raw weights."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from dgl_amd._lib import PROBLEM_TSP
from dgl_amd.engine import PackedPolicy
from dgl_amd.weights import random_state_dict
dev = torch.device('cuda:0')
sd = random_state_dict('TSP', 1)
pol = PackedPolicy(sd, PROBLEM_TSP, dev)
torch.cuda.synchronize()
P = pol.packed.cpu()
D, EMB = 128, 12
PK_VEC0 = D * EMB + D
PKV = D + 512 + D
PK_STREAM0 = PK_VEC0 + 4 * PKV + 2 * D + D + 4
FRAG, BLK, ANCH = 256, 192, 32
WAVE = 4 * BLK + ANCH
NFR = 4 * WAVE + 16
S16 = PK_STREAM0 + NFR * FRAG
H = P[S16:S16 + NFR * FRAG].view(torch.float16).view(NFR, 64, 8).float()
lane = torch.arange(64)
def expect(W, ot, s):
    rows = 32 * ot + (lane & 31)
    cols = 16 * s + 8 * (lane >> 5)
    out = torch.stack([W[rows, cols + e] for e in range(8)], 1)
    hi = out.half().float()
    return hi, (out - hi).half().float()
names = ["encoder.layers.0."] + [f"decoder.layers.{i}." for i in range(3)]
bad = 0
for wv in range(4):
    for b in range(4):
        f = wv * WAVE + b * BLK + (ANCH if b > 0 else 0)
        pre = names[b]
        checks = []
        for s in range(8):
            for i, nm in enumerate(("Wq", "Wk", "Wv")):
                checks.append((sd[pre + nm + ".weight"], wv, s, f + 6 * s + 2 * i, nm))
            checks.append((sd[pre + "multi_head_combine.weight"], wv, s, f + 48 + 2 * s, "Wc"))
            for hc in range(4):
                checks.append((sd[pre + "feedForward.W1.weight"], 4 * hc + wv, s, f + 64 + 32 * hc + 2 * s, f"W1.{hc}"))
                checks.append((sd[pre + "feedForward.W2.weight"], wv, 8 * hc + s, f + 64 + 32 * hc + 16 + 2 * s, f"W2.{hc}"))
        for W, ot, s, fr, nm in checks:
            hi, lo = expect(W, ot, s)
            if not (torch.equal(H[fr], hi) and torch.equal(H[fr + 1], lo)):
                bad += 1
                if bad < 6: print("mismatch", wv, b, nm, s, fr)
print("bad fragments:", bad)
