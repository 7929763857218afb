"""Logits of the network cut after L blocks (libraries built with -DDGL_DEBUG_LAYERS=L) against the oracle's for the
same recorded tiles: finds the block in which a deviation first shows.
    python tools/dbg_policy_layers.py scratch/policy_cases_cvrp.npz L          (run with DGL_AMD_LIB=.../libdgl_amd_dbgL<L>.so)"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
from conftest import load_golden
from dgl_amd import _lib
from dgl_amd.engine import CvrpEngine, PackedPolicy

rec = np.load(sys.argv[1]); L = int(sys.argv[2])
_, weights = load_golden("cvrp_n1000_b512_nn_tours")
dev = torch.device("cuda:0")
pol = PackedPolicy(weights, _lib.PROBLEM_CVRP, dev)
N, R = 1000, 1
g = torch.Generator().manual_seed(0)
xy = torch.rand(R, N + 1, 2, generator=g).to(dev)
dem = torch.cat([torch.zeros(R, 1), torch.randint(1, 200, (R, N), generator=g) / 500.0], 1).to(dev)
eng = CvrpEngine(xy, dem, 1, 15, 15)
out = []
for b, t in zip(rec["bad"], rec["first"]):
    b, t = int(b), int(t)
    eng.t = t
    n = eng.kl + eng.kd + 2
    tile = torch.from_numpy(rec[f"tile_{b}"]).to(dev)
    eng.tile.zero_(); eng.tile[:, :tile.numel()] = tile
    eng.idx_last.zero_(); eng.idx_dep.zero_(); eng.infeasible.zero_()
    eng.logits.zero_()
    eng.forward(pol)
    torch.cuda.synchronize()
    lg = eng.logits.cpu().numpy()[0, :n]
    ref = rec[f"trunc_{b}"][L][:n]
    out.append(f"{b}@{t}: {np.abs(lg - ref).max():.1e}/{np.abs(ref).max():.0f}")
print(f"L={L}: " + "  ".join(out))
