"""Feed recorded token tiles straight to the policy kernels (both arithmetic modes, one row and many rows per launch)
and compare the logits with the oracle's for the same tiles.
    python tools/dbg_policy_tile.py gpurun_out/policy_cases_cvrp.npz"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
from conftest import load_golden
from dgl_amd import _lib
from dgl_amd.engine import CvrpEngine, PackedPolicy

rec = np.load(sys.argv[1])
_, weights = load_golden("cvrp_n1000_b512_nn_tours")
dev = torch.device("cuda:0")
pol = PackedPolicy(weights, _lib.PROBLEM_CVRP, dev)
N = 1000
g = torch.Generator().manual_seed(0)
for R in (1, 2, 512):
    xy = torch.rand(R, N + 1, 2, generator=g).to(dev)
    dem = torch.cat([torch.zeros(R, 1), torch.randint(1, 200, (R, N), generator=g) / 500.0], 1).to(dev)
    eng = CvrpEngine(xy, dem, 1, 15, 15)
    for b, t in zip(rec["bad"], rec["first"]):
        b, t = int(b), int(t)
        eng.t = t
        n = eng.kl + eng.kd + 2
        tile = torch.from_numpy(rec[f"tile_{b}"]).to(dev)
        eng.tile.zero_()
        eng.tile[:, :tile.numel()] = tile
        eng.idx_last.zero_(); eng.idx_dep.zero_(); eng.infeasible.zero_()
        res = []
        for mode in (_lib.ARITH_F32, _lib.ARITH_SPLIT_F16):
            eng.arithmetic = mode
            eng.logits.zero_()
            eng.forward(pol)
            torch.cuda.synchronize()
            lg = eng.logits.cpu().numpy()[:, :n]
            err = np.abs(lg - rec[f"oracle_{b}"][None, :n]).max(axis=1)
            res.append((err.max(), int(err.argmax()), np.abs(lg[0] - rec[f"devlogits_{b}"][:n]).max()))
        print(f"R={R} inst {b} step {t} n={n}: f32 kernel err {res[0][0]:.2e} (row {res[0][1]}), split-f16 err {res[1][0]:.2e} "
              f"(row {res[1][1]}); split vs the recorded device logits {res[1][2]:.2e}")
