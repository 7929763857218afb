"""CPU half of tools/dbg_mismatch.py: replay the reference's actions through the oracle up to each first differing
decision and compare what the device saw there with what the oracle sees.
    python tools/dbg_mismatch_cpu.py cvrp_n1000_b512_nn_tours gpurun_out/mismatch_cvrp.npz"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
from conftest import load_golden
from oracle.dgl_oracle import CVRPOracle, TSPOracle

name, rec_path = sys.argv[1], sys.argv[2]
chain = len(sys.argv) > 3 and sys.argv[3] == "chain"
case, weights = load_golden(name)
rec = np.load(rec_path)
cvrp = "node_xy" in case
for b, t in zip(rec["bad"], rec["first"]):
    b, t = int(b), int(t)
    if cvrp:
        N = case["node_xy"].shape[1]
        o = CVRPOracle(weights, int(case["meta/knn"]), int(case["meta/depot_knn"]))
        o.reset(torch.from_numpy(case["depot_xy"][b:b + 1]), torch.from_numpy(case["node_xy"][b:b + 1]),
                torch.from_numpy(case["demand"][b:b + 1]), chain=chain)
        gold = case["nodes"][b].astype(np.int64) + (N + 1) * case["flags"][b].astype(np.int64)
    else:
        N = case["xy"].shape[1]
        o = TSPOracle(weights, int(case["meta/knn"]))
        o.reset(torch.from_numpy(case["xy"][b:b + 1]), chain=chain)
        gold = case["tour"][b].astype(np.int64)
    for s in range(t):
        o.step(torch.tensor([gold[s]]))
    out = o.forward()
    if cvrp:
        kl = out["idx_last"].shape[1]; kd = out["idx_dep"].shape[1]
        n = kl + 1 + kd + 1
        lg = out["logits"][0].numpy()                     # decoder order: last', knn.., depot', depot-knn..
        enc = np.concatenate([lg[1:kl + 1], lg[0:1], lg[kl + 2:], lg[kl + 1:kl + 2]])   # the device's order
        same_idx = np.array_equal(rec[f"idx_last_{b}"][:kl], out["idx_last"][0].numpy()) and \
            np.array_equal(rec[f"idx_dep_{b}"][:kd], out["idx_dep"][0].numpy())
        F = 9
        cand = np.ones(n, bool); cand[kl] = False; cand[n - 1] = False
        cand[:kl] &= (out["knn_ninf"][0].numpy() == 0)
    else:
        k = out["idx"].shape[1]; n = k + 1
        enc = out["logits"][0].numpy()
        same_idx = np.array_equal(rec[f"idx_{b}"][:k], out["idx"][0].numpy())
        F = 7
        cand = np.ones(n, bool); cand[n - 1] = False
    tile_o = out["tile"][0].numpy()
    tile_g = rec[f"tile_{b}"].reshape(-1, 12)[:n, :F]
    dl = rec[f"logits_{b}"][:n]
    terr = np.abs(tile_g - tile_o)
    col_err = terr.max(axis=0)
    srt = np.sort(enc[cand])[::-1]
    margin = srt[0] - srt[1] if len(srt) > 1 else np.inf
    lerr = np.abs(dl - enc)[cand].max()
    print(f"inst {b} step {t} (left {N - t}): idx same {same_idx}; oracle margin {margin:.3e} (top {srt[0]:.3f}); "
          f"logit err max {lerr:.2e}; tile err by column {np.array2string(col_err, precision=1)}")
    # how well conditioned is this evaluation?  the same tile through the oracle's network in binary64
    o64 = (CVRPOracle if cvrp else TSPOracle)(weights)
    o64.w = {k_: v.double() for k_, v in o64.w.items()}
    if cvrp:
        l64 = o64.logits_from_tile(out["tile"].double(), kl)[0].numpy()
        l64 = np.concatenate([l64[1:kl + 1], l64[0:1], l64[kl + 2:], l64[kl + 1:kl + 2]])
    else:
        l64 = o64.logits_from_tile(out["tile"].double())[0].numpy()
    print(f"      vs binary64 network on the oracle's tile: oracle f32 err {np.abs(enc - l64)[cand].max():.2e}, "
          f"device err {np.abs(dl - l64)[cand].max():.2e}; f64 margin {np.sort(l64[cand])[::-1][0] - np.sort(l64[cand])[::-1][1]:.3e}")
