"""Where do the greedy routes that differ from the reference's first leave it?  GPU half: decode a tours-only
population, and for every instance whose route differs replay the reference's actions up to the first differing
decision and keep what the device saw there (index sets, token tile, logits, capacity).
    python tools/dbg_mismatch.py cvrp_n1000_b512_nn_tours gpurun_out/mismatch_cvrp.npz
The CPU half (tools/dbg_mismatch_cpu.py) replays the same prefix through the oracle and compares."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
from conftest import load_golden
from dgl_amd import _lib
from dgl_amd.engine import CvrpEngine, TspEngine, PackedPolicy

name, out_path = sys.argv[1], sys.argv[2]
case, weights = load_golden(name)
dev = torch.device("cuda:0")
cvrp = "node_xy" in case
if cvrp:
    xy = np.concatenate([case["depot_xy"][:, None, :], case["node_xy"]], axis=1)
    dem = np.concatenate([np.zeros((xy.shape[0], 1), np.float32), case["demand"]], axis=1)
    N = case["node_xy"].shape[1]
    gold = case["nodes"].astype(np.int32) + (N + 1) * case["flags"].astype(np.int32)
    mk = lambda sel: CvrpEngine(torch.from_numpy(xy[sel]).to(dev), torch.from_numpy(dem[sel]).to(dev), 1,
                                int(case["meta/knn"]), int(case["meta/depot_knn"]))
    pol = PackedPolicy(weights, _lib.PROBLEM_CVRP, dev)
else:
    N = case["xy"].shape[1]
    gold = case["tour"].astype(np.int32)
    mk = lambda sel: TspEngine(torch.from_numpy(case["xy"][sel]).to(dev), 1, int(case["meta/knn"]))
    pol = PackedPolicy(weights, _lib.PROBLEM_TSP, dev)
B = gold.shape[0]
eng = mk(slice(None))
start = torch.from_numpy(gold[:, 0].copy()).to(dev)
eng.greedy_rollout(pol, start)
if cvrp:
    got = eng.state["nodes"].cpu().numpy() + (N + 1) * eng.state["flags"].cpu().numpy().astype(np.int32)
else:
    got = eng.state["tour"].cpu().numpy()
bad = np.nonzero((got != gold).any(axis=1))[0]
first = np.array([int(np.nonzero(got[b] != gold[b])[0][0]) for b in bad])
print(f"{name}: {len(bad)} of {B} differ; first differing decision at", dict(zip(bad.tolist(), first.tolist())))
rec = {"bad": bad, "first": first, "got_action": np.array([got[b, t] for b, t in zip(bad, first)]),
       "gold_action": np.array([gold[b, t] for b, t in zip(bad, first)])}
for b, t in zip(bad, first):
    e = mk(slice(b, b + 1))
    e.reset()
    g = torch.from_numpy(gold[b:b + 1].copy()).to(dev)
    for s in range(t):
        e.step(g[:, s].contiguous())
    e.forward(pol)
    torch.cuda.synchronize()
    rec[f"logits_{b}"] = e.logits.cpu().numpy()[0]
    rec[f"prob_{b}"] = e.prob.cpu().numpy()[0]
    rec[f"tile_{b}"] = e.tile.cpu().numpy()[0]
    rec[f"sel_{b}"] = e.selected.cpu().numpy()
    if cvrp:
        rec[f"idx_last_{b}"] = e.idx_last.cpu().numpy()[0]
        rec[f"idx_dep_{b}"] = e.idx_dep.cpu().numpy()[0]
        rec[f"cap_{b}"] = e.state["capacity"].cpu().numpy()
        rec[f"infeasible_{b}"] = e.infeasible.cpu().numpy()[0]
    else:
        rec[f"idx_{b}"] = e.idx.cpu().numpy()[0]
np.savez_compressed(out_path, **rec)
