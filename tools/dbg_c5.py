"""Debug: CVRP-5000 training-reset distances (x, y, demand) against the oracle's torch.cdist, step by step."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from dgl_amd.engine import CvrpEngine
from oracle.dgl_oracle import CVRPOracle, random_weights
dev = torch.device("cuda:0")
Bc, Nc, knn = 2, int(sys.argv[1]) if len(sys.argv) > 1 else 5000, 100
train = (sys.argv[2] != "0") if len(sys.argv) > 2 else True
Mc = Nc + 1
gen = torch.Generator().manual_seed(55)
depot, nodes = torch.rand(Bc, 2, generator=gen), torch.rand(Bc, Nc, 2, generator=gen)
demand = torch.randint(1, 200, (Bc, Nc), generator=gen) / 500.0
orc = CVRPOracle(random_weights("CVRP", 55), knn, knn)
orc.reset(depot, nodes, demand, train=train, chain=True)
xy = torch.cat((depot[:, None, :], nodes), 1)
dem = torch.cat((torch.zeros(Bc, 1), demand), 1)
eng = CvrpEngine(xy.to(dev), dem.to(dev), 1, knn, knn, demand_in_distance=train)
eng.reset()
order = torch.stack([torch.randperm(Nc, generator=gen) + 1 for _ in range(Bc)])
with torch.no_grad():
    for t in range(1, 2501):
        act = order[:, t - 1] + (Mc if t == 1 else 0)
        eng.step(act.to(device=dev, dtype=torch.int32)); orc.step(act.clone())
        d = eng.data()[:, :, 3].cpu().numpy(); dr = orc.d_cur.numpy()
        bad = np.argwhere(d != dr)
        a, ar = eng.state["avg"].cpu().numpy(), orc.avg.numpy()
        if t > 5 and t % 250: continue
        print(t, "d mismatches", len(bad), "max |dd|", np.abs(d - dr).max(), "avg max rel", (np.abs(a - ar) / np.abs(ar)).max())
        for (b, j) in bad[:3]:
            print("   ", b, j, d[b, j], dr[b, j], "cur", int(act[b]) % Mc)
        rel = np.abs(a - ar) / np.abs(ar)
        w = np.argwhere(rel > 2e-6)
        vis = orc.nodes.numpy()
        print("    avg rel > 2e-6:", len(w), [(int(b), int(j), bool((vis[b] == j).any()), float(rel[b, j])) for b, j in w[:6]])
