"""Debug aid: first-step gradient of the TSP training batch of the self-improvement golden, device vs oracle autograd."""
import os, sys, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from conftest import GOLDEN
from dgl_amd.engine import TspEngine
from dgl_amd.train import PolicyTrainer
from dgl_amd._lib import PROBLEM_TSP
from dgl_amd.weights import load_checkpoint
from oracle.dgl_oracle import TSPOracle, enable_grad, train_forward_backward
g = np.load(os.path.join(GOLDEN, "selfimp_tsp_n40_b6.npz"))
dev = torch.device('cuda')
w = load_checkpoint(os.path.join(GOLDEN, "ckpt_tsp_nn.pt"))
tr = PolicyTrainer(w, PROBLEM_TSP, dev)
xy = torch.from_numpy(g["train/aug_xy"])
eng = TspEngine(xy.to(dev), 1, 30)
sol = torch.from_numpy(g["chunk1/aug_solution"])
act = sol.to(device=dev, dtype=torch.int32)
eng.reset(); eng.step(act[:, 0].contiguous())
orc = TSPOracle(w, knn=30)
orc.w = enable_grad(w)
orc.reset(xy); orc.step(sol[:, 0].clone())
opt = torch.optim.Adam(list(orc.w.values()), lr=1e-4, weight_decay=1e-6)
names = list(orc.w.keys())
for t in range(1, 40):
    tr.zero_grad()
    prob = tr.forward_backward(eng, act[:, t].contiguous())
    p_ref, loss = train_forward_backward(orc, sol[:, t])
    gd = tr.gradient_dict()
    worst, wk, top = 0.0, "", 0.0
    for k, v in orc.w.items():
        if v.grad is None or k not in gd:
            continue
        a, b = gd[k].cpu().double(), v.grad.double()
        err = float((a - b).norm() / max(float(b.norm()), 1e-30))
        top = max(top, float(b.norm()))
        if err > worst:
            worst, wk = err, k
    tr.optimizer_step()
    opt.step()
    sd = tr.state_dict()
    dmax, dk = 0.0, ""
    for k, v in orc.w.items():
        d = float((sd[k].cpu() - v.detach()).abs().max())
        if d > dmax:
            dmax, dk = d, k
    clean = ["encoder.embedding.weight", "decoder.layers.0.feedForward.W1.bias", "decoder.layers.0.feedForward.W1.weight", "decoder.Linear_final.weight"]
    print("      clean-tensor max diffs / grad rel err:", " ".join(
        f"{float((sd[k].cpu() - orc.w[k].detach()).abs().max()):.1e}/{float((gd[k].cpu().double() - orc.w[k].grad.double()).norm() / max(float(orc.w[k].grad.double().norm()), 1e-30)):.1e}" for k in clean))
    print(f"step {t:2d} prob min {float(p_ref.min()):.2e} |dp| {float((prob.cpu() - p_ref).abs().max()):.1e} top|g| {top:.2e} worst grad rel err {worst:.1e} ({wk})  max param diff {dmax:.2e} ({dk})")
    eng.step(act[:, t].contiguous())
    with torch.no_grad():
        orc.step(sol[:, t].clone())
