"""Gradients of one training decode step with the fused chains (DGL_FF_FUSED=1) and with separate launches (=0):
child processes on the same inputs, relative difference per parameter tensor.  The oracle's float64 autograd is the
third party: both are compared with it."""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "..")
OUT = os.path.join(ROOT, "gpurun_out")
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    sys.path.insert(0, ROOT)
    from dgl_amd.engine import CvrpEngine
    from dgl_amd.train import PolicyTrainer
    from dgl_amd.weights import random_state_dict
    from dgl_amd._lib import PROBLEM_CVRP
    R, N, k = 16, 300, 100
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    xy = torch.rand(R, N + 1, 2, generator=g).to(dev)
    dem = torch.cat((torch.zeros(R, 1), torch.randint(1, 10, (R, N), generator=g) / 50.0), 1).to(dev)
    eng = CvrpEngine(xy, dem, 1, k, k, demand_in_distance=True)
    tr = PolicyTrainer(random_state_dict("CVRP", 1), PROBLEM_CVRP, dev)
    eng.reset()
    eng.step(torch.full((R,), 5 + N + 1, dtype=torch.int32, device=dev))
    teacher = eng.idx_last[:, 0].clone()
    tr.zero_grad()
    prob = tr.forward_backward(eng, teacher)
    torch.cuda.synchronize()
    torch.save({"prob": prob.cpu(), "grads": tr.grads.cpu()}, os.path.join(OUT, f"ffgrads_{os.environ.get('DGL_FF_FUSED', '1')}.pt"))
    sys.exit(0)
import torch
for v in ("0", "1"):
    r = subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, DGL_FF_FUSED=v), capture_output=True, text=True)
    if r.returncode:
        print(r.stdout[-2000:], r.stderr[-2000:])
        sys.exit(1)
a, b = (torch.load(os.path.join(OUT, f"ffgrads_{v}.pt")) for v in ("0", "1"))
print("teacher probability: max relative difference", float(((a["prob"] - b["prob"]).abs() / a["prob"].abs().clamp_min(1e-30)).max()))
ga, gb = a["grads"].double(), b["grads"].double()
print("flat gradient: relative Frobenius difference", float((ga - gb).norm() / ga.norm()), "norm", float(ga.norm()))
n = ga.numel()
step = 4096
worst = sorted(((float((ga[i:i + step] - gb[i:i + step]).norm() / (ga[i:i + step].norm() + 1e-30)), i) for i in range(0, n, step)), reverse=True)[:8]
print("worst 4096-element windows (relative difference, offset):", [(f"{w:.2e}", i) for w, i in worst])
