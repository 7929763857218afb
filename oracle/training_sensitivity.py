"""How far the reference's own training trajectory moves when its gradients change in the last bits.

Test infrastructure (CPU, PyTorch): the oracle repeats the Adam steps of the self-improvement capture
(tests/golden/selfimp_tsp_n40_b6.npz: 39 teacher-forced decode steps, CVRP/CVRPTrainer.py:232-282 and its TSP twin) twice -
once as it is, once with gaussian noise of 3e-7 of the tensor's root mean square added to every gradient entry: the error
of a float32 sum is relative to the size of its terms, not to the size of the sum, and two summation orders of the same
sums differ by this much (two builds of the device kernels: 2.8e-7 of the gradient's Frobenius norm).
Adam's first steps move every entry by +-lr whatever the size of its gradient, so entries whose gradient is rounding noise
take their sign from that noise; the teacher probabilities of the following steps then differ.  The numbers printed here
are what the tolerance of tests/test_gpu_selfimprove.py is set by.

    python oracle/training_sensitivity.py [seeds]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.dgl_oracle import TSPOracle, enable_grad, train_forward_backward   # noqa: E402
from dgl_amd.weights import load_checkpoint                                    # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def trajectory(noise, seed):
    g = np.load(os.path.join(GOLDEN, "selfimp_tsp_n40_b6.npz"))
    w = load_checkpoint(os.path.join(GOLDEN, "ckpt_tsp_nn.pt"))
    xy = torch.from_numpy(g["train/aug_xy"])
    sol = torch.from_numpy(g["chunk1/aug_solution"])
    orc = TSPOracle(w, knn=30)
    orc.w = enable_grad(w)
    orc.reset(xy)
    orc.step(sol[:, 0].clone())
    opt = torch.optim.Adam(list(orc.w.values()), lr=1e-4, weight_decay=1e-6)
    gen = torch.Generator().manual_seed(seed)
    probs = []
    for t in range(1, sol.shape[1]):
        opt.zero_grad()
        p, _ = train_forward_backward(orc, sol[:, t])
        probs.append(p.detach().numpy().copy())
        if noise:
            for v in orc.w.values():
                if v.grad is not None:
                    v.grad.add_(noise * float(v.grad.pow(2).mean().sqrt()) * torch.randn(v.grad.shape, generator=gen))
        opt.step()
        with torch.no_grad():
            orc.step(sol[:, t].clone())
    return np.stack(probs)                       # [steps][rows]


def sensitivity(seeds=3, noise=3e-7):
    base = trajectory(0.0, 0)
    likely = base >= 1e-3
    out = []
    for s in range(seeds):
        rel = np.abs(trajectory(noise, s) - base) / base
        early = min(12, rel.shape[0] // 3)
        out.append((rel[:early][likely[:early]].max(), rel[:early].max(), rel[likely].max(), rel.max()))
    return out


if __name__ == "__main__":
    for i, r in enumerate(sensitivity(int(sys.argv[1]) if len(sys.argv) > 1 else 3)):
        print(f"seed {i}: first 12 steps: likely tokens {r[0]:.2e}, all {r[1]:.2e}; all steps: likely {r[2]:.2e}, all {r[3]:.2e}")
