import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from conftest import load_golden
import test_gpu_cvrp as C, test_gpu_tsp as T
from dgl_amd._lib import PROBLEM_CVRP, PROBLEM_TSP
from dgl_amd.engine import PackedPolicy
dev = torch.device('cuda:0')
case, weights = load_golden('cvrp_n1000_b2')
eng = C._engine(case, dev); pol = PackedPolicy(weights, PROBLEM_CVRP, dev)
act = torch.from_numpy(C._actions(case)).to(dev)
eng.reset(); prof = {}
for t in range(1, eng.N):
    eng.step(act[:, t-1].contiguous()); kl, kd = eng.kl, eng.kd; n = kl+kd+2
    eng.forward(pol)
    g = case['logits'][t-1][:, :n]
    g_enc = np.concatenate([g[:, 1:kl+1], g[:, 0:1], g[:, kl+2:n], g[:, kl+1:kl+2]], axis=1)
    lg = eng.logits.cpu().numpy()[:, :n]
    prof[eng.N - t] = float((np.abs(lg-g_enc)/np.abs(g_enc).max(axis=1, keepdims=True)).max())
us = sorted(prof)
for lo, hi in [(1,3),(4,8),(9,16),(17,32),(33,64),(65,128),(129,256),(257,512),(513,1000)]:
    print('cvrp u in', lo, hi, 'max err %.2e' % max(prof[u] for u in us if lo <= u <= hi))
case, weights = load_golden('tsp_n1000_b2')
eng = T._engine(case, dev); pol = PackedPolicy(weights, PROBLEM_TSP, dev)
tour = torch.from_numpy(case['tour']).to(device=dev, dtype=torch.int32)
eng.reset(); prof = {}
for t in range(1, eng.N):
    eng.step(tour[:, t-1].contiguous()); k = eng.k
    eng.forward(pol)
    g = case['logits'][t-1][:, :k+1]; lg = eng.logits.cpu().numpy()[:, :k+1]
    prof[eng.N - t] = float((np.abs(lg-g)/np.abs(g).max(axis=1, keepdims=True)).max())
us = sorted(prof)
for lo, hi in [(1,3),(4,8),(9,16),(17,32),(33,64),(65,128),(129,256),(257,512),(513,1000)]:
    print('tsp u in', lo, hi, 'max err %.2e' % max(prof[u] for u in us if lo <= u <= hi))
