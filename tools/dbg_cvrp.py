import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from conftest import load_golden
from test_gpu_cvrp import _engine, _actions
from dgl_amd._lib import PROBLEM_CVRP, TILE_STRIDE
from dgl_amd.engine import PackedPolicy
dev = torch.device('cuda:0')
case, weights = load_golden('cvrp_n100_b8')
eng = _engine(case, dev); pol = PackedPolicy(weights, PROBLEM_CVRP, dev)
act = torch.from_numpy(_actions(case)).to(dev)
eng.reset()
for t in range(1, eng.N):
    eng.step(act[:, t-1].contiguous())
    kl, kd = eng.kl, eng.kd; n = kl + kd + 2
    eng.forward(pol)
    g = case['logits'][t-1][:, :n]
    g_enc = np.concatenate([g[:, 1:kl+1], g[:, 0:1], g[:, kl+2:n], g[:, kl+1:kl+2]], axis=1)
    lg = eng.logits.cpu().numpy()[:, :n]
    err = np.abs(lg - g_enc) / np.abs(g_enc).max(axis=1, keepdims=True)
    if err.max() > 1e-3:
        r, tok = np.unravel_index(err.argmax(), err.shape)
        print('t', t, 'kl', kl, 'kd', kd, 'row', r, 'tok', tok, 'err', err.max(), 'cap', eng.state['capacity'].cpu().numpy()[r], 'golden cap', case['capacity'][t-1][r])
        print('  ours', lg[r].round(3)); print('  ref ', g_enc[r].round(3))
        break
