import sys, numpy as np, torch
sys.path.insert(0, '.')
from dgl_amd.engine import TspEngine
from oracle import cref
lib = cref.load()
dev = torch.device('cuda:0')
rng = np.random.default_rng(1)
N = 200
xy = rng.random((1, N, 2), dtype=np.float32)
eng = TspEngine(torch.from_numpy(xy).to(dev), 1, 30)
eng.reset()
avg0 = eng.state['avg'].cpu().numpy().copy(); sd0 = eng.state['sd'].cpu().numpy().copy()
sel = torch.tensor([17], dtype=torch.int32, device=dev)
eng.step(sel)
data = eng.data().cpu().numpy()[0]
d = np.empty(N, np.float32); a = avg0[0].copy(); s = sd0[0].copy()
lib.dglref_step_features(xy[0], N, 17, N - 1, 0, a, s, d)
print('dcol mism', (data[:, 2] != d).sum())
ga = eng.state['avg'].cpu().numpy()[0]; gs = eng.state['sd'].cpu().numpy()[0]
print('avg mism', (ga != a).sum(), 'sd mism', (gs != s).sum())
bad = np.nonzero(ga != a)[0][:5]
for j in bad:
    print(j, 'a0', avg0[0, j], 'd', d[j], 'gpu', ga[j], 'c', a[j], 'np', np.float32(np.float32(np.float32(avg0[0, j] * np.float32(N)) - d[j]) / np.float32(N - 1)))
bad = np.nonzero(gs != s)[0][:5]
for j in bad:
    s0 = sd0[0, j]; e = np.float32(d[j] - avg0[0, j])
    item = np.float32(np.float32(np.float32(np.float32(s0 * s0) * np.float32(N)) - np.float32(e * e)) / np.float32(N - 1))
    print(j, 's0', s0, 'gpu', gs[j], 'c', s[j], 'np', np.sqrt(item))
