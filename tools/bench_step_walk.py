"""Step kernel on a walk through the neighbourhoods, as a decode makes it: every step visits a member of the list the
previous step produced (column `pick` of idx: 'near' = the member nearest to the current node, 'any' = a pseudo-random
member).  Prints the time per launch and a digest of the final state, lists and tiles (equal digests = equal bits).
  python tools/bench_step_walk.py [tsp|cvrp] R N steps [near|any]"""
import hashlib, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from dgl_amd.engine import CvrpEngine, TspEngine
prob = sys.argv[1] if len(sys.argv) > 1 else "tsp"
R = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
N = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 40
mode = sys.argv[5] if len(sys.argv) > 5 else "any"
knn = 30 if prob == "tsp" else 15
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(3)
if prob == "cvrp":
    xy = torch.rand(R, N + 1, 2, generator=g).to(dev)
    dem = torch.cat((torch.zeros(R, 1), torch.randint(1, 200, (R, N), generator=g) / 500.0), 1).to(dev)
    eng = CvrpEngine(xy, dem, 1, knn, knn)
    first = torch.full((R,), 7 + 2 + N, dtype=torch.int32, device=dev)
    lists = lambda: eng.idx_last
else:
    xy = torch.rand(R, N, 2, generator=g).to(dev)
    eng = TspEngine(xy, 1, knn)
    first = torch.full((R,), 7, dtype=torch.int32, device=dev)
    lists = lambda: eng.idx
eng.reset(); eng.step(first)
rows = torch.arange(R, device=dev)


def choose(i):
    idx = lists()
    if mode == "any":
        return idx[:, (7 * i + 3) % knn].contiguous()
    cur = eng.state["tour"][:, eng.t - 1].long() if prob == "tsp" else eng.state["nodes"][:, eng.t - 1].long()
    p = xy[rows[:, None], idx.long()]                                  # [R, k, 2]
    d = ((p - xy[rows, cur][:, None, :]) ** 2).sum(-1)
    return idx[rows, d.argmin(1)].contiguous()


sels = []
for i in range(3):
    eng.step(choose(i))
torch.cuda.synchronize()
tot = 0.0
for i in range(steps):
    s = choose(i + 3)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); eng.step(s); b.record(); torch.cuda.synchronize()
    tot += a.elapsed_time(b)
h = hashlib.sha256()
for tns in ([eng.idx, eng.tile] if prob == "tsp" else [eng.idx_last, eng.idx_dep, eng.tile, eng.infeasible]) + [eng.state["avg"], eng.state["sd"]]:
    h.update(tns.cpu().numpy().tobytes())
print(f"{prob} R={R} N={N} walk={mode}: {tot / steps * 1e3:.1f} us per launch (event pair included)  digest {h.hexdigest()[:16]}")
