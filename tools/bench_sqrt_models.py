"""What the "mkl_avx512" square-root model costs: greedy rollouts (TSP / CVRP-1000, 512 rows) and the step kernel at 512 and
16 384 rows under both models."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from dgl_amd import sqrt_model
from dgl_amd._lib import PROBLEM_TSP, PROBLEM_CVRP
from dgl_amd.engine import PackedPolicy, TspEngine, CvrpEngine
from dgl_amd.weights import random_state_dict
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
N = 1000
def step_time(eng, rows, cvrp):
    sel = [torch.full((rows,), 100 + i + (0), dtype=torch.int32, device=dev) for i in range(12)]
    evs = []
    for i in range(12):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); eng.step(sel[i]); b.record(); evs.append((a, b))
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in evs[2:])
    return 1e3 * sum(ms) / len(ms)
for model in ("ieee", "mkl_avx512", "ieee", "mkl_avx512"):
    sqrt_model.use(model, dev)
    out = [model]
    for rows in (512, 16384):
        xy = torch.rand(rows, N, 2, generator=g).to(dev)
        eng = TspEngine(xy, 1, 30); eng.reset(); eng.step(torch.full((rows,), 7, dtype=torch.int32, device=dev))
        out.append(f"tsp step {rows}: {step_time(eng, rows, False):.1f} us")
        cxy = torch.rand(rows, N + 1, 2, generator=g).to(dev)
        dem = torch.cat((torch.zeros(rows, 1), torch.randint(1, 200, (rows, N), generator=g) / 500.0), 1).to(dev)
        ce = CvrpEngine(cxy, dem, 1, 15, 15); ce.reset(); ce.step(torch.full((rows,), 7 + 2 + N, dtype=torch.int32, device=dev))
        out.append(f"cvrp step {rows}: {step_time(ce, rows, True):.1f} us")
        if rows == 512:
            pol = PackedPolicy(random_state_dict('TSP', 1), PROBLEM_TSP, dev)
            start = torch.zeros(rows, dtype=torch.int32, device=dev)
            eng.greedy_rollout(pol, start); torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(3): eng.greedy_rollout(pol, start)
            torch.cuda.synchronize(); out.append(f"tsp rollout {(time.perf_counter() - t0) / 3 * 1e3:.1f} ms")
            cpol = PackedPolicy(random_state_dict('CVRP', 1), PROBLEM_CVRP, dev)
            cstart = torch.full((rows,), 5 + 2 + N, dtype=torch.int32, device=dev)
            ce.greedy_rollout(cpol, cstart); torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(3): ce.greedy_rollout(cpol, cstart)
            torch.cuda.synchronize(); out.append(f"cvrp rollout {(time.perf_counter() - t0) / 3 * 1e3:.1f} ms")
            t0 = time.perf_counter(); eng.reset(); torch.cuda.synchronize(); out.append(f"tsp reset {(time.perf_counter() - t0) * 1e3:.2f} ms")
    print("; ".join(out), flush=True)
