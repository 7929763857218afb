"""Greedy TSP-1000 decode of 512 instances as ONE batch on one stream against TWO half-batches on two streams
(the step kernel and the launch gaps of one half under the policy kernel of the other).  Same instances, same tours."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from dgl_amd._lib import PROBLEM_TSP
from dgl_amd.engine import PackedPolicy, TspEngine
from dgl_amd.weights import random_state_dict
B, N = 512, 1000
parts = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(5)
xy = torch.rand(B, N, 2, generator=g).to(dev)
pol = PackedPolicy(random_state_dict('TSP', 1), PROBLEM_TSP, dev)
whole = TspEngine(xy, 1, 30)
cut = [B * i // parts for i in range(parts + 1)]
halves = [TspEngine(xy[cut[i]:cut[i + 1]].contiguous(), 1, 30) for i in range(parts)]
streams = [torch.cuda.Stream() for _ in range(parts)]
start_w = torch.full((B,), 5, dtype=torch.int32, device=dev)
starts = [torch.full((cut[i + 1] - cut[i],), 5, dtype=torch.int32, device=dev) for i in range(parts)]


def one_stream():
    return whole.greedy_rollout(pol, start_w)


def split_streams():
    for e, s, st in zip(halves, streams, starts):
        with torch.cuda.stream(s):
            e.reset(); e.step(st)
    for t in range(1, N):
        for e, s in zip(halves, streams):
            with torch.cuda.stream(s):
                e.forward(pol, want_logits=False, want_prob=False)
                e.step(e.selected)
    out = []
    for e, s in zip(halves, streams):
        with torch.cuda.stream(s):
            out.append(e.tour_length())
    for s in streams:
        torch.cuda.current_stream().wait_stream(s)
    return torch.cat(out)


os.environ["DGL_FUSED_DECODE"] = "0"
torch.cuda.synchronize()
for name, fn in (("one stream, 512 rows", one_stream), (f"{parts} streams", split_streams)) * 3:
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2): l = fn().clone()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 2
    print(f"{name}: {dt * 1e3:.2f} ms per rollout, {B / dt:.0f} instances/s, mean length {l.mean().item():.6f}", flush=True)
