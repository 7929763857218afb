"""TSP-10000 beam-search decode (large-n stress): wall time of one batch."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from dgl_amd._lib import PROBLEM_TSP
from dgl_amd.engine import PackedPolicy, TspEngine
from dgl_amd.weights import random_state_dict
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
beam = int(sys.argv[2]) if len(sys.argv) > 2 else 16
N = int(sys.argv[3]) if len(sys.argv) > 3 else 10000
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
xy = torch.rand(B, N, 2, generator=g).to(dev)
eng = TspEngine(xy, beam, 30)
pol = PackedPolicy(random_state_dict('TSP', 1), PROBLEM_TSP, dev)
start = torch.zeros(B * beam, dtype=torch.int32, device=dev)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    length = eng.beam_rollout(pol, start, beam) if beam > 1 else eng.greedy_rollout(pol, start)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"TSP-{N} B={B} beam={beam} (R={B*beam}): {dt:.2f} s per batch, {B/dt:.2f} instances/s, {dt/(N-1)*1e6:.0f} us per decode step, "
          f"best length mean {length.view(B, beam).min(1).values.mean().item():.3f}", flush=True)
