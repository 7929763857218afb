"""CVRP step kernel: state update alone (no neighbourhood requested) vs the whole step, for R rows of N customers."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from dgl_amd import _lib
from dgl_amd._lib import check, ptr, stream_ptr
from dgl_amd.engine import CvrpEngine
R = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
knn, kd, iters = 15, 15, 30
dev = torch.device('cuda:0')
dem = torch.cat((torch.zeros(R, 1, device=dev), torch.randint(1, 10, (R, N), device=dev) / 250.0), 1)
eng = CvrpEngine(torch.rand(R, N + 1, 2, device=dev), dem, 1, knn, kd)
lib = _lib.load()
eng.reset()
acts = [torch.randint(1, N + 1, (R,), dtype=torch.int32, device=dev) for _ in range(2 * iters + 6)]
def run(full, base):
    st = eng.state["struct"]
    for i in range(iters + 3):
        if i == 3:
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
        eng.t += 1
        check(lib.dgl_cvrp_step(ctypes.byref(st), ptr(acts[base + i]), eng.t, knn, kd, ptr(eng.idx_last) if full else 0,
                                ptr(eng.idx_dep) if full else 0, ptr(eng.infeasible) if full else 0,
                                ptr(eng.tile) if full else 0, stream_ptr()), "step")
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters
for full in (False, True, False, True):
    ms = run(full, 0 if not full else iters + 3)
    # per-row figure with a demand read per node (the kernel itself reads the demand of the candidates only: 24 N)
    by = (28 * N + N / 8 + (8 * (knn + kd) if full else 0)) * R
    print(f"R={R} N={N} {'whole step' if full else 'state update only'}: {ms*1e3:.1f} us, {by/ms/1e6:.0f} GB/s")
