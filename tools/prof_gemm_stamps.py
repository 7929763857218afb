"""Phase timing inside the training path's GEMM workgroups (library built with -DGEMM_STAMPS:
tools/build_variant.sh stamps wt -DGEMM_STAMPS; run with DGL_AMD_LIB=dgl_amd/lib/libdgl_amd_stamps.so)."""
import ctypes, os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from dgl_amd import _lib
from dgl_amd._lib import PROBLEM_CVRP
from dgl_amd.engine import CvrpEngine
from dgl_amd.train import PolicyTrainer
from dgl_amd.weights import random_state_dict
R, N, k = 64, 1000, 100
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
xy = torch.rand(R, N + 1, 2, generator=g).to(dev)
dem = torch.cat((torch.zeros(R, 1), torch.randint(1, 200, (R, N), generator=g) / 500.0), 1).to(dev)
eng = CvrpEngine(xy, dem, 1, k, k, demand_in_distance=True)
tr = PolicyTrainer(random_state_dict('CVRP', 1), PROBLEM_CVRP, dev)
eng.reset()
eng.step(torch.full((R,), 5 + N + 1, dtype=torch.int32, device=dev))
teacher = eng.idx_last[:, 0].clone()
for _ in range(3):
    tr.zero_grad(); tr.forward_backward(eng, teacher); tr.optimizer_step()
torch.cuda.synchronize()
lib = _lib.load()
lib.dgl_debug_gemm_stamps_clear.restype = ctypes.c_int
assert lib.dgl_debug_gemm_stamps_clear() == 0
tr.zero_grad(); tr.forward_backward(eng, teacher); tr.optimizer_step()
torch.cuda.synchronize()
WGS = 4096
buf = np.zeros(64 * WGS * 16, dtype=np.uint64)
lib.dgl_debug_gemm_stamps.restype = ctypes.c_int
assert lib.dgl_debug_gemm_stamps(buf.ctypes.data_as(ctypes.c_void_p)) == 0
buf = buf.reshape(64, WGS, 16).astype(np.int64)
names = ["start", "loads+stage0", "barrier", "rest of K", "epilogue", "amax"]
for L in range(64):
    b = buf[L]
    if (b[:, 13] > 0).any():         # an activation x weight launch
        s = b[b[:, 13] > 0]
        d = np.diff(s[:, :14], axis=1)
        nm = ["issue loads+scales", "wait A", "split+LDS", "barrier", "tile0 MFMA", "tile0 epilogue", "tile1 MFMA", "tile1 epi",
              "tile2 MFMA", "tile2 epi", "tile3 MFMA", "tile3 epi", "tail"]
        used = [i for i in range(13) if (s[:, i + 1] > 0).all() and (s[:, i] > 0).all()]
        print(f"launch slot {L} (activation x weight): {len(s)} workgroups, per-workgroup total median {np.median(s[:, 13] - s[:, 0]):.0f}")
        print("    " + "  ".join(f"{nm[i]} {np.median(d[:, i]):.0f}" for i in used))
        last = max(i for i in range(13) if (s[:, i] > 0).all())
        print(f"    last tile stamp -> end: {np.median(s[:, 13] - s[:, last]):.0f}")
        continue
    live = (b[:, 0] > 0) & (b[:, 6] > b[:, 0])
    if not live.any():
        continue
    s = b[live]
    t0 = s[:, 0].min()
    d = np.diff(s[:, :7], axis=1)
    print(f"launch slot {L}: {live.sum()} live workgroups, span {s[:, 6].max() - t0} ticks; start spread {np.percentile(s[:, 0] - t0, [50, 90, 100]).astype(int)}; "
          f"per-workgroup total median {np.median(s[:, 6] - s[:, 0]):.0f}")
    print("    " + "  ".join(f"{n} {np.median(d[:, i]):.0f}" for i, n in enumerate(names)))
    s2 = s[s[:, 14] > 0]
    if len(s2):
        d2 = np.diff(s2[:, 8:15], axis=1)
        n2 = ["wait loads", "split+LDS writes", "barrier", "issue next loads", "MFMA", "barrier"]
        print(f"    second stage ({len(s2)} workgroups): " + "  ".join(f"{n} {np.median(d2[:, i]):.0f}" for i, n in enumerate(n2)))
