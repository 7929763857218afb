"""Randomised parity sweep on the GPU (beyond the fixed cases of tests/): for a given number of seconds draws sizes at
random and checks
  * the TSP step kernel (state recurrences, neighbourhoods) against the plain-C oracle, bit for bit, N = 2 ... 12000,
    knn = 1 ... 256, 1 ... 5 rows, whole tours for short instances and prefixes for long ones;
  * the CVRP step kernel likewise (both neighbourhoods, capacities, feasibility);
  * the one-launch policy kernel for 33 ... 128 tokens against the general path, TSP and CVRP layouts.
Every step-kernel case draws its square-root model too: correctly rounded roots, or the reference-bits model with the C
oracle looking MKL's estimate up in the recording of the instruction (oracle/cref.py::use_model).
usage: python tools/stress_parity.py [seconds] [seed]"""
import os, sys, time, types
import numpy as np, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from dgl_amd import sqrt_model
import test_gpu_knn as K
import test_gpu_cvrp as C
import test_gpu_policy as P

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dev = torch.device('cuda:0')
sqrt_model.use("ieee", dev)
rng = np.random.default_rng(seed)
from dgl_amd import _lib
from dgl_amd.engine import PackedPolicy
from dgl_amd.weights import load_checkpoint
wide_policy = PackedPolicy(load_checkpoint(os.path.join(ROOT, "tests", "golden", "ckpt_tsp_nn.pt")), _lib.PROBLEM_TSP, dev)
t0 = time.time()
done = {"tsp": 0, "cvrp": 0, "wide_tsp": 0, "wide_cvrp": 0, "reference_bits": 0}
last = t0
while time.time() - t0 < budget:
    kind = rng.choice(["tsp", "tsp", "cvrp", "wide_tsp", "wide_cvrp"])
    model = str(rng.choice(["ieee", "mkl_avx512"]))
    if kind == "tsp":
        N = int(rng.choice([rng.integers(2, 27), rng.integers(26, 200), rng.integers(200, 1200), rng.integers(1200, 3000),
                            rng.integers(3000, 12000)]))
        knn = int(rng.choice([1, 5, 15, 30, 31, 50, 64, 65, 100, 120, 200, 256]))
        B = int(rng.integers(1, 6))
        xy = rng.random((B, N, 2), dtype=np.float32)
        if rng.random() < 0.2:                                   # duplicates: exact ties
            xy[:, N // 2:] = xy[:, :N - N // 2]
        order = np.stack([rng.permutation(N) for _ in range(B)])
        until = None if N <= 300 else int(rng.integers(20, 80)) if N <= 3000 else int(rng.integers(6, 16))
        try:
            K._run_case(dev, xy, order, knn, check_until=until, model=model)
        except AssertionError as e:
            print("TSP MISMATCH", dict(N=N, knn=knn, B=B, until=until, model=model), str(e)[:300], flush=True); raise
    elif kind == "cvrp":
        N = int(rng.choice([rng.integers(2, 26), rng.integers(25, 200), rng.integers(200, 1500), rng.integers(1500, 6000)]))
        knn, kd = int(rng.choice([1, 8, 15, 30, 50, 100, 128])), int(rng.choice([1, 8, 15, 30, 50, 100, 128]))
        until = None if N <= 200 else int(rng.integers(20, 60)) if N <= 1500 else int(rng.integers(6, 16))
        try:
            C._recurrences_vs_c_oracle(dev, N, knn, kd, until, seed=int(rng.integers(1 << 30)), B=int(rng.integers(1, 4)), model=model)
        except AssertionError as e:
            print("CVRP MISMATCH", dict(N=N, knn=knn, kd=kd, until=until, model=model), str(e)[:300], flush=True); raise
    elif kind == "wide_tsp" and rng.random() < 0.5:
        knn = int(rng.integers(32, 128))
        P.test_tsp_wide_kernel_agrees_with_general_path(knn, str(rng.choice(["random", "trained"])), dev)
    elif kind == "wide_tsp":                                 # row counts from 1 up, instances barely larger than the neighbourhood
        knn, R = int(rng.integers(32, 128)), int(rng.choice([1, 2, 3, int(rng.integers(4, 300))]))
        N = int(rng.choice([knn + 2, knn + int(rng.integers(3, 60)), 400]))
        eng = P._tsp_engine(R, N, knn, dev, seed=int(rng.integers(1 << 30)))
        (la, pa, sa), (lb, pb, sb) = P._wide_vs_general(eng, wide_policy, knn + 1)
        rel = float((np.abs(la - lb) / np.abs(lb).max(axis=1, keepdims=True)).max())
        assert rel < 3e-5, ("WIDE MISMATCH", dict(R=R, N=N, knn=knn), rel)
        np.testing.assert_allclose(pa, pb, rtol=5e-4 + 2 * float(np.abs(la - lb).max()), atol=1e-7)
        assert (sa == sb).mean() >= 0.97 or R < 30, dict(R=R, N=N, knn=knn)
    else:
        kd = int(rng.integers(16, 64)); knn = int(rng.integers(8, kd + 1))      # the direct neighbourhood is never the wider one
        if knn + kd + 2 <= 32 or knn + kd + 2 > 128:
            continue
        P.test_cvrp_wide_kernel_agrees_with_general_path(knn, kd, dev)
    done[kind] += 1
    done["reference_bits"] += kind in ("tsp", "cvrp") and model == "mkl_avx512"
    if time.time() - last > 30:
        last = time.time()
        print(f"{time.time() - t0:.0f} s: {done}", flush=True)
print("sweep done, no mismatch:", done)
