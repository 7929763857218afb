"""The benchmark's own sizes (1000 nodes, 512 instances per card), checked through properties that do not need the
reference's outputs: tours are permutations, routes respect the capacity, the device's lengths equal a host
recomputation, neighbourhoods sampled along the way equal the plain-C oracle's bit for bit, are ascending and
unvisited, and a second rollout repeats the first."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

B, N = 512, 1000


def _visited_u8(words: torch.Tensor, n: int) -> np.ndarray:
    """[R][W] int64 bit masks (bit j%64 of word j/64) -> [R][n] uint8."""
    b = np.unpackbits(words.cpu().numpy().view(np.uint8), axis=1, bitorder="little")
    return np.ascontiguousarray(b[:, :n])


def test_tsp1000_b512_properties(device):
    from dgl_amd._lib import PROBLEM_TSP
    from dgl_amd.engine import PackedPolicy, TspEngine
    from dgl_amd.weights import random_state_dict
    from oracle import cref
    lib = cref.load()
    knn = 30
    xy = torch.rand(B, N, 2, generator=torch.Generator().manual_seed(789))
    eng = TspEngine(xy.to(device), 1, knn)
    pol = PackedPolicy(random_state_dict("TSP", 789), PROBLEM_TSP, device)
    start = torch.full((B,), 17, dtype=torch.int32, device=device)
    xy_np = xy.numpy()

    eng.reset()
    eng.step(start)
    while eng.t < N:
        t = eng.t
        if t in (1, 400, 900, N - 10):
            k = eng.k
            idx = eng.idx.cpu().numpy()
            vis = _visited_u8(eng.state["visited"], N)
            tour = eng.state["tour"].cpu().numpy()
            assert (vis.sum(1) == t).all()
            sel = idx[:, :k]
            assert (np.diff(sel, axis=1) > 0).all() and (idx[:, k:] == -1).all()           # ascending, padded with -1
            assert not np.take_along_axis(vis, sel, 1).any()                               # unvisited only
            for r in (0, 100, 255, 511):
                ref = np.full(knn, -1, np.int32)
                got_k = lib.dglref_tsp_knn(np.ascontiguousarray(xy_np[r]), N, int(tour[r, t - 1]), vis[r], knn, ref)
                assert got_k == k and np.array_equal(idx[r], ref), (t, r)
        eng.forward(pol, want_logits=False, want_prob=False)
        eng.step(eng.selected)
    length = eng.tour_length().cpu().numpy().copy()
    tours = eng.state["tour"].cpu().numpy().copy()
    assert (np.sort(tours, axis=1) == np.arange(N)).all()                                  # permutations
    p = np.take_along_axis(xy_np.astype(np.float64), tours[:, :, None].astype(np.int64), 1)
    host = np.sqrt(((p - np.roll(p, -1, axis=1)) ** 2).sum(-1)).sum(1)
    np.testing.assert_allclose(length, host, rtol=2e-6)
    # the same rollout again: same decisions
    length2 = eng.greedy_rollout(pol, start).cpu().numpy()
    assert np.array_equal(eng.state["tour"].cpu().numpy(), tours) and np.array_equal(length2, length)


def test_cvrp1000_b512_properties(device):
    from dgl_amd._lib import PROBLEM_CVRP
    from dgl_amd.engine import CvrpEngine, PackedPolicy
    from dgl_amd.weights import random_state_dict
    from oracle import cref
    lib = cref.load()
    knn = 15
    M = N + 1
    gen = torch.Generator().manual_seed(789)
    xy = torch.cat((torch.rand(B, 1, 2, generator=gen), torch.rand(B, N, 2, generator=gen)), 1)
    dem = torch.cat((torch.zeros(B, 1), torch.randint(1, 200, (B, N), generator=gen) / 500.0), 1)
    eng = CvrpEngine(xy.to(device), dem.to(device), 1, knn, knn)
    pol = PackedPolicy(random_state_dict("CVRP", 789), PROBLEM_CVRP, device)
    start = torch.full((B,), 17 + 2 + N, dtype=torch.int32, device=device)               # CVRP/CVRPTester.py:183
    xy_np, dem_np = xy.numpy(), dem.numpy()

    eng.reset()
    eng.step(start)
    while eng.t < N:
        t = eng.t
        if t in (1, 400, 900):
            kl, kd = eng.kl, eng.kd
            il, idp = eng.idx_last.cpu().numpy(), eng.idx_dep.cpu().numpy()
            vis = _visited_u8(eng.state["visited"], M)
            nodes = eng.state["nodes"].cpu().numpy()
            assert (np.diff(il[:, :kl], axis=1) > 0).all() and (np.diff(idp[:, :kd], axis=1) > 0).all()
            assert not np.take_along_axis(vis, il[:, :kl], 1).any() and not np.take_along_axis(vis, idp[:, :kd], 1).any()
            for r in (0, 255, 511):
                ril, rid = np.full(knn, -1, np.int32), np.full(knn, -1, np.int32)
                klc = ctypes.c_int(0)
                v = vis[r].copy()
                v[0] = 0                                                                   # the oracle excludes the depot itself
                kdd = lib.dglref_cvrp_knn(np.ascontiguousarray(xy_np[r]), N, int(nodes[r, t - 1]), v, knn, knn, ril, rid,
                                          ctypes.byref(klc))
                assert kdd == kd and klc.value == kl, (t, r)
                assert np.array_equal(il[r], ril) and np.array_equal(idp[r], rid), (t, r)
        eng.forward(pol, want_logits=False, want_prob=False)
        eng.step(eng.selected)
    length = eng.route_length().cpu().numpy().copy()
    nodes = eng.state["nodes"].cpu().numpy().astype(np.int64)
    flags = eng.state["flags"].cpu().numpy().astype(np.int64)
    assert (np.sort(nodes, axis=1) == np.arange(1, M)).all()                              # every customer once
    assert (flags[:, 0] == 1).all()
    # capacity of every sub-route (a flag starts a new one from the depot)
    seg = np.cumsum(flags, axis=1) - 1
    load = np.zeros((B, N), np.float64)
    np.add.at(load, (np.arange(B)[:, None], seg), np.take_along_axis(dem_np.astype(np.float64), nodes, 1))
    assert load.max() <= 1.0 + 1e-5
    # length with the detours through the depot (CVRP/CVRPEnv.py:951-993)
    p = np.take_along_axis(xy_np.astype(np.float64), nodes[:, :, None], 1)
    depot = xy_np[:, :1].astype(np.float64)
    prev = np.concatenate((depot, p[:, :-1]), 1)
    d = lambda a, b: np.sqrt(((a - b) ** 2).sum(-1))
    host = np.where(flags == 1, d(prev, depot) + d(depot, p), d(prev, p)).sum(1) + d(p[:, -1], depot[:, 0])
    np.testing.assert_allclose(length, host, rtol=2e-6)
    length2 = eng.greedy_rollout(pol, start).cpu().numpy()
    assert np.array_equal(eng.state["nodes"].cpu().numpy(), nodes) and np.array_equal(length2, length)
