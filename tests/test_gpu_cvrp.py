"""GPU parity tests of the CVRP path: HIP kernels (through the C ABI) vs the golden vectors
captured from the reference and vs the plain-C oracle."""
import ctypes

import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


def _engine(case, device, rpi=1):
    from dgl_amd.engine import CvrpEngine
    xy = np.concatenate([case["depot_xy"][:, None, :], case["node_xy"]], axis=1)
    dem = np.concatenate([np.zeros((xy.shape[0], 1), np.float32), case["demand"]], axis=1)
    return CvrpEngine(torch.from_numpy(xy).to(device), torch.from_numpy(dem).to(device), rpi,
                      int(case["meta/knn"]), int(case["meta/depot_knn"]))


def _actions(case):
    N = case["node_xy"].shape[1]
    return case["nodes"].astype(np.int32) + (N + 1) * case["flags"].astype(np.int32)


@pytest.mark.parametrize("name", ["cvrp_n60_b4", "cvrp_n100_b8", "cvrp_n120_b2_k20", "cvrp_n20_b4"])
def test_reset_matches_golden(name, device):
    case, _ = load_golden(name)
    eng = _engine(case, device)
    eng.reset()
    np.testing.assert_allclose(eng.state["avg"].cpu().numpy(), case["avg0"], rtol=3e-7)
    np.testing.assert_allclose(eng.state["sd"].cpu().numpy(), case["std0"], rtol=3e-7)
    assert (eng.state["capacity"].cpu().numpy() == 1.0).all()


@pytest.mark.parametrize("name", ["cvrp_n60_b4", "cvrp_n100_b8", "cvrp_n1000_b2", "cvrp_n100_b64_nn", "cvrp_n120_b2_k20",
                                  "cvrp_n200_b4_k50_nn", "cvrp_n20_b4", "cvrp_n20_b64_nn"])
def test_teacher_forced_steps_match_golden(name, device):
    """Drive the state with the reference's own route; at every step: both index sets bit-exact, capacities
    bit-exact, logits within 1e-3 of the row's largest |logit| while more than 8 nodes are left, and for ALL steps
    (the last 8 included) the tile columns that do not come from the avg / sd recurrences within 2e-5 and the greedy
    pick equal to the reference's action wherever the reference's own decision margin exceeds 2e-3 of the row's
    largest |logit|.  Rows of more than 32 tokens (knn = depot_knn = 20 / 50) run on the general path."""
    from dgl_amd._lib import PROBLEM_CVRP, TILE_STRIDE
    from dgl_amd.engine import PackedPolicy
    case, weights = load_golden(name)
    eng = _engine(case, device)
    pol = PackedPolicy(weights, PROBLEM_CVRP, device)
    actions = _actions(case)
    act = torch.from_numpy(actions).to(device)
    N = eng.N
    eng.reset()
    max_logit_err = max_tile_err = max_cap_err = tail_logit_err = max_exact_col_err = 0.0
    tile_errs = []
    agree = total = decided = decided_same = tail_decided = tail_same = 0
    for t in range(1, N):
        eng.step(act[:, t - 1].contiguous())
        kl, kd = eng.kl, eng.kd
        il, idp = eng.idx_last.cpu().numpy(), eng.idx_dep.cpu().numpy()
        assert np.array_equal(il[:, :kl], case["idx_last"][t - 1][:, :kl]), t
        assert np.array_equal(idp[:, :kd], case["idx_dep"][t - 1][:, :kd]), t
        assert (il[:, kl:] == -1).all() and (idp[:, kd:] == -1).all()
        eng.forward(pol)
        n = kl + kd + 2
        tail = N - t <= 8
        sel = eng.selected.cpu().numpy()
        same = sel == actions[:, t]
        agree += int(same.sum())
        total += eng.R
        if "capacity" in case:
            cap = eng.state["capacity"].cpu().numpy()
            max_cap_err = max(max_cap_err, float(np.abs(cap - case["capacity"][t - 1]).max()))
        if "logits" in case:
            # the reference's decoder orders tokens [last', knn.., depot', depot-knn..]; ours keeps
            # the encoder order [knn.., last, depot-knn.., depot]
            g = case["logits"][t - 1][:, :n]
            g_enc = np.concatenate([g[:, 1:kl + 1], g[:, 0:1], g[:, kl + 2:n], g[:, kl + 1:kl + 2]], axis=1)
            logits = eng.logits.cpu().numpy()[:, :n]
            scale = np.abs(g_enc).max(axis=1, keepdims=True)
            err = float((np.abs(logits - g_enc) / scale).max())
            # The avg/sd recurrences amplify a 1-ulp difference by ~N/u (u nodes left), and with <= 3
            # nodes left the running std is analytically ~0, i.e. the reference's own value is rounding
            # noise that min-max normalisation over 2-4 tokens blows up to 0/1.  The 1e-3 bound is
            # therefore asserted while more than 8 nodes are left; for the tail see the decision check below.
            if not tail:
                max_logit_err = max(max_logit_err, err)
            else:
                tail_logit_err = max(tail_logit_err, err)
            # decision margin of the reference: gap between its two most probable actions in logit units
            # (log p differences are logit differences), relative to the row's largest |logit|
            p = np.sort(case["p"][t - 1][:, :kl + kd].astype(np.float64), axis=1)
            if p.shape[1] >= 2:
                margin = (np.log(p[:, -1]) - np.log(np.maximum(p[:, -2], 1e-300))) / scale[:, 0]
                clear = margin > 2e-3
            else:
                clear = np.ones(eng.R, bool)
            decided += int(clear.sum())
            decided_same += int((clear & same).sum())
            if tail:
                tail_decided += int(clear.sum())
                tail_same += int((clear & same).sum())
        if "tile" in case:
            tile = eng.tile.cpu().numpy().reshape(eng.R, -1, TILE_STRIDE)[:, :n, :9]
            gt = case["tile"][t - 1][:, :n * 9].reshape(eng.R, n, 9)
            exact_cols = [0, 1, 2, 3, 6, 7, 8]              # x, y, demand, d, sin, cos, group flag (not avg, sd)
            max_exact_col_err = max(max_exact_col_err, float(np.abs(tile[:, :, exact_cols] - gt[:, :, exact_cols]).max()))
            if not tail:
                max_tile_err = max(max_tile_err, float(np.abs(tile - gt).max()))
                tile_errs.append(np.abs(tile - gt).ravel())
    print(f"{name}: max rel logit err {max_logit_err:.2e} (last 8 steps {tail_logit_err:.2e}), max tile err {max_tile_err:.2e} "
          f"(columns without avg/sd, all steps: {max_exact_col_err:.2e}), capacity err {max_cap_err:.1e}, greedy pick agreement "
          f"{agree}/{total}; where the reference's margin > 2e-3: {decided_same}/{decided} (last 8 steps {tail_same}/{tail_decided})")
    assert max_logit_err < 1e-3
    assert max_cap_err == 0.0
    if "logits" in case:
        assert decided_same == decided, "a greedy pick differs where the reference's decision margin exceeds 2e-3"
        # the tail's logit error is bounded by what min-max normalisation can do to a noise-level sd column: one
        # column of 0/1 flips, i.e. O(1) of the row's scale; it is recorded here and must stay below 2
        assert tail_logit_err < 2.0, tail_logit_err
    if tile_errs:
        errs = np.concatenate(tile_errs)
        assert np.quantile(errs, 0.999) < 2e-5
        assert max_tile_err < 5e-3
        assert max_exact_col_err < 2e-5


@pytest.mark.parametrize("name", ["cvrp_n60_b4", "cvrp_n100_b8", "cvrp_n20_b4"])
def test_state_data_matches_golden(name, device):
    case, _ = load_golden(name)
    eng = _engine(case, device)
    act = torch.from_numpy(_actions(case)).to(device)
    eng.reset()
    snaps = sorted(int(k[4:].split("/")[0]) for k in case if k.startswith("snap") and k.endswith("/data"))
    for t in range(1, eng.N):
        eng.step(act[:, t - 1].contiguous())
        if t in snaps:
            data = eng.data().cpu().numpy()
            g = case[f"snap{t}/data"]
            tol = 4e-7 * eng.M / (eng.M - t)
            np.testing.assert_allclose(data[:, :, :4], g[:, :, :4], rtol=2e-7, atol=1e-7)
            np.testing.assert_allclose(data[:, :, 4:6], g[:, :, 4:6], rtol=tol, atol=tol)
            np.testing.assert_allclose(data[:, :, 6:], g[:, :, 6:], rtol=1e-5, atol=1e-6)
            # the capacity snapshot is taken after env.step, i.e. before the model nudges it
            gcap = case[f"snap{t}/capacity"]
            cap = eng.state["capacity"].cpu().numpy()
            assert np.all((cap == gcap) | ((gcap < 1e-5) & (np.abs(cap - gcap - 1e-7) < 1e-9)))


@pytest.mark.parametrize("N,knn,kd,until", [(150, 15, 20, None), (700, 15, 20, 60), (2500, 15, 15, 40)])
def test_recurrences_bit_exact_vs_c_oracle(device, ieee_roots, N, knn, kd, until):
    """Synthetic instance, random feasible actions: state arithmetic and both neighbourhoods must be
    bit-identical to the plain-C oracle.  700 nodes: four waves share a row; 2500: they also share the
    compaction of the candidates."""
    _recurrences_vs_c_oracle(device, N, knn, kd, until)


@pytest.mark.parametrize("N,knn,kd,until", [(25, 8, 8, None), (150, 15, 20, None), (700, 15, 20, 60), (1000, 50, 50, 30),
                                            (2500, 15, 15, 40), (5000, 100, 100, 10)])
def test_recurrences_under_the_reference_bits_model(device, ieee_roots, N, knn, kd, until):
    """The same comparison under the default square-root model, the C oracle looking MKL's estimate up in the recording
    of the instruction (26 points with the depot: the first size with matrix-product distances)."""
    _recurrences_vs_c_oracle(device, N, knn, kd, until, seed=N, model="mkl_avx512")


def _recurrences_vs_c_oracle(device, N, knn, kd, until, seed=3, B=2, model="ieee"):
    """model: the square-root model of both sides (dgl_amd/sqrt_model.py, oracle/cref.py::use_model); the device is left
    with correctly rounded roots selected."""
    from dgl_amd import sqrt_model
    from oracle import cref
    lib = cref.load()
    sqrt_model.use(model, device)
    cref.use_model(lib, model)
    try:
        _compare_recurrences(lib, device, N, knn, kd, until, seed, B)
    finally:
        cref.use_model(lib, "ieee")
        sqrt_model.use("ieee", device)


def _compare_recurrences(lib, device, N, knn, kd, until, seed, B):
    from dgl_amd.engine import CvrpEngine
    rng = np.random.default_rng(seed)
    M = N + 1
    xy = rng.random((B, M, 2), dtype=np.float32)
    dem = np.concatenate([np.zeros((B, 1), np.float32), (rng.integers(1, 200, (B, N)) / 500.0).astype(np.float32)], 1)
    eng = CvrpEngine(torch.from_numpy(xy).to(device), torch.from_numpy(dem).to(device), 1, knn, kd)
    eng.reset()
    avg, sd = eng.state["avg"].cpu().numpy().copy(), eng.state["sd"].cpu().numpy().copy()
    cap = np.ones(B, np.float32)
    visited = np.zeros((B, M), np.uint8)
    order = np.stack([rng.permutation(N) + 1 for _ in range(B)])
    for t in range(1, N if until is None else until + 1):
        acts = np.zeros(B, np.int32)
        for b in range(B):
            node = int(order[b, t - 1])
            via = t == 1 or cap[b] + 1e-6 < dem[b, node]
            acts[b] = node + (M if via else 0)
        eng.step(torch.from_numpy(acts).to(device))
        gavg, gsd = eng.state["avg"].cpu().numpy(), eng.state["sd"].cpu().numpy()
        gcap = eng.state["capacity"].cpu().numpy()
        il, idp = eng.idx_last.cpu().numpy(), eng.idx_dep.cpu().numpy()
        inf = eng.infeasible.cpu().numpy()
        for b in range(B):
            c = ctypes.c_float(cap[b])
            infeas = np.zeros(M, np.uint8)
            flag = ctypes.c_int(0)
            node = lib.dglref_cvrp_apply(N, int(acts[b]), dem[b], ctypes.byref(c), infeas, ctypes.byref(flag))
            cap[b] = c.value
            visited[b, node] = 1
            d = np.empty(M, np.float32)
            lib.dglref_step_features(np.ascontiguousarray(xy[b]), M, node, M - t, 1, avg[b], sd[b], d)
            ril, rid = np.full(knn, -1, np.int32), np.full(kd, -1, np.int32)
            klc = ctypes.c_int(0)
            kdd = lib.dglref_cvrp_knn(np.ascontiguousarray(xy[b]), N, node, visited[b], knn, kd, ril, rid, ctypes.byref(klc))
            assert np.array_equal(il[b], ril) and np.array_equal(idp[b], rid), (t, b)
            assert np.array_equal(inf[b, :klc.value], infeas[ril[:klc.value]]), (t, b)
            if cap[b] < 1e-5:
                cap[b] = np.float32(cap[b] + np.float32(1e-7))
        assert np.array_equal(gavg, avg) and np.array_equal(gsd, sd), t
        assert np.array_equal(gcap, cap), t
        avg, sd = gavg.copy(), gsd.copy()


@pytest.mark.parametrize("name,min_identical", [("cvrp_n100_b64_nn", 0.9), ("cvrp_n1000_b4_nn", 0.5),
                                                ("cvrp_n100_b256_nn_tours", 0.99),
                                                ("cvrp_n1000_b128_nn_tours", 0.99),
                                                ("cvrp_n1000_b512_nn_tours", 0.99),
                                                ("cvrp_n1000_b128_clustered_nn_tours", 0.95),
                                                ("cvrp_n20_b64_nn", 0.98), ("cvrp_n24_b64_nn_tours", 0.98),
                                                ("cvrp_n25_b64_nn_tours", 0.98),
                                                ("cvrp_n1000_b64_k50_nn_tours", 0.98),
                                                ("cvrp_n5000_b2_k100_nn_tours", 1.0)])     # configuration 5's size: 202 tokens per row    # 102 tokens per row (CVRPLIB neighbourhoods)
def test_greedy_routes_identical_trained_checkpoint(name, min_identical, device):
    """Greedy routes against the reference's own, trained checkpoint, under the engines' default square-root model
    (the reference's torch build, dgl_amd/sqrt_model.py): >= 99 % on every uniform population, the 512 instances of the
    headline size included (507/512; what is left are exact distance ties at the edge of a neighbourhood, which the
    reference's unstable argsort resolves its own way - 32 of the 512 trajectories pass one - and decisions whose two
    best logits differ by less than the policy arithmetic's 1e-5).
    The fixtures that carry a second opinion (oracle/make_ieee_tours.py: the reference's algorithm with correctly
    rounded square roots and a stable argsort) are then decoded once more with correctly rounded roots: that run must
    be reproduced on >= 99 %, while the reference's own run shares only 503 of 512 routes with it - its torch build's
    float sqrt is one ulp low on 0.6 % of inputs, the running standard deviations amplify last bits by N / #unvisited
    and the min-max normalisation by another hundredfold - so about 98 % is what correctly rounded roots can share
    with the reference's run (tools/dbg_mismatch*.py locate every differing decision)."""
    from dgl_amd._lib import PROBLEM_CVRP
    from dgl_amd.engine import PackedPolicy
    case, weights = load_golden(name)
    eng = _engine(case, device)
    pol = PackedPolicy(weights, PROBLEM_CVRP, device)
    start = torch.from_numpy(case["start"]).to(device=device, dtype=torch.int32).repeat(eng.B) + 2 + eng.N
    length = eng.greedy_rollout(pol, start).cpu().numpy()
    nodes, flags = eng.state["nodes"].cpu().numpy(), eng.state["flags"].cpu().numpy()
    for r in range(min(eng.R, 16)):
        assert sorted(nodes[r].tolist()) == list(range(1, eng.N + 1))
    same = (nodes == case["nodes"]).all(axis=1) & (flags == case["flags"]).all(axis=1)
    frac = same.mean()
    rel = abs(length.mean() - case["length"].mean()) / case["length"].mean()
    print(f"{name}: identical routes {int(same.sum())}/{eng.R} ({100 * frac:.2f} %), mean length "
          f"{length.mean():.6f} vs {case['length'].mean():.6f} (rel diff {rel:.2e})")
    assert frac >= min_identical
    assert rel < 1e-4
    assert np.array_equal(length[same], case["length"][same]), "lengths of identical routes differ from the reference's floats"
    if "ieee/nodes" in case:
        from dgl_amd import sqrt_model
        from test_gpu_tsp import _second_opinion
        sqrt_model.use("ieee", device)
        try:
            eng.greedy_rollout(pol, start)
        finally:
            sqrt_model.use(sqrt_model.default_model(), device)
        nodes, flags = eng.state["nodes"].cpu().numpy(), eng.state["flags"].cpu().numpy()
        got = nodes.astype(np.int32) + (eng.N + 1) * flags.astype(np.int32)
        ieee = case["ieee/nodes"].astype(np.int32) + (eng.N + 1) * case["ieee/flags"].astype(np.int32)
        ref = case["nodes"].astype(np.int32) + (eng.N + 1) * case["flags"].astype(np.int32)
        _second_opinion(name, got, ieee, ref, case["ref_tie"])


@pytest.mark.parametrize("name", ["cvrp_n1000_b128_nn_tours", "cvrp_n1000_b128_clustered_nn_tours"])
def test_route_mismatches_attributed_to_reset_moments(name, device, ieee_roots):
    """The CVRP counterpart of test_tour_mismatches_attributed_to_reset_moments (tests/test_gpu_tsp.py): the population
    decoded from the device's own reset and from the reference's captured reset moments; identical routes for both."""
    from dgl_amd._lib import PROBLEM_CVRP
    from dgl_amd.engine import PackedPolicy
    case, weights = load_golden(name)
    eng = _engine(case, device)
    pol = PackedPolicy(weights, PROBLEM_CVRP, device)
    start = torch.from_numpy(case["start"]).to(device=device, dtype=torch.int32).repeat(eng.B) + 2 + eng.N
    same = {}
    for moments in ("own", "reference"):
        eng.reset()
        if moments == "reference":
            eng.state["avg"].copy_(torch.from_numpy(case["avg0"]).to(device))
            eng.state["sd"].copy_(torch.from_numpy(case["std0"]).to(device))
        eng.step(start)
        while eng.t < eng.N:
            eng.forward(pol, want_logits=False, want_prob=False)
            eng.step(eng.selected)
        nodes, flags = eng.state["nodes"].cpu().numpy(), eng.state["flags"].cpu().numpy()
        same[moments] = int(((nodes == case["nodes"]).all(axis=1) & (flags == case["flags"]).all(axis=1)).sum())
    print(f"{name}: identical routes {same['own']}/{eng.R} from the device's reset, {same['reference']}/{eng.R} from the "
          f"reference's reset moments")
    assert same["reference"] >= same["own"] - 2
