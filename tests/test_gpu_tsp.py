"""GPU parity tests of the TSP path: HIP kernels (through the C ABI) vs the oracle and the
golden vectors captured from the reference."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


def _engine(case, device, knn=None, rpi=1):
    from dgl_amd.engine import TspEngine
    xy = torch.from_numpy(case["xy"]).to(device)
    return TspEngine(xy, rpi, int(case["meta/knn"]) if knn is None else knn)


@pytest.mark.parametrize("name", ["tsp_n50_b4", "tsp_n100_b8", "tsp_n1000_b2", "tsp_n20_b4"])
def test_reset_matches_golden(name, device):
    case, _ = load_golden(name)
    eng = _engine(case, device)
    eng.reset()
    avg, sd = eng.state["avg"].cpu().numpy(), eng.state["sd"].cpu().numpy()
    # reduction order differs from ATen's float cascade sum: a few ulp
    np.testing.assert_allclose(avg, case["avg0"], rtol=3e-7, atol=0)
    np.testing.assert_allclose(sd, case["std0"], rtol=3e-7, atol=0)


@pytest.mark.parametrize("name", ["tsp_n50_b4", "tsp_n100_b8", "tsp_n1000_b2", "tsp_n100_b64_nn", "tsp_n200_b4_k50_nn",
                                  "tsp_n20_b4", "tsp_n20_b64_nn"])
def test_teacher_forced_steps_match_golden(name, device):
    """Drive the state with the reference's own tour; at every step compare the kNN index
    sets (bit-exact), the normalised token tile and the logits (1e-3 relative)."""
    from dgl_amd.engine import PackedPolicy
    from dgl_amd._lib import PROBLEM_TSP, TILE_STRIDE
    case, weights = load_golden(name)
    eng = _engine(case, device)
    pol = PackedPolicy(weights, PROBLEM_TSP, device)
    tour = torch.from_numpy(case["tour"]).to(device=device, dtype=torch.int32)
    N, knn = eng.N, eng.knn
    eng.reset()
    bad_rows = 0
    max_logit_err = 0.0
    max_tile_err = 0.0
    tile_errs = []
    agree = total = 0
    for t in range(1, N):
        eng.step(tour[:, t - 1].contiguous())
        k = eng.k
        idx = eng.idx.cpu().numpy()
        gidx = case["idx"][t - 1]
        row_ok = (idx[:, :k] == gidx[:, :k]).all(axis=1)
        tie = case["tie"][t - 1]
        bad_rows += int((~row_ok & ~tie).sum())
        assert (idx[:, k:] == -1).all()
        eng.forward(pol)
        logits = eng.logits.cpu().numpy()[:, :k + 1]
        g = case["logits"][t - 1][:, :k + 1]
        ok = row_ok
        if ok.any():
            scale = np.abs(g[ok]).max(axis=1, keepdims=True)
            max_logit_err = max(max_logit_err, float((np.abs(logits[ok] - g[ok]) / scale).max()))
        if "tile" in case:
            tile = eng.tile.cpu().numpy().reshape(eng.R, knn + 1, TILE_STRIDE)[:, :k + 1, :7]
            gt = case["tile"][t - 1][:, :(k + 1) * 7].reshape(eng.R, k + 1, 7)
            max_tile_err = max(max_tile_err, float(np.abs(tile[ok] - gt[ok]).max()))
            tile_errs.append(np.abs(tile[ok] - gt[ok]).ravel())
        sel = eng.selected.cpu().numpy()
        agree += int((sel == case["tour"][:, t]).sum())
        total += eng.R
    print(f"{name}: kNN rows differing (non-tie) {bad_rows}, max rel logit err {max_logit_err:.2e}, "
          f"max tile err {max_tile_err:.2e}, greedy pick agreement {agree}/{total}")
    assert bad_rows == 0
    assert max_logit_err < 1e-3
    if "tile" in case:
        # tile entries are (v - min) / range over the neighbourhood: where the range of a running
        # feature is tiny, a 1-ulp difference in avg/sd (different summation order at reset, amplified
        # by N/u in the recurrences) is magnified; bound the bulk tightly and the tail loosely
        errs = np.concatenate(tile_errs)
        assert np.quantile(errs, 0.999) < 2e-5, np.quantile(errs, 0.999)
        assert max_tile_err < 5e-3


@pytest.mark.parametrize("name", ["tsp_n50_b4", "tsp_n100_b8", "tsp_n20_b4"])
def test_state_data_matches_golden(name, device):
    case, _ = load_golden(name)
    eng = _engine(case, device)
    tour = torch.from_numpy(case["tour"]).to(device=device, dtype=torch.int32)
    eng.reset()
    snaps = sorted(int(k[4:].split("/")[0]) for k in case if k.startswith("snap") and k.endswith("/data"))
    for t in range(1, eng.N):
        eng.step(tour[:, t - 1].contiguous())
        if t in snaps:
            data = eng.data().cpu().numpy()
            g = case[f"snap{t}/data"]
            u = eng.N - t
            # running features amplify a 1-ulp start difference by ~N/u
            tol = 4e-7 * eng.N / u
            np.testing.assert_allclose(data[:, :, :3], g[:, :, :3], rtol=2e-7, atol=1e-7)
            np.testing.assert_allclose(data[:, :, 3:5], g[:, :, 3:5], rtol=tol, atol=tol)
            np.testing.assert_allclose(data[:, :, 5:], g[:, :, 5:], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("name", ["tsp_n50_b4", "tsp_n100_b8", "tsp_n1000_b2", "tsp_n20_b4"])
def test_greedy_rollout_length(name, device):
    from dgl_amd.engine import PackedPolicy
    from dgl_amd._lib import PROBLEM_TSP
    case, weights = load_golden(name)
    eng = _engine(case, device)
    pol = PackedPolicy(weights, PROBLEM_TSP, device)
    start = torch.from_numpy(case["start"]).to(device=device, dtype=torch.int32).repeat(eng.B)
    length = eng.greedy_rollout(pol, start).cpu().numpy()
    tour = eng.state["tour"].cpu().numpy()
    for r in range(eng.R):
        assert sorted(tour[r].tolist()) == list(range(eng.N)), "not a permutation"
    same = (tour == case["tour"]).all(axis=1)
    print(f"{name}: identical tours {int(same.sum())}/{eng.R}; mean length {length.mean():.5f} vs "
          f"{case['length'].mean():.5f}")
    # lengths of identical tours: the reference's own floats (its torch build's roots, summed in torch.sum's order)
    assert np.array_equal(length[same], case["length"][same])


@pytest.mark.parametrize("name,min_identical", [("tsp_n100_b64_nn", 0.95), ("tsp_n1000_b8_nn", 0.75),
                                                ("tsp_n100_b512_nn_tours", 0.99),
                                                ("tsp_n1000_b128_nn_tours", 0.99),
                                                ("tsp_n1000_b512_nn_tours", 0.99),
                                                ("tsp_n1000_b128_clustered_nn_tours", 0.97),
                                                ("tsp_n20_b64_nn", 0.98), ("tsp_n25_b64_nn_tours", 0.98),
                                                ("tsp_n1000_b64_k50_nn_tours", 0.98),      # 51 tokens per row: the one-launch kernel for 33..128
                                                ("tsp_n10000_b2_nn_tours", 1.0)])          # configuration 4's size: 9 999 decisions per tour
def test_greedy_tours_identical_trained_checkpoint(name, min_identical, device):
    """With the trained checkpoint fixture (peaked action distributions) the greedy tours must be
    the reference's own: >= 99 % of instances on the populations (512 instances of 100 nodes, 128 and 512 instances of
    the headline size, 1000 nodes), mean length within 0.01 %, under the engines' default square-root model (the
    reference's torch build: dgl_amd/sqrt_model.py).  (The small cases only bound the damage one flipped arg-max can do
    to the ratio.)  Clustered instances (seven Gaussian clusters of sigma 0.04: the kind of set the reference evaluates
    beside the uniform one, TSP/test.py:131) hang on the last bit far more often - moving the reset's mean / std by
    one ulp changes a quarter of the reference's own tours there (oracle/rounding_sensitivity.py; none of the uniform
    ones) - and are held to 97 %.
    Fixtures that carry a second opinion are then decoded once more with correctly rounded roots (_second_opinion)."""
    from dgl_amd.engine import PackedPolicy
    from dgl_amd._lib import PROBLEM_TSP
    case, weights = load_golden(name)
    eng = _engine(case, device)
    pol = PackedPolicy(weights, PROBLEM_TSP, device)
    start = torch.from_numpy(case["start"]).to(device=device, dtype=torch.int32).repeat(eng.B)
    length = eng.greedy_rollout(pol, start).cpu().numpy()
    tour = eng.state["tour"].cpu().numpy()
    same = (tour == case["tour"]).all(axis=1)
    frac = same.mean()
    rel = abs(length.mean() - case["length"].mean()) / case["length"].mean()
    print(f"{name}: identical tours {int(same.sum())}/{eng.R} ({100 * frac:.2f} %), mean length "
          f"{length.mean():.6f} vs {case['length'].mean():.6f} (rel diff {rel:.2e})")
    assert frac >= min_identical
    assert rel < 1e-4
    assert np.array_equal(length[same], case["length"][same]), "lengths of identical tours differ from the reference's floats"
    if "ieee/tour" in case:
        from dgl_amd import sqrt_model
        sqrt_model.use("ieee", device)
        try:
            eng.greedy_rollout(pol, start)
        finally:
            sqrt_model.use(sqrt_model.default_model(), device)
        _second_opinion(name, eng.state["tour"].cpu().numpy(), case["ieee/tour"], case["tour"], case["ref_tie"])


def _second_opinion(name, got, ieee, ref, tie):
    """Fixtures that carry the second opinion of oracle/make_ieee_tours.py: the reference's algorithm with its two
    library-dependent choices fixed the way the kernels fix them (correctly rounded square roots instead of the math
    library's, lowest index first among equal distances).  That run is what the device must reproduce; how far the
    reference's own run is from it (its torch build's sqrt is one ulp low on 0.6 % of inputs, and the last decisions
    of a tour hang on the last bits of the running standard deviations) bounds what any implementation can share with
    the reference's tours."""
    same_ieee = (got == ieee).all(axis=1)
    same_ref = (got == ref).all(axis=1)
    ref_vs_ieee = (ref == ieee).all(axis=1)
    print(f"{name}, correctly rounded roots (DGL_SQRT_MODEL=ieee): identical to the reference's own run "
          f"{int(same_ref.sum())}/{len(same_ref)}; identical to the correctly-rounded-sqrt run of the reference's algorithm {int(same_ieee.sum())}/{len(same_ieee)}"
          f" ({100 * same_ieee.mean():.2f} %); the reference's own run shares {int(ref_vs_ieee.sum())}/{len(tie)} with that run; "
          f"{int(tie.sum())} instances pass a neighbourhood tie on the reference's trajectory, without them the device "
          f"shares {int((same_ref & ~tie).sum())}/{int((~tie).sum())} with the reference")
    assert same_ieee.mean() >= 0.99


def test_fused_greedy_decode_is_bit_identical(device, monkeypatch, ieee_roots):
    """dgl_tsp_greedy_decode (the whole decode loop in one launch) against two launches per decision: the same device
    code runs in both, so tours and lengths must be equal bit for bit; odd row count, neighbourhoods that shrink at
    the end of the tour."""
    from dgl_amd.engine import PackedPolicy, TspEngine
    from dgl_amd._lib import PROBLEM_TSP
    case, weights = load_golden("tsp_n100_b64_nn")
    xy = torch.from_numpy(case["xy"][:33]).to(device)
    pol = PackedPolicy(weights, PROBLEM_TSP, device)
    start = torch.from_numpy(case["start"]).to(device=device, dtype=torch.int32).repeat(33)
    eng = TspEngine(xy, 1, 30)
    monkeypatch.setenv("DGL_FUSED_DECODE", "0")
    ref_len = eng.greedy_rollout(pol, start).clone()
    ref_tour = eng.state["tour"].clone()
    monkeypatch.setenv("DGL_FUSED_DECODE", "1")
    assert eng.fused_decode_available()
    got_len = eng.greedy_rollout(pol, start).clone()
    assert torch.equal(eng.state["tour"], ref_tour) and torch.equal(got_len, ref_len)
    assert np.array_equal(ref_tour.cpu().numpy(), case["tour"][:33])


def test_fused_greedy_decode_refuses_the_reference_bits_model(device, monkeypatch):
    """The fused decode kernel's step phase takes correctly rounded roots only: under the default square-root model the
    engine does not offer it, and the library refuses a direct call before anything is launched."""
    import ctypes
    from dgl_amd import sqrt_model
    from dgl_amd.engine import PackedPolicy, TspEngine
    from dgl_amd._lib import PROBLEM_TSP, stream_ptr, ptr
    case, weights = load_golden("tsp_n100_b64_nn")
    eng = TspEngine(torch.from_numpy(case["xy"][:4]).to(device), 1, 30)
    pol = PackedPolicy(weights, PROBLEM_TSP, device)
    sqrt_model.use("mkl_avx512", device)
    monkeypatch.setenv("DGL_FUSED_DECODE", "1")
    assert not eng.fused_decode_available()
    eng.reset()
    eng.step(torch.zeros(4, dtype=torch.int32, device=device))
    io = eng.policy_io(want_logits=False, want_prob=False)
    rc = eng.lib.dgl_tsp_greedy_decode(ctypes.byref(eng.state["struct"]), ptr(pol.packed), ctypes.byref(io), eng.t, eng.N,
                                       eng.knn, stream_ptr())
    assert rc == -2 and b"correctly rounded" in eng.lib.dgl_last_error_string()
    sqrt_model.use(sqrt_model.default_model(), device)


def test_fused_greedy_decode_odd_rows_long_instances(device, monkeypatch, ieee_roots):
    """An odd row count leaves the last workgroup of the fused decode with a spare pair of waves, which walks the last
    row without storing anything.  Rows of 2100 nodes keep several groups of loads in flight per wave (and take the
    shared-compaction path), so a second in-place update of the last row's moments would show as different tours."""
    from dgl_amd.engine import PackedPolicy, TspEngine
    from dgl_amd._lib import PROBLEM_TSP
    _, weights = load_golden("tsp_n100_b64_nn")
    g = torch.Generator().manual_seed(99)
    xy = torch.rand(3, 2100, 2, generator=g).to(device)
    pol = PackedPolicy(weights, PROBLEM_TSP, device)
    start = torch.tensor([5, 5, 5], dtype=torch.int32, device=device)
    eng = TspEngine(xy, 1, 30)
    monkeypatch.setenv("DGL_FUSED_DECODE", "0")
    ref_len = eng.greedy_rollout(pol, start).clone()
    ref_tour = eng.state["tour"].clone()
    ref_avg = eng.state["avg"].clone()
    monkeypatch.setenv("DGL_FUSED_DECODE", "1")
    assert eng.fused_decode_available()
    for _ in range(4):                  # the same bits on every call (the policy phase splits operands in inline assembly)
        got_len = eng.greedy_rollout(pol, start).clone()
        assert torch.equal(eng.state["tour"], ref_tour) and torch.equal(got_len, ref_len)
        # NaN on the terminal step is the reference's own (division by zero unvisited nodes): compare bit patterns
        assert torch.equal(eng.state["avg"].view(torch.int32), ref_avg.view(torch.int32))


@pytest.mark.parametrize("name", ["tsp_n1000_b128_nn_tours", "tsp_n1000_b128_clustered_nn_tours"])
def test_tour_mismatches_attributed_to_reset_moments(name, device, ieee_roots):
    """(With correctly rounded roots; under the default model the reset already carries the reference's moments bit
    for bit.)  Where do the tours that differ from the reference's come from?  The device's reset sums its distances in
    binary64 and rounds once; torch.mean / torch.std round along a float cascade, so the two sets of starting moments
    differ in the last bit here and there, and the recurrences and the min-max normalisation carry that into the
    logits.  This decodes the population twice - from the device's own reset, and with the reference's captured reset
    moments (avg0 / std0 of the golden file) written over the device's - and reports identical tours for both.  What is
    still different in the second decode is owed to the arithmetic after the reset (policy sums, square roots)."""
    from dgl_amd.engine import PackedPolicy
    from dgl_amd._lib import PROBLEM_TSP
    case, weights = load_golden(name)
    eng = _engine(case, device)
    pol = PackedPolicy(weights, PROBLEM_TSP, device)
    start = torch.from_numpy(case["start"]).to(device=device, dtype=torch.int32).repeat(eng.B)
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
        same[moments] = int((eng.state["tour"].cpu().numpy() == case["tour"]).all(axis=1).sum())
    print(f"{name}: identical tours {same['own']}/{eng.R} from the device's reset, {same['reference']}/{eng.R} from the "
          f"reference's reset moments")
    assert same["reference"] >= same["own"] - 2
