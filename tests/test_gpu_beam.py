"""Beam expansion kernels (dense and compact) and the back-pointer permutation vs the golden
beam traces captured from the reference's Beamsearch.advance."""
import ctypes

import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


def _setup(name, device):
    from dgl_amd import _lib
    from dgl_amd.engine import CvrpEngine, PackedPolicy, TspEngine
    case, weights = load_golden(name)
    beam, pomo = int(case["meta/beam"]), int(case["meta/pomo"])
    if str(case["meta/problem"]) == "TSP":
        eng = TspEngine(torch.from_numpy(case["xy"]).to(device), pomo * beam, int(case["meta/knn"]))
        pol = PackedPolicy(weights, _lib.PROBLEM_TSP, device)
        A = eng.N
        start = torch.from_numpy(case["start"]).to(device=device, dtype=torch.int32).repeat(eng.B).repeat_interleave(beam)
        actions = None
    else:
        xy = np.concatenate([case["depot_xy"][:, None, :], case["node_xy"]], axis=1)
        dem = np.concatenate([np.zeros((xy.shape[0], 1), np.float32), case["demand"]], axis=1)
        eng = CvrpEngine(torch.from_numpy(xy).to(device), torch.from_numpy(dem).to(device), pomo * beam,
                         int(case["meta/knn"]), int(case["meta/depot_knn"]))
        pol = PackedPolicy(weights, _lib.PROBLEM_CVRP, device)
        A = 2 * (eng.N + 1)
        start = torch.from_numpy(case["start"]).to(device=device, dtype=torch.int32).repeat(eng.B).repeat_interleave(beam)
        start = start + 2 + eng.N
    return case, eng, pol, beam, A, start


@pytest.mark.parametrize("name", ["tsp_n60_b2_beam4_pomo2", "tsp_n200_b1_beam16", "tsp_n200_b2_beam8_nn",
                                  "cvrp_n60_b2_beam4_pomo2", "cvrp_n150_b2_beam8_nn",
                                  "tsp_n150_b1_beam100_k100_nn", "cvrp_n100_b1_beam100_k100_nn",
                                  "tsp_n25_b2_beam4", "cvrp_n24_b2_beam4",
                                  "tsp_n1000_b2_beam16_nn", "cvrp_n1000_b2_beam16_nn",     # the headline size
                                  "tsp_n10000_b1_beam16_nn"])      # configuration 4 as the reference runs it: 9 999 expansions
def test_teacher_forced_beam_steps(name, device):
    """At every step: dense and compact kernels agree with each other and with the reference's
    (scores, prev_k, action); the state is then advanced with the reference's own choice.
    The beam-100 cases are the reference's training setting (CVRP/train.py:70,130-131: beam 100 over rows of 101 / 202
    tokens): more explicit candidates per group than the compact kernel sorts in LDS."""
    from dgl_amd import _lib
    lib = _lib.load()
    case, eng, pol, beam, A, start = _setup(name, device)
    G = eng.R // beam
    eng.reset()
    eng.step(start)
    scores = torch.zeros(G, beam, device=device)
    pk_d = torch.zeros(G, beam, dtype=torch.int32, device=device)
    ac_d = torch.zeros_like(pk_d)
    pk_c, ac_c = torch.zeros_like(pk_d), torch.zeros_like(pk_d)
    nsteps = case["beam_scores"].shape[0]
    mism_dense = mism_compact = n_sep = 0
    max_score_err = 0.0
    for s in range(nsteps):
        step = s + 1
        eng.forward(pol)
        dense = eng.dense_probs()
        logp = torch.log(dense).view(G, beam, A).contiguous()
        sc_d, sc_c = scores.clone(), scores.clone()
        _lib.check(lib.dgl_beam_advance(logp.data_ptr(), sc_d.data_ptr(), G, beam, A, step, pk_d.data_ptr(),
                                        ac_d.data_ptr(), _lib.stream_ptr()), "dense")
        eng.beam_advance(sc_c, beam, step, pk_c, ac_c)
        g_sc, g_pk, g_ac = case["beam_scores"][s], case["beam_prev_k"][s], case["beam_action"][s]
        # reference: exact agreement expected at every position whose score is separated from both neighbours'
        # (separated: by more than the two score lists differ at this step - candidates whose reference scores lie
        # closer together than the device's scores are off may legitimately change places)
        eps = float(np.abs(sc_c.cpu().numpy() - g_sc).max(initial=0.0))
        gap = np.abs(np.diff(g_sc, axis=1)) > max(1e-4, 2.5 * eps) if beam > 1 else np.ones((G, 0), bool)
        # (the last position competes with a candidate the reference did not keep: its margin is unknown)
        sep = np.concatenate([np.ones((G, 1), bool), gap], axis=1) & np.concatenate([gap, np.zeros((G, 1), bool)], axis=1)
        if beam == 1:
            sep[:] = True
        ok_d = (pk_d.cpu().numpy() == g_pk) & (ac_d.cpu().numpy() == g_ac)
        ok_c = (pk_c.cpu().numpy() == g_pk) & (ac_c.cpu().numpy() == g_ac)
        mism_dense += int((~ok_d & sep).sum())
        mism_compact += int((~ok_c & sep).sum())
        n_sep += int(sep.sum())
        # the two kernels see the same probabilities: they must agree everywhere, ties included
        assert torch.equal(pk_d, pk_c) and torch.equal(ac_d, ac_c), f"dense and compact expansion differ at step {step}"
        # the score lists agree as lists whatever the order among near-ties
        max_score_err = max(max_score_err, float(np.abs(sc_c.cpu().numpy() - g_sc).max(initial=0.0)))
        # continue from the reference's decision
        scores = torch.from_numpy(g_sc).to(device)
        eng.permute(torch.from_numpy(g_pk).to(device).contiguous(), beam)
        eng.step(torch.from_numpy(g_ac).to(device).view(-1).contiguous())
    print(f"{name}: steps {nsteps}, groups {G}, well-separated positions {n_sep} of {nsteps * G * beam}, mismatches "
          f"dense {mism_dense} compact {mism_compact}, max |score err| {max_score_err:.2e}")
    assert n_sep > nsteps * G * beam // 4
    assert mism_dense == 0 and mism_compact == 0
    # log-probabilities of rows of 101 / 202 tokens inherit the logits' tolerance (1e-3 of the row's largest logit)
    assert max_score_err < (1e-2 if "k100" in name else 5e-3)
    # final state equals the reference's tours
    key = "tour" if "tour" in case else "nodes"
    mine = (eng.state["tour"] if key == "tour" else eng.state["nodes"]).cpu().numpy()
    assert np.array_equal(mine, case[key])
    length = eng.tour_length().cpu().numpy()
    assert np.array_equal(length, case["length"])               # the reference's own floats (default square-root model)


@pytest.mark.parametrize("name", ["tsp_n200_b2_beam8_nn", "cvrp_n150_b2_beam8_nn",
                                  "tsp_n150_b1_beam100_k100_nn", "cvrp_n100_b1_beam100_k100_nn",
                                  "tsp_n1000_b2_beam16_nn", "cvrp_n1000_b2_beam16_nn", "tsp_n10000_b1_beam16_nn"])
def test_free_running_beam_rollout_trained(name, device):
    case, eng, pol, beam, A, start = _setup(name, device)
    length = eng.beam_rollout(pol, start, beam).cpu().numpy()
    best = length.reshape(eng.B, -1).min(axis=1)
    gbest = case["length"].reshape(eng.B, -1).min(axis=1)
    print(f"{name}: best-of-beam lengths {best} vs reference {gbest}")
    np.testing.assert_allclose(best, gbest, rtol=1e-3)


def test_dense_advance_synthetic(device):
    """Dense kernel vs torch.topk on random scores, including step 1 (only beam 0 counts), heavy
    ties (flat-index tie-break) and beam*A not a multiple of the block size."""
    from dgl_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(0)
    for (G, beam, A, step, ties) in [(3, 4, 100, 2, False), (2, 16, 1000, 5, False), (2, 8, 333, 1, False),
                                     (2, 32, 2002, 3, True), (1, 100, 5000, 2, True), (1, 1, 77, 2, False)]:
        lp = torch.randn(G, beam, A, generator=g)
        if ties:
            lp = torch.where(torch.rand(G, beam, A, generator=g) < 0.97, torch.full_like(lp, -46.0517), lp)
        sc = torch.randn(G, beam, generator=g)
        lk = lp + sc[:, :, None] if step >= 2 else torch.cat([lp[:, :1], torch.full_like(lp[:, 1:], -1e20)], 1)
        flat = lk.reshape(G, -1)
        # expected: value descending, flat index ascending among equals
        order = np.lexsort((np.arange(flat.shape[1])[None, :].repeat(G, 0), -flat.numpy().astype(np.float64)), axis=1)[:, :beam]
        exp_val = np.take_along_axis(flat.numpy(), order, 1)
        lp_d, sc_d = lp.to(device).contiguous(), sc.to(device).contiguous()
        pk = torch.zeros(G, beam, dtype=torch.int32, device=device)
        ac = torch.zeros_like(pk)
        _lib.check(lib.dgl_beam_advance(lp_d.data_ptr(), sc_d.data_ptr(), G, beam, A, step, pk.data_ptr(), ac.data_ptr(),
                                        _lib.stream_ptr()), "dense")
        got = pk.cpu().numpy().astype(np.int64) * A + ac.cpu().numpy()
        assert np.array_equal(got, order), (G, beam, A, step)
        assert np.array_equal(sc_d.cpu().numpy(), exp_val)


def test_tsp10000_beam_steps_follow_oracle(device):
    """Large-n stress (TSP-10000, beam 4): the first beam steps against the oracle - kNN index sets
    bit-exact, beam decisions identical where the scores are separated.  One wave per row with 40 KB of
    keys in LDS; no [N][N] matrix on the device."""
    from dgl_amd import _lib
    from dgl_amd.engine import PackedPolicy, TspEngine
    from oracle.dgl_oracle import TSPOracle, beam_advance, random_weights
    N, beam, knn, steps = 10000, 4, 30, 6
    g = torch.Generator().manual_seed(31)
    xy = torch.rand(1, N, 2, generator=g)
    w = random_weights("TSP", 8)
    orc = TSPOracle(w, knn)
    orc.reset(xy, rows_per_instance=beam)
    start = torch.full((beam,), 4321, dtype=torch.long)
    orc.step(start)
    eng = TspEngine(xy.to(device), beam, knn)
    pol = PackedPolicy(w, _lib.PROBLEM_TSP, device)
    eng.reset()
    eng.step(start.to(device=device, dtype=torch.int32))
    scores_o = torch.zeros(1, beam)
    scores = torch.zeros(1, beam, device=device)
    pk, ac = torch.zeros(1, beam, dtype=torch.int32, device=device), torch.zeros(1, beam, dtype=torch.int32, device=device)
    for s in range(steps):
        out = orc.forward()
        assert np.array_equal(eng.idx.cpu().numpy(), out["idx"].numpy()), f"kNN sets differ at step {s + 1}"
        best, prev_k, action = beam_advance(torch.log(out["probs"]).view(1, beam, -1), scores_o, s + 1, beam, N)
        eng.forward(pol)
        sc = scores.clone()
        eng.beam_advance(sc, beam, s + 1, pk, ac)
        sep = float(np.abs(np.diff(best.numpy(), axis=1)).min()) > 1e-4
        if sep:
            assert np.array_equal(pk.cpu().numpy(), prev_k.numpy()) and np.array_equal(ac.cpu().numpy(), action.numpy())
        np.testing.assert_allclose(sc.cpu().numpy(), best.numpy(), atol=5e-3)
        scores_o = best
        scores = best.to(device)
        orc.permute_beams(prev_k, beam)
        orc.step(action.view(-1))
        eng.permute(prev_k.to(device=device, dtype=torch.int32).contiguous(), beam)
        eng.step(action.view(-1).to(device=device, dtype=torch.int32).contiguous())
    a, sd_ = eng.state["avg"].cpu().numpy(), eng.state["sd"].cpu().numpy()
    np.testing.assert_allclose(a, orc.avg.numpy(), rtol=2e-5)
    np.testing.assert_allclose(sd_, orc.std.numpy(), rtol=2e-5)
