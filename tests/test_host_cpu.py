"""Host-side logic that needs no GPU: checkpoint layout, sharding, statistics all-reduce (gloo)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import GOLDEN


def test_state_dict_layout_matches_reference_checkpoints():
    from dgl_amd.cvrp import CVRPModel
    from dgl_amd.tsp import TSPModel
    from dgl_amd.weights import load_checkpoint, random_state_dict, state_dict_shapes
    mp_ = dict(mode="test", embedding_dim=128, sqrt_embedding_dim=128 ** 0.5, decoder_layer_num=3, qkv_dim=16, head_num=8,
               ff_hidden_dim=512)
    for cls, pb, n in ((TSPModel, "TSP", 46), (CVRPModel, "CVRP", 48)):
        m = cls(**mp_)
        sd = m.state_dict()
        assert len(sd) == n
        assert {k: tuple(v.shape) for k, v in sd.items()} == state_dict_shapes(pb)
        ck = load_checkpoint(os.path.join(GOLDEN, f"ckpt_{pb.lower()}_nn.pt"))      # written by the reference's classes
        assert set(ck) == set(sd)
        m.load_state_dict(ck)
        assert set(random_state_dict(pb, 1)) == set(sd)
        assert sum(v.numel() for v in sd.values()) == (840193 if pb == "TSP" else 856961)


def test_unsupported_configurations_fail_loudly():
    from dgl_amd.tsp import TSPEnv, TSPModel
    with pytest.raises(NotImplementedError):
        TSPModel(mode="test", embedding_dim=256, head_num=8, qkv_dim=16, ff_hidden_dim=512, decoder_layer_num=3)
    info = [True] * 11
    with pytest.raises(NotImplementedError):
        TSPEnv(mode="test", test_mode="pomo_test", aug_size=1, append_information=info, data_path="", device="cpu")


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from dgl_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.DglError):
        _lib.load()


def test_shard_range_covers_everything():
    from dgl_amd.distributed import shard_range
    for total in (0, 1, 7, 8, 20, 200, 2000):
        for ws in (1, 2, 3, 8):
            got = []
            for r in range(ws):
                lo, hi = shard_range(total, r, ws)
                assert 0 <= lo <= hi <= total
                got += list(range(lo, hi))
            assert got == list(range(total))


def _worker(rank, ws, port, out):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    from dgl_amd.distributed import reduce_stats, shard_range, world
    assert world() == (rank, ws)
    lengths = np.arange(101, dtype=np.float64) * 0.25 + 3.0          # per-instance "tour lengths"
    base = lengths * 0.9
    lo, hi = shard_range(len(lengths), rank, ws)
    s, g, c = reduce_stats(float(lengths[lo:hi].sum()), float(((lengths[lo:hi] - base[lo:hi]) / base[lo:hi]).sum()),
                           float(hi - lo))
    out[rank] = (s, g, c)
    dist.destroy_process_group()


def test_stats_allreduce_gloo_world2():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    lengths = np.arange(101, dtype=np.float64) * 0.25 + 3.0
    for r in (0, 1):
        s, g, c = out[r]
        assert c == 101 and abs(s - lengths.sum()) < 1e-9 and abs(g - 101 * (1 / 0.9 - 1)) < 1e-9


def test_cvrp_instance_file_parser(tmp_path):
    from dgl_amd.cvrp import read_cvrp_instances_from_file
    p = tmp_path / "cvrp.txt"
    p.write_text("0.5,0.5 .|. 0.1,0.2 0.3,0.4 0.9,0.8 .|. 10 20 5 .|. 50\n0.1,0.9 .|. 0.2,0.2 0.4,0.4 0.6,0.6 .|. 1 2 3 .|. 10\n")
    depot, nodes, dem = read_cvrp_instances_from_file(str(p))
    assert depot.shape == (2, 2) and nodes.shape == (2, 3, 2) and dem.shape == (2, 3)
    np.testing.assert_allclose(dem.numpy(), [[0.2, 0.4, 0.1], [0.1, 0.2, 0.3]])


def _grad_worker(rank, ws, port, out):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    from dgl_amd.distributed import all_reduce_gradients, shard_range
    from oracle.dgl_oracle import TSPOracle, enable_grad, random_weights
    torch.set_num_threads(1)
    B, N, knn = 6, 40, 10
    g = torch.Generator().manual_seed(5)
    xy = torch.rand(B, N, 2, generator=g)
    start = torch.randint(0, N, (B,), generator=g)
    w = random_weights("TSP", 7)
    lo, hi = shard_range(B, rank, ws)
    orc = TSPOracle(w, knn)
    orc.w = enable_grad(w)
    orc.reset(xy[lo:hi])
    orc.step(start[lo:hi])
    out_f = orc.forward()
    teacher = orc.knn_indices()[:, 1]
    prob = out_f["probs"][torch.arange(hi - lo), teacher]
    loss = -(prob.double().log().sum() / B)                    # this rank's share of the global mean
    loss.backward()
    keys = [k for k, v in orc.w.items() if v.grad is not None]
    flat = torch.cat([orc.w[k].grad.reshape(-1) for k in keys])
    all_reduce_gradients(flat)
    out[rank] = flat.numpy().copy()
    dist.destroy_process_group()


def test_gradient_allreduce_gloo_world2():
    """Per-rank gradients of (local sum of -log p) / (global rows), summed over the ranks, equal the
    gradient of the reference's mean over the whole batch."""
    import torch
    import torch.multiprocessing as mp
    from oracle.dgl_oracle import TSPOracle, enable_grad, random_weights, train_forward_backward
    ctx = mp.get_context("spawn")
    out = ctx.Manager().dict()
    port = 29500 + (os.getpid() % 2000) + 7
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    B, N, knn = 6, 40, 10
    g = torch.Generator().manual_seed(5)
    xy = torch.rand(B, N, 2, generator=g)
    start = torch.randint(0, N, (B,), generator=g)
    w = random_weights("TSP", 7)
    orc = TSPOracle(w, knn)
    orc.w = enable_grad(w)
    orc.reset(xy)
    orc.step(start)
    train_forward_backward(orc, orc.knn_indices()[:, 1])
    ref = torch.cat([v.grad.reshape(-1) for v in orc.w.values() if v.grad is not None]).numpy()
    assert np.array_equal(out[0], out[1])
    np.testing.assert_allclose(out[0], ref, rtol=2e-4, atol=1e-5 * np.abs(ref).max())


def test_multistep_lr_follows_torch_on_a_fresh_run_and_on_resume():
    """The trainers' scheduler against torch.optim.lr_scheduler.MultiStepLR driven the way the reference drives it:
    fresh run, and resume (optimizer state restored, last_epoch = epoch - 1 as TSP/TSPTrainer.py:75-84 sets it)."""
    import types
    import torch
    from dgl_amd.trainer import MultiStepLR
    milestones, gamma, base = list(range(1, 200)), 0.97, 1e-4          # CVRP/train.py:55-58
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.Adam([p], lr=base)
    ref = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=milestones, gamma=gamma)
    mine_tr = types.SimpleNamespace(lr=base)
    mine = MultiStepLR(mine_tr, milestones, gamma)
    for _ in range(10):                                                # ten saved checkpoints = fifty epochs
        opt.step(); ref.step(); mine.step()
        assert abs(mine_tr.lr - opt.param_groups[0]["lr"]) <= 1e-12 * base
    saved_lr = opt.param_groups[0]["lr"]
    # resume at epoch 50: the checkpoint's optimizer state carries the lr, the trainer pokes last_epoch
    opt2 = torch.optim.Adam([p], lr=base)
    ref2 = torch.optim.lr_scheduler.MultiStepLR(opt2, milestones=milestones, gamma=gamma)
    opt2.load_state_dict(opt.state_dict())
    ref2.last_epoch = 50 - 1
    tr2 = types.SimpleNamespace(lr=base)
    mine2 = MultiStepLR(tr2, milestones, gamma)
    tr2.lr = saved_lr                                                  # PolicyTrainer.load_optimizer_state_dict
    mine2.last_epoch = 50 - 1
    for _ in range(3):
        opt2.step(); ref2.step(); mine2.step()
        assert abs(tr2.lr - opt2.param_groups[0]["lr"]) <= 1e-12 * base
    assert abs(tr2.lr - saved_lr * gamma ** 3) <= 1e-12 * base


def test_rsqrt14_segment_constants_reproduce_the_recorded_instruction():
    """The 64 integer pairs compiled into the kernels (dgl_amd/csrc/common.h: DGL_RSQRT14_SEG) are the ones the recording
    of VRSQRT14 determines - one fit per segment - and give back every one of its 2^16 distinct values."""
    import re
    from dgl_amd.sqrt_model import rsqrt14_segments, rsqrt14_table
    table = rsqrt14_table()
    seg = rsqrt14_segments(table)
    src = open(os.path.join(os.path.dirname(GOLDEN), "..", "dgl_amd", "csrc", "common.h")).read()
    body = src[src.index("DGL_RSQRT14_SEG[64] = {") + len("DGL_RSQRT14_SEG[64] = {"):]
    compiled = np.array([int(v) for v in re.findall(r"-?\d+", body[:body.index("};")])], np.int64).reshape(64, 2)
    assert np.array_equal(compiled, seg)
    # the kernels' evaluation, for all 2^16 (parity, upper mantissa) classes: index (bits >> 18) & 63, m = bits 17..8
    cls = np.arange(1 << 16, dtype=np.int64)                  # parity << 15 | mantissa >> 8
    bits = ((1 - (cls >> 15)) << 23) | ((cls & 0x7FFF) << 8)  # a float's low 24 bits with that parity and mantissa
    c, ns = compiled[(bits >> 18) & 63].T
    assert np.array_equal((c + ns * ((bits >> 8) & 1023)) >> 9, table.astype(np.int64))
