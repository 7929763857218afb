"""Child of tests/test_gpu_multirank.py: one rank of a world of 1 or 2 processes (gloo) that all use cuda:0.

Part A - evaluation: TSPTester.run() over 16 instances read from files; the ranks shard the instances and all-reduce
[sum of lengths, sum of gaps, count] (TSP/TSPTester.py:123-139).  Part B - training: one decode step of PolicyTrainer
(forward in mode 'train', backward, gradient all-reduce, Adam; CVRP/CVRPTrainer.py:268-272) on this rank's block of a
global batch of 8 rows.  Rank 0 writes what the test compares: score, gap, the all-reduced gradient, the parameters
after the update.

    python tests/multirank_child.py OUT_DIR          (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT from the environment)
"""
import json
import logging
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, HERE)
from conftest import GOLDEN, load_golden  # noqa: E402

TSP_INFO = [True, True, True, False, True, True, False, False, False, False, False]
MODEL_PARAMS = dict(mode="test", embedding_dim=128, sqrt_embedding_dim=128 ** 0.5, decoder_layer_num=3, qkv_dim=16,
                    head_num=8, ff_hidden_dim=512)


def main():
    out_dir = sys.argv[1]
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    from dgl_amd import TSPTester
    from dgl_amd._lib import PROBLEM_TSP
    from dgl_amd.distributed import shard_range
    from dgl_amd.engine import TspEngine
    from dgl_amd.train import PolicyTrainer

    # ---- A: file-driven evaluation --------------------------------------------------------
    case, weights = load_golden("tsp_n100_b64_nn")
    xy = case["xy"][:16]
    data, base = os.path.join(out_dir, "tsp100.txt"), os.path.join(out_dir, "base.txt")
    if rank == 0:
        with open(data, "w") as f:
            f.write(" ".join(f"{x:.9g},{y:.9g}" for x, y in xy.reshape(-1, 2)))
        with open(base, "w") as f:
            f.write("\n".join(f"{i} {l:.9g}" for i, l in enumerate(case["length"][:16])))
    if world > 1:
        dist.barrier()
    env_params = dict(data_path=data, baseline_path=base, mode="test", test_mode="pomo_test", aug_size=1,
                      append_information=TSP_INFO, aggregation_nums=50, pomo_size=1)
    tester_params = dict(use_cuda=True, cuda_device_num=0, test_mode="pomo_test", knn=30, beam_size=1, problem_size=100,
                         test_episodes_all=16, test_episodes=16, test_batch_size=4,
                         model_load={"enable": True, "path": GOLDEN, "file": "ckpt_tsp_nn.pt"})
    t = TSPTester(env_params, dict(MODEL_PARAMS, append_information=TSP_INFO), None, tester_params, None)
    gap = t.run()

    # ---- B: one training decode step on a split batch ----------------------------------------
    g = torch.Generator().manual_seed(11)
    B, N, knn = 8, 60, 30
    gxy = torch.rand(B, N, 2, generator=g)
    start = torch.randint(0, N, (B,), generator=g, dtype=torch.int32)
    lo, hi = shard_range(B, rank, world)
    eng = TspEngine(gxy[lo:hi].to(dev), 1, knn)
    eng.reset()
    eng.step(start[lo:hi].to(dev))
    teacher = eng.idx[:, 3].clone()
    tr = PolicyTrainer(weights, PROBLEM_TSP, dev)
    tr.zero_grad()
    prob = tr.forward_backward(eng, teacher)                 # scaled by 1 / (rows here * world) = 1 / B
    tr.optimizer_step()                                      # gradient all-reduce, then Adam
    torch.cuda.synchronize()
    if rank == 0:
        np.save(os.path.join(out_dir, f"grad_w{world}.npy"), tr.grads.cpu().numpy())
        np.save(os.path.join(out_dir, f"params_w{world}.npy"), tr.params.cpu().numpy())
        with open(os.path.join(out_dir, f"eval_w{world}.json"), "w") as f:
            json.dump({"score": t.score, "gap": gap, "world": world, "prob0": float(prob[0])}, f)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    logging.disable(logging.INFO)
    main()
