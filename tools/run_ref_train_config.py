"""One loop of the reference's own training configuration (CVRP/train.py with its debug overrides:
batch 64, beam 4, N = 100, knn = depot_knn = 100, step_size 32, pomo 16) through dgl_amd.CVRPTrainer."""
import logging, os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from dgl_amd import CVRPTrainer
info = [True, True, True, False, True, True, False, False, False, False, False, False, False]
logging.basicConfig(level=logging.INFO, format="%(message)s")
tr = CVRPTrainer(
    env_params=dict(mode='train', sub_path=False, replace_size=4, append_information=info, pomo_size=16, aggregation_nums=10,
                    max_replace_ratio=0.125, min_replace_ratio=0.125, test_mode="none"),
    model_params=dict(mode='train', embedding_dim=128, sqrt_embedding_dim=128 ** 0.5, decoder_layer_num=3, qkv_dim=16, head_num=8,
                      ff_hidden_dim=512, append_information=info),
    optimizer_params=dict(optimizer=dict(lr=1e-4, weight_decay=1e-6), scheduler=dict(milestones=list(range(1, 200)), gamma=0.97)),
    trainer_params=dict(use_cuda=True, cuda_device_num=0, epochs=int(sys.argv[1]) if len(sys.argv) > 1 else 1, train_batch_size=64,
                        beam_size=4, problem_size=100, knn=100, depot_knn=100, step_size=32, loops_per_epoch=int(sys.argv[2]) if len(sys.argv) > 2 else 2,
                        result_folder="/tmp/dgl_train_result", model_load={"enable": False}))
t0 = time.time()
tr.run()
torch.cuda.synchronize()
print(f"{time.time() - t0:.1f} s; label length mean {tr.env.solution_len.mean().item():.4f}; losses {tr.result_log.get('train_loss')}")
