"""Checkpoint layout of the reference's models (TSP/TSPModel.py:238-339, CVRP/CVRPModel.py:378-427).

``state_dict_shapes`` lists every key of ``model_state_dict`` (46 tensors for TSP, 48 for
CVRP), including ``decoder.embedding_last_node_pos`` and ``decoder.k_1`` which the
reference constructs and saves but never uses.
"""
from __future__ import annotations

import math
from typing import Dict

import torch


def _block(prefix: str) -> Dict[str, tuple]:
    return {
        prefix + "Wq.weight": (128, 128), prefix + "Wk.weight": (128, 128), prefix + "Wv.weight": (128, 128),
        prefix + "multi_head_combine.weight": (128, 128), prefix + "multi_head_combine.bias": (128,),
        prefix + "feedForward.W1.weight": (512, 128), prefix + "feedForward.W1.bias": (512,),
        prefix + "feedForward.W2.weight": (128, 512), prefix + "feedForward.W2.bias": (128,),
    }


def state_dict_shapes(problem: str) -> Dict[str, tuple]:
    feat = 7 if problem == "TSP" else 9
    s = {"encoder.embedding.weight": (128, feat), "encoder.embedding.bias": (128,)}
    s.update(_block("encoder.layers.0."))
    if problem == "TSP":
        s["decoder.embedding_last_node.weight"] = (128, 128)
        s["decoder.embedding_last_node.bias"] = (128,)
    else:
        for nm in ("not_via_depot", "via_depot"):
            s[f"decoder.embedding_last_node_{nm}.weight"] = (128, 128)
            s[f"decoder.embedding_last_node_{nm}.bias"] = (128,)
    s["decoder.embedding_last_node_pos.weight"] = (128, 128)
    s["decoder.embedding_last_node_pos.bias"] = (128,)
    for l in range(3):
        s.update(_block(f"decoder.layers.{l}."))
    s["decoder.k_1.weight"] = (128, 128)
    s["decoder.k_1.bias"] = (128,)
    s["decoder.Linear_final.weight"] = (1, 128)
    s["decoder.Linear_final.bias"] = (1,)
    return s


def random_state_dict(problem: str, seed: int) -> Dict[str, torch.Tensor]:
    """Random-init weights with nn.Linear's scale (uniform +-1/sqrt(fan_in)) from a private
    generator: synthetic weights of the right architecture for benchmarks and smoke tests."""
    g = torch.Generator().manual_seed(seed)
    shapes = state_dict_shapes(problem)
    out = {}
    for k, shp in shapes.items():
        fan_in = shp[1] if len(shp) == 2 else shapes[k[:-4] + "weight"][1]
        out[k] = (torch.rand(shp, generator=g) * 2 - 1) / math.sqrt(fan_in)
    return out


def load_checkpoint(path: str) -> Dict[str, torch.Tensor]:
    """Reads a checkpoint written by the reference's trainers
    ({'epoch', 'model_state_dict', 'optimizer_state_dict', ...}; CVRP/CVRPTrainer.py:126-135)."""
    ck = torch.load(path, map_location="cpu", weights_only=True)
    sd = ck["model_state_dict"] if "model_state_dict" in ck else ck
    return {k: v.float() for k, v in sd.items()}
