"""nn.Module parameter containers whose state_dict keys, shapes and default initialisation are
those of the reference's encoder/decoder (TSP/TSPModel.py:238-457, CVRP/CVRPModel.py:378-642).
They hold weights only (checkpoint interchange); no arithmetic happens in these modules."""
from __future__ import annotations

import torch.nn as nn


class FeedForwardParams(nn.Module):
    def __init__(self):
        super().__init__()
        self.W1 = nn.Linear(128, 512)
        self.W2 = nn.Linear(512, 128)


class BlockParams(nn.Module):
    def __init__(self):
        super().__init__()
        self.Wq = nn.Linear(128, 128, bias=False)
        self.Wk = nn.Linear(128, 128, bias=False)
        self.Wv = nn.Linear(128, 128, bias=False)
        self.multi_head_combine = nn.Linear(128, 128)
        self.feedForward = FeedForwardParams()


class EncoderParams(nn.Module):
    def __init__(self, features: int):
        super().__init__()
        self.embedding = nn.Linear(features, 128, bias=True)
        self.layers = nn.ModuleList([BlockParams()])


class DecoderParams(nn.Module):
    def __init__(self, problem: str):
        super().__init__()
        if problem == "TSP":
            self.embedding_last_node = nn.Linear(128, 128, bias=True)
        else:
            self.embedding_last_node_not_via_depot = nn.Linear(128, 128, bias=True)
            self.embedding_last_node_via_depot = nn.Linear(128, 128, bias=True)
        self.embedding_last_node_pos = nn.Linear(128, 128, bias=True)      # saved by the reference, never used
        self.layers = nn.ModuleList([BlockParams() for _ in range(3)])
        self.k_1 = nn.Linear(128, 128, bias=True)                          # saved by the reference, never used
        self.Linear_final = nn.Linear(128, 1, bias=True)
