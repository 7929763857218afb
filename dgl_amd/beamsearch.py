"""Beamsearch with the reference's interface (utils/beamsearch.py:8-131); the expansion runs in
dgl_beam_advance, the state re-indexing in the environments' perm_attr (device kernels)."""
from __future__ import annotations

import torch

from . import _lib
from ._lib import check, stream_ptr


class Beamsearch(object):
    def __init__(self, beam_size, batch_size, num_nodes, dtypeFloat=None, dtypeLong=None, probs_type="raw",
                 random_start=False, problem="TSP", device="cuda"):
        self.batch_size = batch_size
        self.beam_size = beam_size
        self.num_nodes = num_nodes
        self.probs_type = probs_type
        self.problem = problem
        self.device = torch.device(device)
        self.start_nodes = torch.zeros(batch_size, beam_size, dtype=torch.int64, device=self.device)
        if random_start:
            self.start_nodes = torch.randint(0, num_nodes, (batch_size, beam_size), device=self.device)
        self.scores = torch.zeros(batch_size, beam_size, dtype=torch.float32, device=self.device)
        self.all_scores = []
        self.prev_Ks = []
        self.next_nodes = [self.start_nodes]
        self._lib = _lib.load()

    def advance(self, trans_probs, env, knn, current_step):
        """trans_probs: log-probabilities [batch, beam, A] (A = num_nodes for TSP, 2*num_nodes for CVRP)."""
        A = self.num_nodes if self.problem == "TSP" else 2 * self.num_nodes
        lp = trans_probs.to(self.device, torch.float32).contiguous().view(self.batch_size, self.beam_size, A)
        scores = self.scores.clone()
        prev_k = torch.empty(self.batch_size, self.beam_size, dtype=torch.int32, device=self.device)
        action = torch.empty_like(prev_k)
        check(self._lib.dgl_beam_advance(lp.data_ptr(), scores.data_ptr(), self.batch_size, self.beam_size, A,
                                         int(current_step), prev_k.data_ptr(), action.data_ptr(), stream_ptr()),
              "dgl_beam_advance")
        self.scores = scores
        self.prev_Ks.append(prev_k.to(torch.int64))
        self.next_nodes.append(action.to(torch.int64))
        env.perm_attr("data", prev_k)
        return env.selected_node_list
