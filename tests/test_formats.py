"""File formats and library scoring against what the reference's own readers return for the same
fixture files (tests/golden/formats, written by oracle/make_format_goldens.py)."""
import json
import os

import numpy as np
import torch

from conftest import GOLDEN
from dgl_amd import formats as F

FX = os.path.join(GOLDEN, "formats")
with open(os.path.join(FX, "expected.json")) as _f:
    EXP = json.load(_f)


def test_tsplib_file_normalisation_and_integer_scoring():
    e = EXP["tsplib"]
    inst, name = F.load_tsplib_file(FX, "syn17")
    assert name == e["name"] and inst.tolist() == e["nodes"]
    assert F.normalize_nodes_to_unit_board(inst).tolist() == e["normalized"]
    tour = torch.tensor(e["tour"])
    assert F.calculate_tour_length_by_dist_matrix(F.get_dist_matrix(inst), tour).item() == e["tour_len"]
    length, gap = F.tsplib_gap(inst, tour, 1000)
    assert length == int(np.ceil(e["tour_len"])) and gap == length / 1000 - 1
    assert list(F.parse_tsplib_name("syn17")) == e["parse_name"]
    assert [F.choose_bsz(s) for s in (100, 200, 201, 1000, 1001, 5000, 5001)] == e["bsz"]


def test_cvrplib_file_and_route_scoring():
    e = EXP["cvrplib"]
    depot, nodes, demands, capacity, name = F.load_cvrplib_file(FX, "S-n12-k3")
    assert depot.tolist() == e["depot"] and nodes.tolist() == e["nodes"] and demands.tolist() == e["demands"]
    assert int(capacity) == e["capacity"] and name == e["name"]
    assert list(F.parse_cvrplib_name("S-n12-k3")) == e["parse_name"]
    label, flags = torch.tensor(e["label"]), torch.tensor(e["flags"])
    assert F.cvrplib_route(label, flags, nodes.shape[0]).tolist() == e["route"]
    length, gap = F.cvrplib_gap(depot, nodes, label, flags, 500)
    assert length == int(np.ceil(e["route_len"])) and gap == length / 500 - 1


def test_instance_and_solution_text(tmp_path):
    e = EXP["cvrp_text"]
    depot, nodes, demands = F.read_cvrp_instances_from_file(os.path.join(FX, "cvrp_instances.txt"))
    assert depot.tolist() == e["depot"] and nodes.tolist() == e["nodes"] and demands.tolist() == e["demands"]
    tours, lens, secs = F.read_solutions_from_file(os.path.join(FX, "solutions.txt"))
    s = EXP["solutions"]
    assert tours.tolist() == s["tours"] and lens.tolist() == s["lens"] and secs.tolist() == s["secs"]
    t = EXP["tsp_text"]
    assert F.read_tsp_instances(os.path.join(FX, "tsp_instances.txt"), 2, 5).tolist() == t["nodes"]
    assert F.read_baseline_lengths(os.path.join(FX, "tsp_baseline.txt")).tolist() == t["baseline"]
    # writers round-trip through the readers
    p = tmp_path / "tsp.txt"
    F.write_tsp_instances(p, torch.tensor(t["nodes"]))
    assert F.read_tsp_instances(p, 2, 5).tolist() == t["nodes"]
    q = tmp_path / "cvrp.txt"
    ints = torch.randint(1, 10, (3, 5))
    F.write_cvrp_instances(q, depot, nodes, ints, torch.tensor([20, 21, 22]))
    d2, n2, dem2 = F.read_cvrp_instances_from_file(q)
    assert torch.equal(d2, depot) and torch.equal(n2, nodes)
    assert torch.equal(dem2, ints.float() / torch.tensor([20.0, 21.0, 22.0])[:, None])
    r = tmp_path / "sol.txt"
    F.write_solutions(r, [[0, 2, 1, 0], [0, 1, 0]], [1.5, 2.0], [0.25, 0.5])
    t2, l2, s2 = F.read_solutions_from_file(r)
    assert t2.tolist() == [[0, 2, 1, 0], [0, 1, 0, 0]] and l2.tolist() == [1.5, 2.0] and s2.tolist() == [0.25, 0.5]


def test_best_known_tables():
    assert F.tsplib_collections["pr1002"] == 259045 and F.cvrplib_collections["X-n1001-k43"] == 72355
    assert len(F.tsplib_collections) == 77 and len(F.cvrplib_collections) == 100
    names = sorted(F.cvrplib_collections, key=lambda n: F.parse_cvrplib_name(n)[1])
    assert names[0] == "X-n101-k25" and names[-1] == "X-n1001-k43"
