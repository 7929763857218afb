"""On-disk formats of the reference and the benchmark-library scoring protocol.

  * TSP instance text: one file, points as ``x,y`` tokens separated by blanks (TSP/TSPEnv.py:82-106);
    baseline file: one line per instance whose second token is the reference length (``:98-104``);
  * CVRP instance text: one instance per line, ``depot .|. nodes .|. demands .|. capacity``
    (CVRP/CVRPEnv.py:83-111); solution text: ``tour,len time`` per line (``:54-80``);
  * TSPLIB ``.tsp`` / CVRPLIB ``.vrp`` files and the tables of best-known values
    (utils/utils.py:47-125,159-260,266-405);
  * scoring on a library instance: decide on the instance scaled to the unit square, measure the tour on the
    original coordinates, round the length up to an integer, gap = len / best-known - 1
    (TSP/test_tsplib.py:173-233, CVRP/test_cvrplib.py:175-255).
"""
from __future__ import annotations

import json
import math
import os
from pathlib import Path
from typing import Dict, List, Tuple

import torch

_DATA = Path(__file__).parent / "data"


def _table(name: str) -> Dict[str, int]:
    with open(_DATA / name, "r", encoding="utf8") as f:
        return json.load(f)


tsplib_collections: Dict[str, int] = _table("tsplib_optima.json")
cvrplib_collections: Dict[str, int] = _table("cvrplib_optima.json")


def avg_list(values):
    return sum(values) / len(values) if len(values) > 0 else 0


def parse_tsplib_name(name: str) -> Tuple[str, int]:
    """'kroA100' -> ('kroA', 100)."""
    return "".join(ch for ch in name if ch.isalpha()), int("".join(ch for ch in name if ch.isdigit()))


def parse_cvrplib_name(name: str) -> Tuple[str, int]:
    """'X-n101-k25' -> ('X', 100): the customer count excludes the depot."""
    problem_set, size, _ = name.split("-")
    return problem_set, int("".join(ch for ch in size if ch.isdigit())) - 1


def choose_bsz(size: int) -> int:
    """Augmentation factor per instance size (utils/utils.py:921-929)."""
    if size <= 200:
        return 64
    if size <= 1000:
        return 32
    if size <= 5000:
        return 16
    return 4


# ---------------------------------------------------------------------- instance / solution text
def read_tsp_instances(path, episodes: int, problem_size: int) -> torch.Tensor:
    with open(path, "r", encoding="utf8") as f:
        pts = [[float(v) for v in tok.split(",")] for tok in f.read().split()]
    return torch.tensor(pts, dtype=torch.float32).view(episodes, problem_size, 2)


def write_tsp_instances(path, instances: torch.Tensor) -> None:
    """Inverse of read_tsp_instances; ``str(float)`` round-trips float32 values exactly, as the library
    drivers of the reference write them (TSP/test_tsplib.py:203-205)."""
    flat = instances.reshape(-1, 2)
    with open(path, "w", encoding="utf8") as f:
        for x, y in flat.tolist():
            f.write(f"{x},{y} ")


def read_baseline_lengths(path) -> torch.Tensor:
    with open(path, "r", encoding="utf8") as f:
        return torch.tensor([float(line.split()[1]) for line in f if line.strip()])


def read_cvrp_instances_from_file(path):
    """-> (depot [B,2], nodes [B,N,2], demands [B,N] divided by the line's capacity in float32)."""
    depot, nodes, demands = [], [], []
    with open(path, "r", encoding="utf8") as f:
        for line in f:
            if not line.strip():
                continue
            part = line.strip().split(" .|. ")
            depot.append([float(x) for x in part[0].strip().split(",")])
            nodes.append([[float(x) for x in tok.split(",")] for tok in part[1].strip().split(" ")])
            cap = int(part[3])
            demands.append((torch.Tensor([int(x) for x in part[2].strip().split(" ")]) / cap).tolist())
    return torch.Tensor(depot), torch.Tensor(nodes), torch.Tensor(demands)


def write_cvrp_instances(path, depot: torch.Tensor, nodes: torch.Tensor, demands: torch.Tensor, capacity) -> None:
    """``demands`` integer [B,N]; ``capacity`` int or [B] (CVRP/test_cvrplib.py:191-203 writes one such line)."""
    B = nodes.shape[0]
    caps = [int(capacity)] * B if not torch.is_tensor(capacity) or capacity.dim() == 0 else [int(c) for c in capacity]
    with open(path, "w", encoding="utf8") as f:
        for b in range(B):
            f.write(f"{depot[b][0].item()},{depot[b][1].item()} .|. ")
            f.write(" ".join(f"{x},{y}" for x, y in nodes[b].tolist()))
            f.write(" .|. " + " ".join(str(int(d)) for d in demands[b].tolist()) + f" .|. {caps[b]}\n")


def read_solutions_from_file(path):
    """Lines ``n0,n1,... length seconds`` -> (tours padded with 0 [B,L], lengths [B], seconds [B])."""
    tours, lens, secs = [], [], []
    with open(path, "r", encoding="utf8") as f:
        for line in f:
            if not line.strip():
                continue
            tour_text, len_text, time_text = line.strip().split(" ")
            tours.append(torch.tensor([int(v) for v in tour_text.split(",")]))
            lens.append(float(len_text))
            secs.append(float(time_text))
    padded = torch.nn.utils.rnn.pad_sequence(tours, batch_first=True, padding_value=0)
    return padded, torch.tensor(lens), torch.tensor(secs)


def write_solutions(path, tours: List[List[int]], lengths, seconds) -> None:
    with open(path, "w", encoding="utf8") as f:
        for t, l, s in zip(tours, lengths, seconds):
            f.write(",".join(str(int(v)) for v in t) + f" {float(l)} {float(s)}\n")


# ------------------------------------------------------------------------------- library files
def _fields(line: str) -> List[str]:
    return line.replace("\t", " ").split()


def read_tsplib_file(path):
    """-> (nodes [[x, y], ...], NAME).  ``KEY : VALUE`` header lines up to the first line without a colon,
    then ``index x y`` lines; reading stops at the first blank line, as in the reference."""
    props, nodes, header = {}, [], True
    with open(path, "r", encoding="utf8") as f:
        for line in f:
            if not line.strip():
                break
            if header and ":" in line:
                key, val = [x.strip() for x in line.split(":")]
                props[key] = val
                continue
            header = False
            if line.startswith(("NODE_COORD_SECTION", "EOF")):
                continue
            _, x, y = _fields(line)
            nodes.append([float(x), float(y)])
    return nodes, props["NAME"]


def read_cvrplib_file(path):
    """-> (depot [x, y], customers [[x, y], ...], integer demands of the customers, header properties)."""
    props, pts, demands = {}, [], []
    section = "header"
    with open(path, "r", encoding="utf8") as f:
        for line in f:
            if not line.strip():
                break
            if section == "header" and ":" in line:
                key, val = [x.strip() for x in line.split(":")]
                props[key] = val
                continue
            if line.startswith("NODE_COORD_SECTION"):
                section = "nodes"
            elif line.startswith("DEMAND_SECTION"):
                section = "demands"
            elif line.startswith("DEPOT_SECTION"):
                section = "depot"
            elif line.startswith("EOF"):
                section = "done"
            elif section == "nodes":
                _, x, y = _fields(line)
                pts.append([float(x), float(y)])
            elif section == "demands":
                demands.append(int(_fields(line)[1]))
            elif section == "header":
                section = "nodes"            # a first non-header line that is no section title
    return pts[0], pts[1:], demands[1:], props


def load_tsplib_file(root, name):
    nodes, nm = read_tsplib_file(Path(root) / "tsplib" / f"{name}.tsp")
    return torch.tensor(nodes), nm


def load_cvrplib_file(root, name):
    depot, nodes, demands, props = read_cvrplib_file(Path(root) / "vrplib" / f"{name}.vrp")
    return (torch.tensor(depot), torch.tensor(nodes), torch.tensor(demands), torch.tensor(int(props["CAPACITY"])),
            props["NAME"])


# --------------------------------------------------------------------------------------- scoring
def normalize_nodes_to_unit_board(nodes: torch.Tensor) -> torch.Tensor:
    """Translate to the origin and divide by the larger extent (utils/utils.py:409-422)."""
    lo = nodes.min(dim=0).values
    factor = (nodes.max(dim=0).values - lo).max()
    return (nodes - lo) / factor


normalize_tsp_to_unit_board = normalize_nodes_to_unit_board


def get_dist_matrix(instance: torch.Tensor) -> torch.Tensor:
    return torch.norm(instance.unsqueeze(0) - instance.unsqueeze(1), p=2, dim=-1)


def calculate_tour_length_by_dist_matrix(dist_matrix: torch.Tensor, tours: torch.Tensor) -> torch.Tensor:
    if tours.dim() == 1:
        tours = tours.unsqueeze(0)
    return dist_matrix[tours, torch.roll(tours, shifts=-1, dims=1)].sum(dim=1)


def tsplib_gap(instance: torch.Tensor, tour: torch.Tensor, best_known: int) -> Tuple[int, float]:
    """Closed-tour length on the original coordinates, rounded up; gap to the best-known value
    (TSP/test_tsplib.py:229-233)."""
    length = math.ceil(calculate_tour_length_by_dist_matrix(get_dist_matrix(instance), tour.cpu().long()).item())
    return length, length / best_known - 1


def cvrplib_route(nodes_list: torch.Tensor, flags: torch.Tensor, size: int) -> torch.Tensor:
    """Label (customer ids 1..N, via-depot flags) -> closed node sequence over [customers 0..N-1, depot = N]
    (CVRP/test_cvrplib.py:237-248)."""
    out = []
    for node, flag in zip(nodes_list.tolist(), flags.tolist()):
        if flag == 1:
            out.append(size)
        out.append(node - 1)
    return torch.tensor(out)


def cvrplib_gap(depot: torch.Tensor, nodes: torch.Tensor, nodes_list: torch.Tensor, flags: torch.Tensor,
                best_known: int) -> Tuple[int, float]:
    """CVRP/test_cvrplib.py:180-181,253-255: distances over [customers..., depot], length rounded up."""
    size = nodes.shape[0]
    dist = get_dist_matrix(torch.cat((nodes, depot.unsqueeze(0)), dim=0))
    route = cvrplib_route(nodes_list.cpu(), flags.cpu(), size)
    length = math.ceil(calculate_tour_length_by_dist_matrix(dist, route).item())
    return length, length / best_known - 1
