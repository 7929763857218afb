"""Fixtures for the file formats: small synthetic files (TSPLIB .tsp, CVRPLIB .vrp, instance text,
solution text) written here, and what the imported REFERENCE's own readers and scoring helpers return
for them (utils/utils.py:266-438, CVRP/CVRPEnv.py:54-111, TSP/TSPEnv.py:82-106).  Build container only.
Writes tests/golden/formats/* and tests/golden/formats/expected.json.

usage: python oracle/make_format_goldens.py
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_goldens import _import_reference  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "formats")


def main():
    import torch
    _import_reference()
    import utils.utils as U
    from CVRP.CVRPEnv import read_cvrp_instances_from_file, read_solutions_from_file
    from TSP.TSPEnv import TSPEnv
    os.makedirs(os.path.join(OUT, "tsplib"), exist_ok=True)
    os.makedirs(os.path.join(OUT, "vrplib"), exist_ok=True)
    rng = np.random.RandomState(5)
    exp = {}
    # TSPLIB-style file
    pts = rng.randint(0, 2000, size=(17, 2))
    with open(os.path.join(OUT, "tsplib", "syn17.tsp"), "w") as f:
        f.write("NAME : syn17\nCOMMENT : synthetic fixture\nTYPE : TSP\nDIMENSION : 17\nEDGE_WEIGHT_TYPE : EUC_2D\n"
                "NODE_COORD_SECTION\n")
        for i, (x, y) in enumerate(pts):
            f.write(f"{i + 1} {x} {y}.5\n" if i % 3 == 0 else f"{i + 1}  {x}   {y}\n")
        f.write("EOF\n")
    nodes, name = U.read_tsplib_file(os.path.join(OUT, "tsplib", "syn17.tsp"))
    inst = torch.tensor(nodes)
    tour = torch.from_numpy(rng.permutation(17))
    dist = U.get_dist_matrix(inst)
    exp["tsplib"] = {"name": name, "nodes": nodes, "normalized": U.normalize_nodes_to_unit_board(inst).tolist(),
                     "tour": tour.tolist(), "tour_len": U.calculate_tour_length_by_dist_matrix(dist, tour).item(),
                     "parse_name": list(U.parse_tsplib_name("syn17")),
                     "bsz": [U.choose_bsz(s) for s in (100, 200, 201, 1000, 1001, 5000, 5001)]}
    # CVRPLIB-style file (tab-separated as the X set is)
    cpts = rng.randint(0, 1000, size=(12, 2))
    dem = rng.randint(1, 30, size=12)
    dem[0] = 0
    with open(os.path.join(OUT, "vrplib", "S-n12-k3.vrp"), "w") as f:
        f.write("NAME : \tS-n12-k3\nCOMMENT : \t\"synthetic fixture\"\nTYPE : \tCVRP\nDIMENSION : \t12\n"
                "EDGE_WEIGHT_TYPE : \tEUC_2D\nCAPACITY : \t60\nNODE_COORD_SECTION\t\t\n")
        for i, (x, y) in enumerate(cpts):
            f.write(f"{i + 1}\t{x}\t{y}\n")
        f.write("DEMAND_SECTION\t\t\n")
        for i, d in enumerate(dem):
            f.write(f"{i + 1}\t{d}\n")
        f.write("DEPOT_SECTION\n\t1\n\t-1\nEOF\n")
    depot, cnodes, demands, props = U.read_cvrplib_file(os.path.join(OUT, "vrplib", "S-n12-k3.vrp"))
    exp["cvrplib"] = {"depot": depot, "nodes": cnodes, "demands": demands, "capacity": int(props["CAPACITY"]),
                      "name": props["NAME"], "parse_name": list(U.parse_cvrplib_name("S-n12-k3"))}
    # label -> route -> length as CVRP/test_cvrplib.py:237-255 does it
    label = torch.from_numpy(rng.permutation(11) + 1)
    flags = torch.tensor([1, 0, 0, 0, 1, 0, 0, 1, 0, 0, 0])
    route = []
    for i in range(label.shape[0]):
        if flags[i] == 1:
            route.append(torch.tensor(11))
        route.append(label[i] - 1)
    dn = torch.cat((torch.tensor(cnodes), torch.tensor(depot).unsqueeze(0)), dim=0)
    exp["cvrplib"].update({"label": label.tolist(), "flags": flags.tolist(), "route": [int(v) for v in route],
                           "route_len": U.calculate_tour_length_by_dist_matrix(U.get_dist_matrix(dn), torch.tensor(route)).item()})
    # instance text, CVRP
    with open(os.path.join(OUT, "cvrp_instances.txt"), "w") as f:
        for b in range(3):
            xy = rng.rand(6, 2).astype(np.float32)
            d = rng.randint(1, 10, size=5)
            f.write(f"{xy[0][0]},{xy[0][1]} .|. " + " ".join(f"{x},{y}" for x, y in xy[1:]) + " .|. " +
                    " ".join(str(v) for v in d) + f" .|. {20 + b}\n")
    dd, nn, de = read_cvrp_instances_from_file(os.path.join(OUT, "cvrp_instances.txt"))
    exp["cvrp_text"] = {"depot": dd.tolist(), "nodes": nn.tolist(), "demands": de.tolist()}
    # solution text (ragged tours)
    with open(os.path.join(OUT, "solutions.txt"), "w") as f:
        f.write("0,3,1,2,0,5,4,0 3.25 0.5\n0,1,2,3,4,5,0 2.75 1.25\n")
    t, l, s = read_solutions_from_file(os.path.join(OUT, "solutions.txt"))
    exp["solutions"] = {"tours": t.tolist(), "lens": l.tolist(), "secs": s.tolist()}
    # instance text, TSP, and its baseline file
    xy = rng.rand(2, 5, 2).astype(np.float32)
    with open(os.path.join(OUT, "tsp_instances.txt"), "w") as f:
        for b in range(2):
            for x, y in xy[b]:
                f.write(f"{x},{y} ")
    with open(os.path.join(OUT, "tsp_baseline.txt"), "w") as f:
        f.write("0 2.5 0.1\n1 3.125 0.2\n")
    env = TSPEnv(mode="test", test_mode="pomo_test", aug_size=1, data_path=os.path.join(OUT, "tsp_instances.txt"),
                 baseline_path=os.path.join(OUT, "tsp_baseline.txt"), append_information=[True] * 11, aggregation_nums=5)
    env.load_data(2, 5)
    exp["tsp_text"] = {"nodes": env.raw_data_nodes.tolist(), "baseline": env.baseline_len.tolist()}
    with open(os.path.join(OUT, "expected.json"), "w") as f:
        json.dump(exp, f, indent=1)
    print("wrote", OUT, sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
