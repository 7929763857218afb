"""Print the measured peaks of this card (the helper bench.py uses)."""
import os, sys, json, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench
print(json.dumps(bench.measured_peaks(torch.device("cuda:0")), indent=1))
