"""Bitwise comparison of dgl_policy_forward between library builds (DGL_AMD_LIB): every build must print the same
digests for the TSP and CVRP rows (scheduling changes of the policy kernel must not change a single bit)."""
import hashlib, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
CHILD = r'''
import hashlib, os, sys, torch
sys.path.insert(0, os.path.join(%r, '..'))
from dgl_amd._lib import PROBLEM_TSP, PROBLEM_CVRP
from dgl_amd.engine import PackedPolicy, TspEngine, CvrpEngine
from dgl_amd.weights import random_state_dict
dev = torch.device('cuda:0')
out = []
for R in (512, 37):
    g = torch.Generator().manual_seed(7 + R)
    eng = TspEngine(torch.rand(R, 300, 2, generator=g).to(dev), 1, 30)
    pol = PackedPolicy(random_state_dict('TSP', 1), PROBLEM_TSP, dev)
    eng.reset(); eng.step(torch.zeros(R, dtype=torch.int32, device=dev))
    for t in range(3):
        eng.forward(pol, want_logits=True, want_prob=True)
        out += [eng.logits.cpu().numpy().tobytes(), eng.prob.cpu().numpy().tobytes(), eng.selected.cpu().numpy().tobytes()]
        eng.step(eng.selected)
    cxy = torch.rand(R, 301, 2, generator=g)
    dem = torch.randint(1, 10, (R, 301), generator=g) / 50.0
    dem[:, 0] = 0
    ce = CvrpEngine(cxy.to(dev), dem.to(dev), 1, 15, 15)
    cpol = PackedPolicy(random_state_dict('CVRP', 1), PROBLEM_CVRP, dev)
    ce.reset(); ce.step(torch.full((R,), 5 + 2 + 300, dtype=torch.int32, device=dev))
    for t in range(3):
        ce.forward(cpol, want_logits=True, want_prob=True)
        out += [ce.logits.cpu().numpy().tobytes(), ce.prob.cpu().numpy().tobytes(), ce.selected.cpu().numpy().tobytes()]
        ce.step(ce.selected)
print(hashlib.sha256(b''.join(out)).hexdigest(), ' '.join(hashlib.sha256(o).hexdigest()[:6] for o in out) if os.environ.get('CMP_VERBOSE') else '')
''' % HERE
digests = {}
for lib in sys.argv[1].split(","):
    env = dict(os.environ, DGL_AMD_LIB=os.path.join(HERE, "..", "dgl_amd", "lib", lib))
    res = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    digests[lib] = res.stdout.strip() if res.returncode == 0 else "FAILED " + res.stderr[-400:]
    print(lib, digests[lib], flush=True)
sys.exit(0 if len(set(digests.values())) == 1 and not any(d.startswith("FAILED") for d in digests.values()) else 1)
