"""Counters of tools/bench_step_parts.py under rocprofv3 --pmc: averages per launch for the two modes (state update
only / whole step) of the step kernel, told apart by launch order (33 launches per mode, four runs)."""
import sqlite3, sys
from collections import defaultdict
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select dispatch_id, counter_name, sum(value) from counters_collection where kernel_name like '%step_kernel%' "
                 "group by dispatch_id, counter_name order by dispatch_id").fetchall()
ids = sorted({r[0] for r in rows})
pos = {d: i for i, d in enumerate(ids)}
acc = defaultdict(list)
for d, name, v in rows:
    i = pos[d]
    mode = (i // 33) % 2
    if i % 33 >= 3:
        acc[(name, mode)].append(v)
for name in sorted({k[0] for k in acc}):
    a, b = acc[(name, 0)], acc[(name, 1)]
    print(f"{name:28s} update only {sum(a)/len(a):16.0f}   whole step {sum(b)/len(b):16.0f}   (n={len(a)},{len(b)})")
