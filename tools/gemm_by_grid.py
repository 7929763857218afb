"""Durations of one kernel's launches grouped by grid size, from a rocprofv3 results database.
Usage: python tools/gemm_by_grid.py <results.db> <kernel name fragment>"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
rows = c.execute(f"select d.grid_size_x / d.workgroup_size_x, count(*), avg(d.end-d.start)/1e3, min(d.end-d.start)/1e3 from {kd} d "
                 f"join {ks} s on d.kernel_id = s.id where s.kernel_name like ? group by 1 order by 1", (f"%{sys.argv[2]}%",)).fetchall()
print("workgroups  calls   avg us   min us")
for r in rows:
    print(f"{r[0]:10d} {r[1]:6d} {r[2]:8.2f} {r[3]:8.2f}")
