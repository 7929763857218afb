"""Per-kernel totals from a rocprofv3 results database (the default output format): calls, total and average
duration, share of the GPU time.  Usage: python tools/rocpd_stats.py <results.db> [name filter]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
cols = [r[1] for r in c.execute(f"pragma table_info({ks})")]
name = 'kernel_name' if 'kernel_name' in cols else 'display_name'
rows = c.execute(f"select s.{name}, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start) from {kd} d join {ks} s "
                 f"on d.kernel_id = s.id group by s.{name} order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
flt = sys.argv[2] if len(sys.argv) > 2 else ''
print(f"{'kernel':60s} {'calls':>7s} {'total ms':>10s} {'avg us':>9s} {'min us':>9s} {'%':>6s}")
for n, k, t, a, m in rows:
    if flt in n:
        print(f"{n[:60]:60s} {k:7d} {t / 1e6:10.3f} {a / 1e3:9.2f} {m / 1e3:9.2f} {100 * t / tot:6.2f}")
