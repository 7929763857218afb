"""Print per-kernel averages of the counters in a rocprofv3 --pmc result database."""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else '%'
for r in c.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection "
                   "where kernel_name like ? group by kernel_name, counter_name", (pat,)):
    print(f"{r[0][:48]:48s} {r[1]:24s} {r[2]:16.1f}  n={r[3]}")
