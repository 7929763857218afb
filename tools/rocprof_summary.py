"""Write the per-kernel summary (calls, total/average duration in us, share) of a rocprofv3
--kernel-trace --stats result database as CSV."""
import csv, sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "Percentage"])
    for r in rows:
        w.writerow([r[0], r[1], f"{r[2]:.3f}", f"{r[3]:.3f}", f"{r[4]:.3f}"])
