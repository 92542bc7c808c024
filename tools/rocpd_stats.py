#!/usr/bin/env python
"""Kernel statistics (the table `rocprofv3 --kernel-trace --stats` prints) from the rocpd SQLite file it writes by default:
    python tools/rocpd_stats.py gpurun_out/prof/run_results.db profiles/rNN/kernel_stats.csv"""
import csv
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
rows = con.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name order by sum(duration) desc").fetchall()
total = sum(r[2] for r in rows)
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for name, calls, tot, avg, mn, mx in rows:
        w.writerow([name, calls, int(tot), round(avg, 1), round(100.0 * tot / total, 4), int(mn), int(mx)])
print(f"{len(rows)} kernels, {total / 1e6:.3f} ms of kernel time")
