#!/usr/bin/env python
"""Group a rocprofv3 kernel_stats.csv by kernel family: python tools/stats_groups.py <kernel_stats.csv> [steps]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
groups = {}
for r in rows:
    n = r["Name"]
    m = re.search(r"(?:anonymous namespace\)::|native::)?([A-Za-z_0-9]+)(<[^>]*>)?\(", n)
    key = m.group(1) if m else n[:40]
    if key in ("vectorized_elementwise_kernel", "elementwise_kernel", "unrolled_elementwise_kernel", "reduce_kernel"):
        f = re.search(r"native::(\w+)[<,]", n[n.find("kernel<") + 7:])
        key = "aten:" + (f.group(1) if f else key)
    g = groups.setdefault(key, [0, 0.0])
    g[0] += int(r["Calls"])
    g[1] += float(r["TotalDurationNs"])
tot = sum(v[1] for v in groups.values())
for k, (c, t) in sorted(groups.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{k[:60]:60s} calls/step {c / steps:7.1f}  {t / 1e6 / steps:8.3f} ms/step {100 * t / tot:5.1f}%")
print(f"total {tot / 1e6 / steps:.3f} ms/step")
