#!/usr/bin/env python
"""Register / scratch summary of the kernels in a `-Rpass-analysis=kernel-resource-usage` log (development aid):
    hipcc ... -Rpass-analysis=kernel-resource-usage -c conv2d_h8.hip 2> log ; python tools/h8_resources.py log [name-substring ...]"""
import collections
import re
import sys

txt = open(sys.argv[1]).read()
rows = {}
for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
    name = b.split("\n")[0].split(" ")[0]
    g = lambda pat: int(re.search(pat, b).group(1))
    rows[name] = (g(r"ScratchSize \[bytes/lane\]: (\d+)"), g(r" VGPRs: (\d+)"), g(r"AGPRs: (\d+)"), g(r"SGPRs: (\d+)"), g(r"Occupancy \[waves/SIMD\]: (\d+)"))
print(len(rows), "kernels; scratch histogram:", dict(collections.Counter(v[0] for v in rows.values())))
for pat in sys.argv[2:]:
    for k, v in rows.items():
        if pat in k:
            print(f"scratch {v[0]:4d}  vgpr {v[1]:3d}  agpr {v[2]:3d}  sgpr {v[3]:3d}  occ {v[4]}  {k}")
