#!/usr/bin/env python
"""Per-kernel table from rocprofv3 --pmc CSV output directories: `python tools/pmc_table.py DIR [DIR ...]`.
Joins counter_collection.csv (summing a counter over its dimensions per dispatch) with kernel_trace.csv durations and
prints, per dispatch of the LAST iteration, duration and every collected counter."""
import csv
import glob
import os
import sys
from collections import OrderedDict, defaultdict


def load(d):
    cc = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    dur, name = {}, {}
    for r in csv.DictReader(open(kt)):
        did = int(r["Dispatch_Id"])
        dur[did] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        name[did] = r["Kernel_Name"]
    vals = defaultdict(lambda: defaultdict(float))
    for r in csv.DictReader(open(cc)):
        vals[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    return dur, name, vals


def main():
    rows = OrderedDict()
    counters = []
    for d in sys.argv[1:]:
        dur, name, vals = load(d)
        ids = sorted(dur)
        for k, did in enumerate(ids):           # dispatch ids line up across runs of the same deterministic program
            row = rows.setdefault(k, {"name": name[did], "dur": []})
            row["dur"].append(dur[did])
            for c, v in vals.get(did, {}).items():
                row[c] = v
                if c not in counters:
                    counters.append(c)
    half = len(rows) // 2
    print("dur_us " + " ".join(counters) + " kernel")
    for k, row in rows.items():
        if k < half:
            continue
        nm = row["name"]
        nm = nm[nm.find("conv"):] if "conv" in nm else nm
        print(f"{sum(row['dur'])/len(row['dur']):8.1f} " + " ".join(f"{row.get(c, float('nan')):.4g}" for c in counters) + "  " + nm[:90])


if __name__ == "__main__":
    main()
