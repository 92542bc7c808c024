#!/usr/bin/env python
"""Per-kernel table from rocprofv3 --pmc output directories (CSV or rocpd .db): `python tools/pmc_table.py DIR [DIR ...]`.
Joins counter_collection.csv (summing a counter over its dimensions per dispatch) with kernel_trace.csv durations and
prints, per dispatch of the LAST iteration, duration and every collected counter."""
import csv
import glob
import os
import sys
from collections import OrderedDict, defaultdict


def load_rocpd(db):
    """The same three maps from a rocpd SQLite file (rocprofv3's default output when no --output-format is given)."""
    import sqlite3
    con = sqlite3.connect(db)
    dur, name = {}, {}
    vals = defaultdict(lambda: defaultdict(float))
    for did, nm, cn, v, t0, t1 in con.execute("select dispatch_id, kernel_name, counter_name, value, start, end from counters_collection"):
        dur[did] = (t1 - t0) / 1e3
        name[did] = nm
        vals[did][cn] += float(v)
    return dur, name, vals


def load(d):
    if not glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        return load_rocpd(glob.glob(os.path.join(d, "**", "*.db"), recursive=True)[0])
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
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    json_out = next((a[len("--json="):] for a in sys.argv[1:] if a.startswith("--json=")), None)
    meta_images = next((int(a[len("--meta-images="):]) for a in sys.argv[1:] if a.startswith("--meta-images=")), None)
    for d in args:
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
    starts = [k for k, row in rows.items() if "nchw_to_h8_kernel" in row["name"]]     # first kernel of an f16 forward
    if starts:
        half = starts[-1]
    if json_out:
        # per kernel name, over the dispatches of the LAST iteration: launches, average duration and counters per launch
        import json
        import re
        agg = {}
        for k, row in rows.items():
            if k < half:
                continue
            nm = row["name"].replace("(anonymous namespace)::", "").replace("void ", "")
            nm = re.sub(r"\(.*$", "", nm)
            a = agg.setdefault(nm, {"launches": 0, "dur_us": 0.0})
            a["launches"] += 1
            a["dur_us"] += sum(row["dur"]) / len(row["dur"])
            for c in counters:
                a[c] = a.get(c, 0.0) + row.get(c, 0.0)
        # the dispatches of the last iteration in order (bench.py maps its timed launches onto them: per-launch traffic, encoder totals)
        disp = []
        for k, row in rows.items():
            if k < half:
                continue
            nm = re.sub(r"\(.*$", "", row["name"].replace("(anonymous namespace)::", "").replace("void ", ""))
            d = {"name": nm, "dur_us": round(sum(row["dur"]) / len(row["dur"]), 2)}
            for c in counters:
                d[c] = row.get(c, 0.0)
            disp.append(d)
        for a in agg.values():
            n = a["launches"]
            for key in list(a):
                if key != "launches":
                    a[key] = a[key] / n
        if meta_images is not None:
            # bench.py attaches these numbers only to the build of libslu_hip.so they were measured with
            import hashlib
            lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "semanticlidarunc_amd", "libslu_hip.so")
            agg["_meta"] = {"images": meta_images, "lib_sha256": hashlib.sha256(open(lib, "rb").read()).hexdigest(),
                            "command": "rocprofv3 --pmc FETCH_SIZE (and, separately, WRITE_SIZE) --kernel-trace -- python3 tools/prof_forward.py "
                                       f"{meta_images} 2 mc with SLU_CONV_PRECISION=f16 (the launches of one default bench.py step)",
                            "units": "FETCH_SIZE / WRITE_SIZE in KB per launch (average over the launches of the last MC step); gfx950: double "
                                     "FETCH_SIZE for 16 B/lane loads (MI355X_MICROARCH.md, HBM)"}
        agg["_dispatches"] = disp
        with open(json_out, "w") as f:
            json.dump(agg, f, indent=1, sort_keys=True)
    print("dur_us " + " ".join(counters) + " kernel")
    for k, row in rows.items():
        if k < half:
            continue
        nm = row["name"]
        nm = nm[nm.find("conv"):] if "conv" in nm else nm
        print(f"{sum(row['dur'])/len(row['dur']):8.1f} " + " ".join(f"{row.get(c, float('nan')):.4g}" for c in counters) + "  " + nm[:90])


if __name__ == "__main__":
    main()
