#!/usr/bin/env python3
"""profiles/<name>_traffic_pmc.json from a tools/run_bench_traffic.sh directory: the coarse and the fine MLP launch of
the timed step (the last two launches of the kernel), FETCH_SIZE doubled per MI355X_MICROARCH.md, both counters in KB.

    python tools/bench_traffic_digest.py gpurun_out/<tag> <kernel-substring> <dtype> [composite.json] > profiles/r03_<dtype>_traffic_pmc.json

With a fourth argument the compositing / resampling launches of the same run go to that file (the fine compositing
launch is bench.py's `roofline_composite.traffic`).
"""
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_amd.source_hash import kernel_source_sha256  # noqa: E402

root, pat, dtype = sys.argv[1:4]
R, NC, NF = 640000, 64, 192


def launches(name, counter):
    path = glob.glob(os.path.join(root, name, "*counter_collection.csv"))[0]
    rows = [r for r in csv.DictReader(open(path)) if pat in r["Kernel_Name"] and r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), float(r["Counter_Value"])) for r in rows[-2:]]


fetch, write = launches("fetch", "FETCH_SIZE"), launches("write", "WRITE_SIZE")
out = {"command": f"rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python bench.py --dtype {dtype} --steps 1 "
                  "--warmup 1 --no-cpu-baseline --no-also",
       "kernel": pat,
       "kernel_source_sha256": kernel_source_sha256(dtype),   # bench.py reports this profile only for these sources
       "unit_note": "rocprofv3 reports KB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B read requests "
                    "at 64 B), WRITE_SIZE exact"}
tot = 0.0
for i, (name, n) in enumerate((("coarse_launch", NC), ("fine_launch", NF))):
    hbm = (fetch[i][1] * 2 + write[i][1]) * 1024
    out[name] = {"ns": fetch[i][0], "FETCH_SIZE_KB": fetch[i][1], "WRITE_SIZE_KB": write[i][1], "hbm_bytes": hbm,
                 "algorithmic_bytes": R * (n * 16 + 24 + (n * 4 if i else 0))}   # the coarse depths are one shared table
    tot += hbm
out["traffic_bytes_per_launch_mean"] = tot / 2
out["note"] = "the timed step's two launches (the warm-up step's coarse launch also fetches the cold weight stream)"
json.dump(out, sys.stdout, indent=1)


if len(sys.argv) > 4:
    def one(kernel, algorithmic):
        rec = {}
        for name, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
            path = glob.glob(os.path.join(root, name, "*counter_collection.csv"))[0]
            rows = [r for r in csv.DictReader(open(path)) if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter]
            rows.sort(key=lambda r: int(r["Dispatch_Id"]))
            r = rows[-1]
            rec[counter + "_KB"] = float(r["Counter_Value"])
            rec["ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        rec["hbm_bytes"] = (rec["FETCH_SIZE_KB"] * 2 + rec["WRITE_SIZE_KB"]) * 1024
        rec["algorithmic_bytes"] = algorithmic
        return rec

    comp = {"command": out["command"], "unit_note": out["unit_note"] + "; figures of the timed (second) step",
            "kernel_source_sha256": kernel_source_sha256("composite"),
            "composite_kernel<3> (fine pass)": one("composite_kernel<3>", R * (NF * 20 + 20)),
            "composite_kernel<1> (coarse pass)": one("composite_kernel<1>", R * (NC * 20 + 20)),
            "sample_fine_kernel<2>": one("sample_fine_kernel<2>", R * 2048)}
    comp["traffic_bytes_fine_composite"] = comp["composite_kernel<3> (fine pass)"]["hbm_bytes"]
    with open(sys.argv[4], "w") as f:
        json.dump(comp, f, indent=1)
