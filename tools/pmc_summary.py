#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs: one JSON object per dispatch of kernels matching a pattern."""
import collections
import csv
import json
import sys

pat = sys.argv[1]
agg = collections.OrderedDict()
for path in sys.argv[2:]:
    for r in csv.DictReader(open(path)):
        if pat not in r["Kernel_Name"]:
            continue
        key = (path.split("/")[-1].split("_counter")[0], r["Dispatch_Id"])
        d = agg.setdefault(key, {"kernel": r["Kernel_Name"][:60], "grid": r["Grid_Size"], "wg": r["Workgroup_Size"],
                                 "lds": r["LDS_Block_Size"], "scratch": r["Scratch_Size"], "vgpr": r["VGPR_Count"],
                                 "agpr": r["Accum_VGPR_Count"], "sgpr": r["SGPR_Count"],
                                 "ns": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
        d[r["Counter_Name"]] = float(r["Counter_Value"])
print(json.dumps([{"pass": k[0], "dispatch": k[1], **v} for k, v in agg.items()], indent=1))
