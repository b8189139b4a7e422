#!/usr/bin/env python3
"""GPU timeline of one steady-state step from a rocprofv3 --kernel-trace csv: start offset, duration and the idle gap
before every kernel between two occurrences of a marker kernel.
    python tools/debug/step_timeline.py gpurun_out/kt3 raygen_kernel"""
import csv
import glob
import sys

root, marker = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open(glob.glob(root + "/**/*kernel_trace.csv", recursive=True)[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
a, b = idx[-4], idx[-3]
t0 = int(rows[a]["Start_Timestamp"])
prev = t0
busy = gaps = 0.0
for r in rows[a:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:56]
    print(f"{(s - t0) / 1000:9.1f} us  dur {(e - s) / 1000:7.1f}  gap {(s - prev) / 1000:7.1f}  {name}")
    if r is not rows[b]:
        busy += (e - s) / 1000
        gaps += max(0.0, (s - prev) / 1000)
    prev = e
steps = [(int(rows[idx[i + 1]]["Start_Timestamp"]) - int(rows[idx[i]]["Start_Timestamp"])) / 1000 for i in range(len(idx) - 8, len(idx) - 1)]
print(f"busy {busy:.1f} us, idle {gaps:.1f} us; step-to-step {steps}")
