#!/usr/bin/env python3
"""Digest of a tools/run_pmc.sh output directory: per kernel, the slowest dispatch of each pass merged into one record
with the derived figures the DESIGN quotes (MFMA busy share, effective clock, wait shares, HBM traffic with the
MI355X_MICROARCH.md corrections: FETCH_SIZE x2 on gfx950, both in KB).

    python tools/pmc_digest.py gpurun_out/r02_bf16 [kernel-substring ...] [--family=kilo] > profiles/r02_bf16_pmc.json
"""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
root = sys.argv[1]
pats = [a for a in sys.argv[2:] if not a.startswith("--family=")]
family = next((a.split("=", 1)[1] for a in sys.argv[2:] if a.startswith("--family=")), None)   # nerf_amd/source_hash.py
rows = collections.defaultdict(dict)      # (kernel, pass) -> {dispatch: {counter: value, meta}}
for path in sorted(glob.glob(os.path.join(root, "pmc*", "*counter_collection.csv"))):
    p = os.path.basename(os.path.dirname(path))
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if pats and not any(s in k for s in pats):
            continue
        d = rows[(k, p)].setdefault(r["Dispatch_Id"], {"ns": int(r["End_Timestamp"]) - int(r["Start_Timestamp"]),
                                                       "grid": int(r["Grid_Size"]), "wg": int(r["Workgroup_Size"]),
                                                       "lds": int(r["LDS_Block_Size"]), "scratch": int(r["Scratch_Size"]),
                                                       "vgpr": int(r["VGPR_Count"]), "agpr": int(r["Accum_VGPR_Count"]),
                                                       "sgpr": int(r["SGPR_Count"])})
        d[r["Counter_Name"]] = float(r["Counter_Value"])
out = {}
for (k, p), disp in rows.items():
    e = max(disp.values(), key=lambda d: d["ns"])            # the longest launch of the kernel in this pass
    short = k.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
    rec = out.setdefault(short, {"launches_seen": len(disp)})
    for key in ("grid", "wg", "lds", "scratch", "vgpr", "agpr", "sgpr"):
        rec[key] = e[key]
    rec.setdefault("ns_by_pass", {})[p] = e["ns"]
    for c, v in e.items():
        if c[0].isupper():
            rec[c] = v
for short, r in out.items():
    d = {}
    ns = r["ns_by_pass"].get("pmc1")
    if ns and "GRBM_GUI_ACTIVE" in r:
        cyc = r["GRBM_GUI_ACTIVE"] / 8.0                      # summed over the 8 XCDs
        d["cycles"] = cyc
        d["effective_clock_GHz"] = cyc / ns
        if r.get("SQ_VALU_MFMA_BUSY_CYCLES"):
            d["mfma_busy_share_of_cycles"] = r["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / cyc   # 1024 SIMDs
    if r.get("SQ_WAVE_CYCLES"):
        w = r["SQ_WAVE_CYCLES"]
        for c, name in (("SQ_WAIT_ANY", "wave_time_parked_in_waitcnt_or_barrier"), ("SQ_WAIT_INST_ANY", "wave_time_issue_stalled"),
                        ("SQ_ACTIVE_INST_ANY", "wave_time_issuing")):
            if c in r:
                d[name] = r[c] / w
    if "FETCH_SIZE" in r or "WRITE_SIZE" in r:
        d["hbm_read_bytes"] = 2.0 * 1024.0 * r.get("FETCH_SIZE", 0.0)
        d["hbm_write_bytes"] = 1024.0 * r.get("WRITE_SIZE", 0.0)
        d["hbm_bytes"] = d["hbm_read_bytes"] + d["hbm_write_bytes"]
    if r.get("TCC_HIT_sum") is not None and r.get("TCC_MISS_sum") is not None:
        d["l2_hit_rate"] = r["TCC_HIT_sum"] / max(1.0, r["TCC_HIT_sum"] + r["TCC_MISS_sum"])
    r["derived"] = d
doc = {"source": root, "note": "longest dispatch per kernel per pass; counters are chip totals", "kernels": out}
if family:
    from nerf_amd.source_hash import kernel_source_sha256
    doc["kernel_source_sha256"] = kernel_source_sha256(family)   # bench.py reports figures of this profile only for these sources
json.dump(doc, sys.stdout, indent=1)
