#!/usr/bin/env python3
"""Runs of non-MFMA VALU instructions between two MFMAs of a kernel (static code of the built library, no GPU needed):

    python tools/kernel_lumps.py <kernel-name-substring> [min_valu_in_run, default 40]

prints, per run of at least that many VALU instructions, its address, length, the most frequent opcodes and whether a branch
sits inside -- work that cannot hide in an MFMA gap because no MFMA is in flight beside it (NRF_LIB: another build)."""
import collections
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kernel_loop_mix as k  # noqa: E402

want = sys.argv[1]
least = int(sys.argv[2]) if len(sys.argv) > 2 else 40
for text in k.disassembly():
    for name, ins in k.functions(text).items():
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        if want not in dem:
            continue
        print(dem[:110])
        n_mfma = sum("mfma" in t for _, t in ins)
        run, start, total = [], None, 0
        for a, t in ins:
            op = t.split()[0]
            if "mfma" in op:
                v = [x for x in run if x.startswith("v_")]
                if len(v) >= least and start is not None:
                    c = collections.Counter(x.replace("_e32", "").replace("_e64", "") for x in v)
                    br = sum(x.startswith(("s_cbranch", "s_branch")) for x in run)
                    total += len(v)
                    print(f"   after {start:x}: {len(v)} VALU, {br} branches  " + "  ".join(f"{o} {n}" for o, n in c.most_common(6)))
                run, start = [], a
            else:
                run.append(op)
        print(f"   {n_mfma} MFMAs in the kernel's code, {total} VALU instructions in such runs")
