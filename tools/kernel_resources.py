#!/usr/bin/env python3
"""Registers, scratch and LDS of every kernel in the built library, from the gfx950 code objects' metadata notes.

    python tools/kernel_resources.py [substring ...]

No GPU needed (reads nerf_amd/lib/libnerf_amd.so).  A kernel with scratch (`private_segment_fixed_size` > 0) spills: in
the kernels that prefetch with counted waits a scratch reload's vmcnt(0) drains the queue."""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(blob):
    for m in re.finditer(MAGIC, blob):
        p = m.start()
        (n,) = struct.unpack_from("<Q", blob, p + 24)
        o = p + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, o)
            triple = blob[o + 24:o + 24 + tl]
            o += 24 + tl
            if b"gfx950" in triple and size:
                yield blob[p + off:p + off + size]


def main():
    want = sys.argv[1:]
    blob = open(os.environ.get("NRF_LIB", os.path.join(ROOT, "nerf_amd", "lib", "libnerf_amd.so")), "rb").read()   # NRF_LIB: another build (A/B variants)
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        for i, obj in enumerate(code_objects(blob)):
            f = os.path.join(tmp, f"co{i}.o")
            open(f, "wb").write(obj)
            notes = subprocess.run([READELF, "--notes", f], check=True, capture_output=True, text=True).stdout
            for blk in notes.split("- .agpr_count:")[1:]:
                def field(name):
                    m = re.search(r"\." + name + r":\s+(\S+)", blk)
                    return m.group(1) if m else "?"
                agpr = blk.split()[0]
                name = subprocess.run(["c++filt", field("name")], capture_output=True, text=True).stdout.strip()
                name = re.sub(r"^void \(anonymous namespace\)::", "", name)
                rows.append((name, field("vgpr_count"), agpr, field("sgpr_count"), field("private_segment_fixed_size"),
                             field("group_segment_fixed_size")))
    print("%-78s %5s %5s %5s %8s %8s" % ("kernel", "vgpr", "agpr", "sgpr", "scratch", "lds"))
    for r in sorted(rows):
        if not want or any(w in r[0] for w in want):
            print("%-78s %5s %5s %5s %8s %8s" % ((r[0][:78],) + r[1:]))


if __name__ == "__main__":
    main()
