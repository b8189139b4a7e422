#!/usr/bin/env python3
"""Instruction mix of a kernel's hot loop, from the built library's gfx950 code (no GPU needed).

    python tools/kernel_loop_mix.py <kernel-name-substring> [min_mfma_in_loop] [--ops]

Disassembles nerf_amd/lib/libnerf_amd.so, finds for every matching kernel the smallest loop (backward branch) that
holds at least `min_mfma_in_loop` MFMAs (default 8) and prints its instruction classes; --ops lists the VALU opcodes,
--seq prints the number of VALU instructions between consecutive MFMAs (how the two pipes are interleaved).
This is the static count behind statements like "540 VALU instructions per 32-point tile" in DESIGN.md."""
import collections
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def disassembly():
    blob = open(os.environ.get("NRF_LIB", os.path.join(ROOT, "nerf_amd", "lib", "libnerf_amd.so")), "rb").read()   # NRF_LIB: another build (A/B variants)
    with tempfile.TemporaryDirectory() as tmp:
        i = 0
        for m in re.finditer(MAGIC, blob):
            p = m.start()
            (n,) = struct.unpack_from("<Q", blob, p + 24)
            o = p + 32
            for _ in range(n):
                off, size, tl = struct.unpack_from("<QQQ", blob, o)
                triple = blob[o + 24:o + 24 + tl]
                o += 24 + tl
                if b"gfx950" in triple and size:
                    f = os.path.join(tmp, f"co{i}.o")
                    open(f, "wb").write(blob[p + off:p + off + size])
                    i += 1
                    yield subprocess.run([OBJDUMP, "-d", "--mcpu=gfx950", f], check=True, capture_output=True, text=True).stdout


def functions(text):
    funcs, cur = {}, None
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            cur = funcs.setdefault(m.group(1), [])
            continue
        m = re.match(r"^\s+(\S.*?)\s*//\s*([0-9A-Fa-f]+):", line)
        if m and cur is not None:
            cur.append((int(m.group(2), 16), m.group(1)))
    return funcs


def classify(op):
    if "mfma" in op:
        return "mfma"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_nop"):
        return "s_nop"
    if op.startswith("s_"):
        return "salu"
    return op


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    want, min_mfma = args[0], int(args[1]) if len(args) > 1 else 8
    for text in disassembly():
        for name, ins in functions(text).items():
            dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
            if want not in dem:
                continue
            idx = {a: i for i, (a, _) in enumerate(ins)}
            best = None
            for i, (a, t) in enumerate(ins):
                op = t.split()[0]
                if op.startswith("s_cbranch") or op == "s_branch":
                    imm = int(t.split()[1])
                    tgt = a + 4 + 4 * (imm - 65536 if imm >= 32768 else imm)
                    if tgt < a and tgt in idx:
                        b = idx[tgt]
                        if sum("mfma" in t2 for _, t2 in ins[b:i + 1]) >= min_mfma and (best is None or i - b < best[1] - best[0]):
                            best = (b, i)
            if best is None:
                continue
            body = ins[best[0]:best[1] + 1]
            c = collections.Counter(classify(t.split()[0]) for _, t in body)
            print(f"{dem[:100]}\n   loop of {len(body)} instructions: " + "  ".join(f"{k} {v}" for k, v in sorted(c.items(), key=lambda kv: -kv[1])))
            if "--seq" in sys.argv:   # the loop as a string: M = MFMA, then the count of VALU (and other) instructions up to the next one
                seq, run = [], 0
                for _, t in body:
                    if "mfma" in t:
                        seq.append(str(run)), seq.append("M")
                        run = 0
                    elif t.startswith("v_"):
                        run += 1
                print("   " + " ".join(seq + [str(run)]))
            if "--ops" in sys.argv:
                ops = collections.Counter(t.split()[0].replace("_e32", "").replace("_e64", "") for _, t in body if t.startswith("v_") and "mfma" not in t)
                print("   " + "  ".join(f"{k} {v}" for k, v in ops.most_common(30)))


if __name__ == "__main__":
    main()
