"""Lint of the built device code: loads the compiler does not see must be waited for before their register is touched.

mlp_fp16x2_kernel and mlp_dgrad_h2_kernel (nerf_amd/csrc/mlp_fp16x2.hip) load a tile's inputs one stage ahead with
`global_load_dword` written in inline asm, into accumulation registers, and wait for them with a counted
`s_waitcnt vmcnt` that names the registers.  The compiler knows nothing of the latency: if register allocation ever
puts a copy of such a register between the load and its wait, the copy reads the old contents and the results are
wrong only sometimes.  This test reads the gfx950 code objects out of the built library and checks, in program order,
that no instruction names an accumulation register between a load into it and the next `s_waitcnt vmcnt`.
"""
import os
import re
import struct
import subprocess

import pytest

from nerf_amd import _lib

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _code_objects(blob):
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


def _agprs(text):
    regs = {int(r) for r in re.findall(r"\ba(\d+)\b", text)}
    for lo, hi in re.findall(r"\ba\[(\d+):(\d+)\]", text):
        regs.update(range(int(lo), int(hi) + 1))
    return regs


def _functions(disassembly):
    """-> {name: [(address, instruction text)]} from llvm-objdump -d output"""
    funcs, cur = {}, None
    for line in disassembly.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            cur = funcs.setdefault(m.group(1), [])
            continue
        m = re.match(r"^\s+(\S.*?)\s*//\s*([0-9A-Fa-f]+):", line)
        if m and cur is not None:
            cur.append((int(m.group(2), 16), m.group(1)))
    return funcs


def lint(disassembly):
    """-> (number of asm loads seen, list of violations).

    Forward data flow over each function's control-flow graph: the set of accumulation registers that MAY hold a load
    not yet waited for (union over predecessors); `s_waitcnt vmcnt` empties it, a `global_load_dword aN` adds N, any
    other instruction naming a register of the set is a violation."""
    loads, bad = 0, []
    for func, ins in _functions(disassembly).items():
        index = {addr: i for i, (addr, _) in enumerate(ins)}
        succ = []
        for i, (addr, text) in enumerate(ins):
            op = text.split()[0]
            nxt = [i + 1] if i + 1 < len(ins) else []
            if op == "s_branch" or op.startswith("s_cbranch"):
                imm = int(text.split()[1])
                target = addr + 4 + 4 * (imm - 65536 if imm >= 32768 else imm)
                assert target in index, (func, text)
                succ.append([index[target]] + ([] if op == "s_branch" else nxt))
            elif op in ("s_endpgm", "s_setpc_b64"):
                succ.append([])
            else:
                succ.append(nxt)
        loads += sum(1 for _, t in ins if re.match(r"global_load_dword\s+a\d+\b", t))
        state = [None] * len(ins)   # pending set at the entry of instruction i
        work = [0] if ins else []
        if ins:
            state[0] = frozenset()
        flagged = set()
        while work:
            i = work.pop()
            pending = state[i]
            text = ins[i][1]
            m = re.match(r"global_load_dword\s+a(\d+)\b", text)
            if text.startswith("s_waitcnt") and "vmcnt" in text:
                out = frozenset()
            elif m:
                out = pending | {int(m.group(1))}
            else:
                out = pending
                hit = _agprs(text) & pending
                if hit and i not in flagged:
                    flagged.add(i)
                    bad.append((func, text, sorted(hit)))
            for j in succ[i]:
                merged = out if state[j] is None else state[j] | out
                if merged != state[j]:
                    state[j] = merged
                    work.append(j)
    return loads, bad


def test_lint_flags_a_copy_before_the_wait():
    good = """
0000000000001000 <k>:
	s_cbranch_scc1 3                                            // 000000001000: BF850003
	global_load_dword a5, v3, s[8:9]                            // 000000001004: DC508000 05080003
	s_branch 2                                                  // 00000000100C: BF820001
	v_accvgpr_write_b32 a5, 0                                   // 000000001010: D3D94005 18000080
	v_mfma_f32_32x32x16_f16 v[0:15], v[16:19], a[8:11], v[0:15] // 000000001018: D3D60000 04022110
	s_waitcnt vmcnt(40)                                         // 000000001020: BF8C0F78
	v_accvgpr_read_b32 v1, a5                                   // 000000001024: D3D84001 18000105
	s_endpgm                                                    // 00000000102C: BF810000
"""
    assert lint(good) == (1, [])
    for stale in ("v_accvgpr_mov_b32 a7, a5", "v_accvgpr_read_b32 v1, a5", "v_mfma_f32_32x32x16_f16 v[0:15], v[16:19], a[4:7], v[0:15]"):
        n, bad = lint(good.replace("\ts_waitcnt vmcnt(40)  ", "\t" + stale + " // 00000000101C: 0\n\ts_waitcnt vmcnt(40)  "))
        assert n == 1 and len(bad) == 1 and bad[0][0] == "k"


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason="llvm-objdump of the ROCm toolchain not present")
def test_prefetched_registers_are_not_touched_before_their_wait(tmp_path):
    path = _lib.LIB_PATH
    blob = open(path, "rb").read()
    total, kernels = 0, set()
    for i, obj in enumerate(_code_objects(blob)):
        f = tmp_path / f"co{i}.o"
        f.write_bytes(obj)
        out = subprocess.run([OBJDUMP, "-d", "--mcpu=gfx950", str(f)], check=True, capture_output=True, text=True).stdout
        n, bad = lint(out)
        assert not bad, f"register read or written between an asm load into it and its wait: {bad[:3]}"
        total += n
        if n:
            kernels.update(re.findall(r"^[0-9a-f]+ <(.+)>:$", out, re.M))
    # the two-term forward kernels (7 loads a tile, two instances) and the backward chain (20 a tile + 6 up front)
    assert total >= 2 * 14 + 26, total
    assert any("mlp_fp16x2_kernel" in k for k in kernels) and any("mlp_dgrad_h2_kernel" in k for k in kernels)
