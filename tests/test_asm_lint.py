"""Lint of the built device code: loads the compiler does not see must be waited for before their register is touched.

mlp_fp16x2_kernel and mlp_dgrad_h2_kernel (nerf_amd/csrc/mlp_fp16x2.hip) load a tile's inputs one stage ahead with
`global_load_dword` written in inline asm, into accumulation registers, and wait for them with a counted
`s_waitcnt vmcnt` that names the registers.  The compiler knows nothing of the latency: if register allocation ever
puts a copy of such a register between the load and its wait, the copy reads the old contents and the results are
wrong only sometimes.  This test reads the gfx950 code objects out of the built library and checks, in program order,
that no instruction names a register while a load into it may still be in flight (an in-order model of vmcnt over
each kernel's control-flow graph).  wgrad_h2p_kernel and wgrad_x3p_kernel (mlp_bwd.hip: one loader, wgrad_p_body) load their
operand sets the same way.
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


def _regs(text):
    """-> {('v' | 'a', n)} named by an operand string"""
    out = set()
    for kind, n in re.findall(r"\b([va])(\d+)\b", text):
        out.add((kind, int(n)))
    for kind, lo, hi in re.findall(r"\b([va])\[(\d+):(\d+)\]", text):
        out.update((kind, i) for i in range(int(lo), int(hi) + 1))
    return out


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


VMEM = re.compile(r"^(global|buffer|scratch|flat)_(load|store|atomic)")
QUEUE = 64   # vmcnt is a 6-bit counter: anything older than the newest 63 is retired by every wait


def _merge(p, q):
    """two in-flight queues (tuples of register sets, newest last) that may reach one point: aligned at the newest end"""
    if p is None:
        return q
    if len(p) < len(q):
        p, q = q, p
    pad = len(p) - len(q)
    return tuple(p[i] | (q[i - pad] if i >= pad else frozenset()) for i in range(len(p)))


def lint(disassembly):
    """-> (number of register-destination vector loads seen, list of violations).

    Forward data flow over each function's control-flow graph.  The state is the queue of vector-memory operations
    that MAY still be in flight, in issue order, each with the registers its data lands in (stores, LDS-DMA: none);
    `s_waitcnt vmcnt(N)` retires all but the newest N (the counter is in-order).  An instruction that names a register
    of the queue -- as source or as destination -- before that is a violation.  The compiler's own loads pass by
    construction; the rule exists for the loads and waits written in inline asm, which it cannot see."""
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
        state = [None] * len(ins)   # queue at the entry of instruction i
        work = [0] if ins else []
        if ins:
            state[0] = ()
        flagged = set()
        counted = set()
        while work:
            i = work.pop()
            queue = state[i]
            text = ins[i][1]
            op = text.split()[0]
            operands = text[len(op):]
            m = re.match(r"s_waitcnt\b.*vmcnt\((\d+)\)", text)
            if m:
                n = int(m.group(1))
                out = queue[len(queue) - n:] if n else ()
            else:
                pending = frozenset().union(*queue) if queue else frozenset()
                if VMEM.match(op):
                    parts = operands.split(",")
                    is_load = "_load" in op and "_lds_" not in op
                    dest = frozenset(_regs(parts[0])) if is_load else frozenset()
                    used = _regs(",".join(parts[1:]) if is_load else operands)
                    if is_load and i not in counted:
                        counted.add(i)
                        loads += 1
                    out = queue + (dest,)
                    if len(out) > QUEUE:
                        out = (out[0] | out[1],) + out[2:]
                else:
                    used = _regs(operands)
                    out = queue
                hit = used & pending
                if hit and i not in flagged:
                    flagged.add(i)
                    bad.append((func, text, sorted(hit)[:4]))
            for j in succ[i]:
                merged = _merge(state[j], out)
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
    n, bad = lint(good)   # vmcnt(40) with one operation in flight retires nothing: the read after it is stale
    assert n == 1 and len(bad) == 1 and bad[0][1].startswith("v_accvgpr_read")
    assert lint(good.replace("vmcnt(40)", "vmcnt(0) ")) == (1, [])
    for stale in ("v_accvgpr_mov_b32 a7, a5", "v_accvgpr_read_b32 v1, a5", "v_mfma_f32_32x32x16_f16 v[0:15], v[16:19], a[4:7], v[0:15]"):
        n, bad = lint(good.replace("vmcnt(40)", "vmcnt(0) ").replace("\ts_waitcnt vmcnt(0)   ", "\t" + stale + " // 00000000101C: 0\n\ts_waitcnt vmcnt(0)   "))
        assert n == 1 and len(bad) == 1 and bad[0][0] == "k"


def test_lint_counts_in_issue_order():
    # two loads, then a store: vmcnt(1) retires the two loads' data but not what came after them
    text = """
0000000000002000 <q>:
	global_load_dwordx4 v[4:7], v[0:1], off                     // 000000002000: 0
	global_load_dwordx4 v[8:11], v[0:1], off                    // 000000002008: 0
	global_store_dword v[0:1], v2, off                          // 000000002010: 0
	s_waitcnt vmcnt(1)                                          // 000000002018: 0
	v_add_f32_e32 v3, v4, v9                                    // 00000000201C: 0
	s_endpgm                                                    // 000000002020: 0
"""
    assert lint(text) == (2, [])
    n, bad = lint(text.replace("vmcnt(1)", "vmcnt(2)"))   # the second load may still be in flight
    assert n == 2 and len(bad) == 1 and bad[0][2] == [("v", 9)]
    n, bad = lint(text.replace("global_store_dword v[0:1], v2, off ", "global_store_dword v[0:1], v5, off "))   # stores a register in flight
    assert len(bad) == 1


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
        assert not bad, f"register named while a load into it may be in flight: {bad[:4]}"
        total += n
        kernels.update(re.findall(r"^[0-9a-f]+ <(.+)>:$", out, re.M))
    assert total > 500, total
    for k in ("mlp_fp16x2_kernel", "mlp_dgrad_h2_kernel", "wgrad_h2p_kernel", "wgrad_x3p_kernel"):   # the kernels with asm loads
        assert any(k in name for name in kernels), k
