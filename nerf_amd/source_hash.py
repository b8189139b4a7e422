"""Hash of the HIP sources a kernel is built from, for tying a recorded profile to the code it measured.

bench.py's ``roofline.traffic`` is a figure recorded by rocprofv3 PMC passes (counters cannot be read from inside the
benchmark process).  A recorded figure describes the kernel as it was when the passes ran: ``tools/bench_traffic_digest.py``
stores ``kernel_source_sha256`` of the sources at that time in the profile, and bench.py reports the figure only while
the same sources are in the tree -- otherwise ``traffic`` is null and the line says the profile is stale."""
from __future__ import annotations

import hashlib
import os

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
_COMMON = ("nrf_common.h", "mlp_chain.h")
# kernel family -> the sources its code is generated from (beyond the common headers)
SOURCES = {
    "fp32": ("mlp_fp32.hip",),
    "bf16": ("mlp_bf16.hip", "m16.h"),
    "bf16x3": ("mlp_bf16x3.hip", "split3.h", "split_stream.h", "m16.h"),
    "fp16x2": ("mlp_fp16x2.hip", "split2h.h", "split_stream.h"),
    "composite": ("composite.hip",),
    "kilo": ("kilo.hip", "kilo_common.h", "split2h.h", "split3.h"),
}


def kernel_source_sha256(family: str) -> str:
    h = hashlib.sha256()
    for name in _COMMON + SOURCES[family]:
        with open(os.path.join(_CSRC, name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read() + b"\0")
    return h.hexdigest()
