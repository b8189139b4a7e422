"""Times the three MLP kernels (fp32 MFMA, bf16x3, bf16) on 65,536 rays x 192 samples in one process, rounds interleaved."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nerf_amd import ops  # noqa: E402
from nerf_amd.weights import make_state_dict_arrays  # noqa: E402

dev = torch.device("cuda:0")
R, S = int(os.environ.get("RAYS", 65536)), 192
g = torch.Generator().manual_seed(0)
o = (torch.tensor([0.0, 0.0, 4.0]) + 0.1 * torch.randn(R, 3, generator=g)).to(dev)
d = torch.randn(R, 3, generator=g)
d[:, 2] = -d[:, 2].abs() - 1
d = d.to(dev)
z = torch.sort(2 + 4 * torch.rand(R, S, generator=g), -1).values.to(dev)
sd = make_state_dict_arrays(0, 30.0, 2.0)
kinds = {"fp32": (ops.pack_state_dict, ops.mlp_forward_rays), "bf16x3": (ops.pack_state_dict_bf16x3, ops.mlp_forward_rays_bf16x3),
         "bf16": (ops.pack_state_dict_bf16, ops.mlp_forward_rays_bf16)}
only = sys.argv[1:] or list(kinds)
packed = {k: kinds[k][0](sd, "model_fine", dev) for k in only}
outs = {}
for k in only:
    outs[k] = kinds[k][1](packed[k], o, d, z)
torch.cuda.synchronize()
if "fp32" in outs:
    for k in only:
        print(k, "max |raw - fp32 kernel| rgb %.2e sigma %.2e" % (float((outs[k][..., :3] - outs["fp32"][..., :3]).abs().max()),
                                                                 float((outs[k][..., 3] - outs["fp32"][..., 3]).abs().max())))
best = {k: 1e9 for k in only}
for _ in range(3):
    for k in only:
        t = time.perf_counter()
        kinds[k][1](packed[k], o, d, z)
        torch.cuda.synchronize()
        best[k] = min(best[k], time.perf_counter() - t)
for k in only:
    print(f"{k:8s} {best[k] * 1e3:9.3f} ms  {R * S * 1186816 / best[k] / 1e12:8.1f} TFLOP/s", flush=True)
