"""Runs one kernel of the training step back to back for a few seconds (clock sampling beside it: train_fwd_clock.sh).
    python tools/debug/train_fwd_loop.py <kind: bf16x3|fp16x2|fp32> <seconds> [infer]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nerf_amd import ops  # noqa: E402
from nerf_amd.weights import make_state_dict_arrays  # noqa: E402

kind, secs = sys.argv[1], float(sys.argv[2])
infer = len(sys.argv) > 3 and sys.argv[3] == "infer"
dev = torch.device("cuda:0")
R, S = 4096, 192
g = torch.Generator().manual_seed(0)
o = (torch.tensor([0.0, 0.0, 4.0]) + 0.1 * torch.randn(R, 3, generator=g)).to(dev)
d = torch.randn(R, 3, generator=g)
d[:, 2] = -d[:, 2].abs() - 1
d = d.to(dev)
z = torch.sort(2 + 4 * torch.rand(R, S, generator=g), -1).values.to(dev)
sd = make_state_dict_arrays(0, 30.0, 2.0)
pack = {"fp32": ops.pack_state_dict, "bf16x3": ops.pack_state_dict_bf16x3, "fp16x2": ops.pack_state_dict_fp16x2}[kind]
fwd = {"fp32": ops.mlp_forward_rays, "bf16x3": ops.mlp_forward_rays_bf16x3, "fp16x2": ops.mlp_forward_rays_fp16x2}[kind]
packed = pack(sd, "model_fine", dev)
run = (lambda: fwd(packed, o, d, z)) if infer else (lambda: ops.mlp_forward_rays_train(packed, o, d, z, kind))
run()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 0
while time.perf_counter() - t0 < secs:
    for _ in range(20):
        out = run()
        del out
    torch.cuda.synchronize()
    n += 20
dt = time.perf_counter() - t0
print(f"{kind} {'inference' if infer else 'training'} forward, {R} x {S}: {dt / n * 1e3:.3f} ms per launch over {n} launches", flush=True)
