"""Times nrf_mlp_forward_rays_bf16 of the in-tree library on 65,536 rays x 192 samples (a tenth of a frame's fine pass),
rounds interleaved in one process; NRF_BF16_SETS picks the tiling at library load, so run it once per setting."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nerf_amd import ops  # noqa: E402
from nerf_amd.weights import make_state_dict_arrays  # noqa: E402

dev = torch.device("cuda:0")
R, S = 65536, 192
g = torch.Generator().manual_seed(0)
o = (torch.tensor([0.0, 0.0, 4.0]) + 0.1 * torch.randn(R, 3, generator=g)).to(dev)
d = torch.randn(R, 3, generator=g)
d[:, 2] = -d[:, 2].abs() - 1
d = d.to(dev)
z = torch.sort(2 + 4 * torch.rand(R, S, generator=g), -1).values.to(dev)
pb = ops.pack_state_dict_bf16(make_state_dict_arrays(0, 30.0, 2.0), "model_fine", dev)
ops.mlp_forward_rays_bf16(pb, o, d, z)
torch.cuda.synchronize()
best = 1e9
for _ in range(5):
    t = time.perf_counter()
    for _ in range(3):
        ops.mlp_forward_rays_bf16(pb, o, d, z)
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t) / 3)
print(f"NRF_BF16_SETS={os.environ.get('NRF_BF16_SETS', '1')}: {best * 1e3:.3f} ms  {R * S * 1186816 / best / 1e12:.1f} TFLOP/s "
      f"= {R * S * 1186816 / best / 2.5e15:.3f} of 2.5 PF", flush=True)
