"""The collectives of the N > 1 path on RCCL itself (backend "nccl"), through a ONE-rank group: the only kind a one-GPU
box can form (RCCL refuses two ranks on one device; the two-rank tests run over gloo).  A one-rank group still goes
through RCCL's initialisation with this image's environment (HSA_ENABLE_IPC_MODE_LEGACY=0) and enqueues every
collective on the device, so what is pinned here is what gloo cannot show: `init_process_group("nccl", device_id=...)` as
bench.py calls it (the reference's train.py:116-120), device tensors handed to the collectives without host staging
(nerf_amd/parallel.py's non-gloo branches), and that their results are the identities a one-rank sum / max / gather
must be.  Runs in a child process so that the test session itself holds no process group."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, sys
import torch
import torch.distributed as dist
sys.path.insert(0, os.environ["NRF_ROOT"])
from nerf_amd import parallel

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group(backend="nccl", init_method="env://", device_id=dev)       # bench.py main()
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1

# bench.py: barrier between synchronisations, max over ranks of the elapsed time (float64 on the device)
torch.cuda.synchronize(dev); dist.barrier(); torch.cuda.synchronize(dev)
t = torch.tensor([1.25], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert float(t.item()) == 1.25

# TrainStep: one sum all-reduce of the flat gradient buffer (coarse + fine 8x256 networks: about 1.2 M floats)
g = torch.Generator(device="cpu").manual_seed(0)
flat = torch.randn(1191688, generator=g).to(dev)
keep = flat.clone()
assert parallel.allreduce_gradients(flat) == 1.0
parallel._all_reduce(flat, dist.ReduceOp.SUM)
assert torch.equal(flat, keep)

# render_view_sharded: ONE all_gather_into_tensor of the packed per-ray outputs, device to device
n = 4099
local = {"fine_rgb_map": torch.randn(n, 3, generator=g).to(dev), "fine_depth_map": torch.randn(n, generator=g).to(dev),
         "fine_acc_map": torch.randn(n, generator=g).to(dev)}
out = parallel.gather_rows(local, n, 0, 1)
assert list(out) == list(local)
for k in local:
    assert out[k].shape == local[k].shape and torch.equal(out[k], local[k]), k

# KiloTrainStep: union of touched cells (int32 MAX), sum of the touched cells' 6212-float records, scatter back
from nerf_amd.weights import KILO_TENSORS
grads = [torch.randn((64,) + tuple(shape), generator=g).to(dev) for _, shape, _ in KILO_TENSORS]
keep = [x.clone() for x in grads]
touched = torch.zeros(64, dtype=torch.int32, device=dev)
touched[[3, 17, 40, 63]] = torch.tensor([5, 1, 9, 2], dtype=torch.int32, device=dev)
parallel.exchange_cell_gradients(grads, touched)
assert touched.tolist() == [1 if c in (3, 17, 40, 63) else 0 for c in range(64)]
for a, b in zip(grads, keep):
    assert torch.equal(a, b)

dist.barrier()
dist.destroy_process_group()
print("rccl ok")
"""


def test_collectives_of_the_sharded_path_on_a_one_rank_rccl_group():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", LOCAL_WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", NRF_ROOT=ROOT)
    p = subprocess.run([sys.executable, "-u", "-c", CHILD], env=env, capture_output=True, text=True, timeout=240, cwd=ROOT)
    assert p.returncode == 0 and "rccl ok" in p.stdout, (p.stdout[-2000:], p.stderr[-4000:])
