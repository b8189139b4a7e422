import sys, numpy as np, torch
sys.path.insert(0, ".")
from nerf_amd import kilo_ops
from nerf_amd.weights import make_kilo_state_dict_arrays
from oracle import kilo_oracle as K
AABB = [-1., -1, -1, 1, 1, 1]; RES = [16, 16, 16]
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
pts = rng.uniform(-1, 1, (512, 3)).astype(np.float32)
dirs = rng.normal(size=(512, 3)).astype(np.float32)
def run(sd, tag):
    packed = kilo_ops.kilo_pack_state_dict(sd, dev)
    raw = kilo_ops.kilo_forward(packed, 4096, AABB, RES, torch.from_numpy(pts).to(dev), torch.from_numpy(dirs).to(dev)).cpu().numpy()
    ref = K.network_forward(sd, torch.from_numpy(pts), torch.from_numpy(dirs), AABB[:3], AABB[3:], RES).numpy()
    print(tag, "rgb err", np.abs(raw[:, :3] - ref[:, :3]).max(), "sigma err", np.abs(raw[:, 3] - ref[:, 3]).max(), flush=True)
base = make_kilo_state_dict_arrays(7, 4096, 1.0, 0.0)
run(base, "full")
sd = dict(base); w = sd["model.view_linears_weights"].copy(); w[:, :, 32:] = 0; sd["model.view_linears_weights"] = w
run(sd, "no dir cols")
sd = dict(base); w = sd["model.view_linears_weights"].copy(); w[:, :, :32] = 0; sd["model.view_linears_weights"] = w
run(sd, "no feature cols")
for c0, c1, nm in [(32, 35, "raw dir only"), (35, 59, "dir trig only"), (35, 41, "dir f0"), (41, 47, "dir f1"), (47, 53, "dir f2"), (53, 59, "dir f3")]:
    sd = dict(base); w = np.zeros_like(sd["model.view_linears_weights"]); w[:, :, c0:c1] = sd["model.view_linears_weights"][:, :, c0:c1]; sd["model.view_linears_weights"] = w
    run(sd, nm)
for c0, c1, nm in [(0, 3, "raw pts only"), (3, 63, "pts trig only"), (3, 9, "f0"), (57, 63, "f9")]:
    sd = dict(base); w = np.zeros_like(sd["model.pts_linears_weights.0"]); w[:, :, c0:c1] = sd["model.pts_linears_weights.0"][:, :, c0:c1]; sd["model.pts_linears_weights.0"] = w
    run(sd, nm)
