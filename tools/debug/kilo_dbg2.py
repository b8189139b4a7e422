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
def run(sd):
    packed = kilo_ops.kilo_pack_state_dict(sd, dev)
    raw = kilo_ops.kilo_forward(packed, 4096, AABB, RES, torch.from_numpy(pts).to(dev), torch.from_numpy(dirs).to(dev)).cpu().numpy()
    ref = K.network_forward(sd, torch.from_numpy(pts), torch.from_numpy(dirs), AABB[:3], AABB[3:], RES).numpy()
    return raw, ref
base = make_kilo_state_dict_arrays(7, 4096, 1.0, 0.0)
raw, ref = run(base)
e = np.abs(raw[:, 3] - ref[:, 3])
print("bad points", (e > 1e-4).sum(), "of", len(e), "first bad", np.nonzero(e > 1e-4)[0][:20], flush=True)
eye = np.broadcast_to(np.eye(32, dtype=np.float32), (4096, 32, 32)).copy()
errs = []
for f in range(32):
    sd = dict(base)
    sd["model.pts_linears_weights.1"] = eye; sd["model.pts_linears_biases.1"] = np.zeros((4096, 32), np.float32)
    ws = np.zeros((4096, 1, 32), np.float32); ws[:, 0, f] = 1; sd["model.sigma_linear_weights"] = ws
    sd["model.sigma_linear_biases"] = np.zeros((4096, 1), np.float32)
    raw, ref = run(sd)
    errs.append(float(np.abs(raw[:, 3] - ref[:, 3]).max()))
print("layer0 per-feature err", np.round(errs, 5), flush=True)
# layer 0 with single input column
errs = []
for c in list(range(0, 63)):
    sd = dict(base)
    w = np.zeros_like(base["model.pts_linears_weights.0"]); w[:, :, c] = 1.0; sd["model.pts_linears_weights.0"] = w
    sd["model.pts_linears_biases.0"] = np.full((4096, 32), 2.0, np.float32)
    sd["model.pts_linears_weights.1"] = eye; sd["model.pts_linears_biases.1"] = np.zeros((4096, 32), np.float32)
    ws = np.zeros((4096, 1, 32), np.float32); ws[:, 0, 5] = 1; sd["model.sigma_linear_weights"] = ws
    sd["model.sigma_linear_biases"] = np.zeros((4096, 1), np.float32)
    raw, ref = run(sd)
    errs.append(float(np.abs(raw[:, 3] - ref[:, 3]).max()))
print("layer0 per-column err", np.round(errs, 5), flush=True)
