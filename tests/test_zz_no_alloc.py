"""nrf_render_static: the allocation-free entry point (include/nerf_amd.h) against nrf_render.

Same kernels, same inputs, so the bar is bit-identity on every output.  Two calls in a row share the library's static
scratch; the second is smaller, so stale contents of the first must not leak into it."""
import ctypes as C

import numpy as np
import pytest
import torch

from nerf_amd import ops
from nerf_amd._lib import RenderArgs, lib
from nerf_amd.camera import focal_from_angle, pose_spherical
from nerf_amd.renderer import near_far_table
from nerf_amd.weights import make_state_dict

pytestmark = pytest.mark.gpu

NC, NI = 64, 128
OUT = (("coarse_rgb", 3), ("coarse_depth", 1), ("fine_rgb", 3), ("fine_depth", 1), ("fine_acc", 1))


def _setup(dev):
    sd = make_state_dict(0, 30.0, 2.0)
    pc = ops.pack_state_dict(sd, "model", dev)
    pf = ops.pack_state_dict(sd, "model_fine", dev)
    bins = torch.from_numpy(near_far_table(2.0, 6.0, NC)).to(dev)
    return pc, pf, bins


def _call(fn, pc, pf, bins, rays_o, rays_d, u, u2, u_c, u_f, workspace):
    dev = rays_o.device
    R = rays_o.shape[0]
    outs = {k: torch.full((R, w), float("nan"), dtype=torch.float32, device=dev) for k, w in OUT}
    a = RenderArgs()
    a.packed_coarse, a.packed_fine = pc.data_ptr(), pf.data_ptr()
    a.rays_o, a.rays_d, a.bins = rays_o.data_ptr(), rays_d.data_ptr(), bins.data_ptr()
    a.u, a.u2 = u.data_ptr(), u2.data_ptr()
    a.u_coarse = u_c.data_ptr() if u_c is not None else None
    a.u_fine = u_f.data_ptr() if u_f is not None else None
    a.n_rays, a.n_coarse, a.n_importance, a.white_bkgd = R, NC, NI, 1
    for k, _ in OUT:
        setattr(a, k, outs[k].data_ptr())
    if workspace is not None:
        a.workspace, a.workspace_bytes = workspace.data_ptr(), workspace.numel()
    else:
        a.workspace, a.workspace_bytes = None, 0
    with torch.cuda.device(dev):
        st = fn(C.byref(a), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    torch.cuda.synchronize(dev)
    return st, outs


def _inputs(dev, first, R, seed, perturbed):
    rays_o, rays_d = ops.raygen(pose_spherical(20.0), 800, 800, focal_from_angle(800), dev, first_ray=first, n_rays=R)
    g = torch.Generator().manual_seed(seed)
    u, u2 = torch.rand(R, NI, generator=g).to(dev), torch.rand(R, NI, generator=g).to(dev)
    u_c = torch.rand(R, NC, generator=g).to(dev) if perturbed else None
    u_f = torch.rand(R, NI, generator=g).to(dev) if perturbed else None
    return rays_o, rays_d, u, u2, u_c, u_f


def test_render_static_matches_render_twice_second_smaller():
    dev = torch.device("cuda:0")
    L = lib()
    pc, pf, bins = _setup(dev)
    for first, R, seed, perturbed in ((400 * 800, 4096, 1, False), (380 * 800 + 100, 1000, 2, True)):
        inp = _inputs(dev, first, R, seed, perturbed)
        ws = torch.empty(int(L.nrf_render_workspace_bytes(R, NC, NI)), dtype=torch.uint8, device=dev)
        st_ref, ref = _call(L.nrf_render, pc, pf, bins, *inp, ws)
        st, got = _call(L.nrf_render_static, pc, pf, bins, *inp, None)
        assert st_ref == 0 and st == 0, (st_ref, st, L.nrf_strerror(st))
        for k, _ in OUT:
            a, b = got[k].cpu().numpy(), ref[k].cpu().numpy()
            assert np.isfinite(a).all(), k
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (k, R, float(np.abs(a - b).max()))
        assert float(ref["fine_acc"].max()) > 0.5             # the rays do cross the scene
    # the device's free memory does not move across a call (the outputs below reuse torch's cached blocks)
    del got, ref
    free0 = torch.cuda.mem_get_info(dev)[0]
    st, got = _call(L.nrf_render_static, pc, pf, bins, *inp, None)
    assert st == 0 and torch.cuda.mem_get_info(dev)[0] == free0


def test_render_static_rejects_sizes_beyond_its_scratch():
    dev = torch.device("cuda:0")
    L = lib()
    pc, pf, bins = _setup(dev)
    R = 4097
    inp = _inputs(dev, 0, R, 3, False)
    st, out = _call(L.nrf_render_static, pc, pf, bins, *inp, None)
    assert st == -6 and b"NRF_RENDER_STATIC_MAX_RAYS" in L.nrf_strerror(st)
    assert torch.isnan(out["fine_rgb"]).all()                 # nothing was launched
