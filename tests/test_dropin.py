"""The drop-in boundary: the files under dropin/ are loaded the way the reference's make_renderer does
(imp.load_source(cfg.renderer_module, cfg.renderer_path).Renderer(network), make_renderer.py:4-8) with a stand-in
for the reference's global `src.config.cfg`; construction needs the library but no GPU."""
import importlib.util
import os
import sys
import types

import pytest
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DROPIN = os.path.join(ROOT, "dropin")


def _load_source(module_name, rel_path):
    """imp.load_source(name, path): run the file at `path` (relative to the reference's CWD) as module `name`."""
    spec = importlib.util.spec_from_file_location(module_name, os.path.join(DROPIN, rel_path))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture
def fake_src_config(tmp_path):
    """`from src.config import cfg` as the reference's modules do (config.py:224-234), with lego.yaml's values."""
    ns = types.SimpleNamespace
    cfg = ns(task_arg=ns(N_samples=64, N_importance=128, white_bkgd=1, perturb=1, raw_noise_std=0, near=2, far=6,
                         chunk_size=4096, grid_resolution=[16, 16, 16], aabb={"min": [-1.0] * 3, "max": [1.0] * 3}),
             sampler=ns(occ_factor=16, occ_threshold=10, ert_threshold=0.01, near=2, far=6, render_step_size=0.005),
             result_dir=str(tmp_path / "result"), teacher_model_dir=str(tmp_path / "no_teacher"))
    src, config = types.ModuleType("src"), types.ModuleType("src.config")
    config.cfg = cfg
    src.config = config
    old = {k: sys.modules.get(k) for k in ("src", "src.config")}
    sys.modules["src"], sys.modules["src.config"] = src, config
    yield cfg
    for k, v in old.items():
        if v is None:
            sys.modules.pop(k, None)
        else:
            sys.modules[k] = v


def test_nerf_renderer_module_path(fake_src_config):
    renderer_module = "src.models.nerf.renderer.volume_renderer"           # configs/nerf/lego.yaml:9
    mod = _load_source(renderer_module, renderer_module.replace(".", "/") + ".py")
    from nerf_amd.network import Network
    from nerf_amd.renderer import Renderer
    assert mod.Renderer is Renderer
    net = Network()
    r = mod.Renderer(net)                                                   # one positional argument, as make_renderer calls it
    assert isinstance(r, nn.Module) and r.net is net
    assert dict(r.named_parameters()).keys() == {"net." + k for k in dict(net.named_parameters())}   # optimiser reaches them
    assert (r.N_samples, r.N_importance, r.near, r.far, r.perturb) == (64, 128, 2, 6, 1)             # read from src.config.cfg
    assert callable(r.render) and callable(r.forward)


class _RefStyleKiloNet(nn.Module):
    """A module with the reference's KiloNeRF layout but none of this package's classes (network.py:172-220)."""

    def __init__(self, cells=8):
        super().__init__()
        from nerf_amd.weights import KILO_TENSORS
        m = nn.Module()
        m.pts_linears_weights = nn.ParameterList([nn.Parameter(torch.zeros((cells,) + KILO_TENSORS[0][1])),
                                                  nn.Parameter(torch.zeros((cells,) + KILO_TENSORS[2][1]))])
        m.pts_linears_biases = nn.ParameterList([nn.Parameter(torch.zeros((cells,) + KILO_TENSORS[1][1])),
                                                 nn.Parameter(torch.zeros((cells,) + KILO_TENSORS[3][1]))])
        for name, shape, _ in KILO_TENSORS[4:]:
            setattr(m, name, nn.Parameter(torch.zeros((cells,) + shape)))
        self.model = m
        self.register_buffer("aabb_min", torch.tensor([-1.0, -1.0, -1.0]))
        self.register_buffer("aabb_max", torch.tensor([1.0, 1.0, 1.0]))
        self.register_buffer("grid_resolution", torch.tensor([2, 2, 2]))


def test_kilonerf_renderer_module_path(fake_src_config):
    renderer_module = "src.models.kilonerf.renderer.volumn_renderer"       # configs/kilonerf/lego.yaml:11
    mod = _load_source(renderer_module, renderer_module.replace(".", "/") + ".py")
    from nerf_amd.kilonerf import KiloNetView, KiloRenderer
    from nerf_amd.weights import KILO_TENSORS
    assert mod.Renderer is KiloRenderer
    net = _RefStyleKiloNet()
    r = mod.Renderer(net)
    assert r.net is net and r.occ_grid is None                              # no cached grid, no teacher checkpoint
    assert r.occ_grid_resolution == [256, 256, 256] and (r.step, r.ert) == (0.005, 0.01)
    assert r.grid_path.endswith("occ_grid_res256_256_256_thresh10.pth")     # volumn_renderer.py:37-38
    view = r.view
    assert isinstance(view, KiloNetView) and view.num_models == 8
    assert [tuple(t.shape[1:]) for t in view.tensors()] == [s for _, s, _ in KILO_TENSORS]
    assert view.grid() == ([-1.0, -1.0, -1.0, 1.0, 1.0, 1.0], [2, 2, 2])
    with pytest.raises(RuntimeError, match="occupancy grid"):
        r({"xyz": torch.zeros(1, 4, 3), "viewdirs": torch.ones(1, 4, 3)}, is_training=False)


def test_kilonerf_renderer_loads_cached_grid(fake_src_config):
    os.makedirs(fake_src_config.result_dir)
    grid = torch.zeros((256, 256, 256), dtype=torch.bool)
    grid[100:140, 100:140, 100:140] = True
    torch.save(grid, os.path.join(fake_src_config.result_dir, "occ_grid_res256_256_256_thresh10.pth"))   # the reference's format
    mod = _load_source("src.models.kilonerf.renderer.volumn_renderer", "src/models/kilonerf/renderer/volumn_renderer.py")
    r = mod.Renderer(_RefStyleKiloNet())
    assert r.occ_grid is not None and r.occ_grid.dtype == torch.uint8 and int(r.occ_grid.sum()) == 40 ** 3
