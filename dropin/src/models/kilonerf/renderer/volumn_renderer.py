"""Drop-in for the reference file of the same path (configs/kilonerf/lego.yaml:11
`renderer_module: src.models.kilonerf.renderer.volumn_renderer`): occupancy-grid marching, per-cell tiny MLPs and
packed compositing on the MI355X, forward and backward."""
from nerf_amd.kilonerf import KiloRenderer as Renderer  # noqa: F401
