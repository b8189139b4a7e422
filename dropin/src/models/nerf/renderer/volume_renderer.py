"""Drop-in for the reference file of the same path: the MI355X HIP renderer behind the reference's plug-in point
(configs/nerf/lego.yaml:9 `renderer_module: src.models.nerf.renderer.volume_renderer`)."""
from nerf_amd.renderer import Renderer  # noqa: F401  (reads src.config.cfg itself when it is importable)
