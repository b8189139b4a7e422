"""The handful of config keys the render path reads.

The reference keeps them in a global yacs ``cfg`` built from ``sys.argv`` at import time
(/root/reference/src/config/config.py:224-234); ``Renderer.__init__`` reads
``cfg.task_arg.{N_samples,N_importance,white_bkgd,perturb,raw_noise_std,near,far,chunk_size}``
(/root/reference/src/models/nerf/renderer/volume_renderer.py:22-35).  Inside the reference
tree the drop-in uses that object; stand-alone (tests, bench) it uses these defaults, which
are the values of /root/reference/configs/nerf/lego.yaml:15-39.
"""
from __future__ import annotations

from types import SimpleNamespace


def default_task_arg(**overrides) -> SimpleNamespace:
    d = dict(N_rays=1024, chunk_size=4096, white_bkgd=1, N_samples=64, N_importance=128,
             perturb=1, raw_noise_std=0, near=2, far=6, use_viewdirs=True)
    d.update(overrides)
    return SimpleNamespace(**d)


def default_cfg(**task_overrides) -> SimpleNamespace:
    return SimpleNamespace(task_arg=default_task_arg(**task_overrides))


def load_yaml(path: str) -> SimpleNamespace:
    """Read ``task_arg`` from a reference-style yaml (no parent_cfg recursion, no CLI merge)."""
    import yaml
    with open(path) as f:
        y = yaml.safe_load(f)
    return default_cfg(**{k: v for k, v in (y.get("task_arg") or {}).items()
                          if not isinstance(v, (dict, list))})
