"""Stand-alone ``Network`` with the reference's parameter layout, evaluated by the HIP MLP.

Same module tree and state-dict keys as /root/reference/src/models/nerf/network.py:126-159
(``model`` / ``model_fine``, each ``pts_linears.{0..7}``, ``views_linears.0``,
``feature_linear``, ``alpha_linear``, ``rgb_linear``), so reference checkpoints load with
``load_state_dict`` and the drop-in ``Renderer`` can read either this class or the
reference's own ``Network``.  ``forward`` keeps the reference signature
(network.py:171-195) but runs nrf_mlp_forward_points: there is no torch fallback.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops


class NeRF(nn.Module):
    """Parameter container of one 8x256 MLP (network.py:9-47 with D=8, W=256, skips=[4])."""

    def __init__(self, D=8, W=256, input_ch=63, input_ch_views=27, skips=(4,)):
        super().__init__()
        if (D, W, input_ch, input_ch_views, tuple(skips)) != (8, 256, 63, 27, (4,)):
            raise ValueError("the HIP kernels are built for D=8, W=256, skips=[4], 10/4-frequency encodings")
        self.pts_linears = nn.ModuleList(
            [nn.Linear(input_ch, W)] +
            [nn.Linear(W, W) if i not in skips else nn.Linear(W + input_ch, W) for i in range(D - 1)])
        self.views_linears = nn.ModuleList([nn.Linear(input_ch_views + W, W // 2)])
        self.feature_linear = nn.Linear(W, W)
        self.alpha_linear = nn.Linear(W, 1)
        self.rgb_linear = nn.Linear(W // 2, 3)

    def layer_params(self):
        ws, bs = [], []
        for name in ops.LAYER_NAMES:
            mod = self
            for part in name.split("."):
                mod = mod[int(part)] if part.isdigit() else getattr(mod, part)
            ws.append(mod.weight)
            bs.append(mod.bias)
        return ws, bs


class Network(nn.Module):
    def __init__(self):
        super().__init__()
        self.model = NeRF()
        self.model_fine = NeRF()

    @torch.no_grad()
    def forward(self, inputs, viewdirs, model=""):
        """(R,S,3)+(R,3) or (P,3)+(P,3) -> (...,4) raw [r,g,b,sigma]; dirs used as given."""
        m = self.model_fine if model == "fine" else self.model
        packed = ops.pack_weights(*m.layer_params())
        flat = inputs.reshape(-1, 3)
        dirs = viewdirs[:, None].expand(inputs.shape) if inputs.ndim == 3 else viewdirs
        out = ops.mlp_forward_points(packed, flat, dirs.reshape(-1, 3))
        return out.reshape(list(inputs.shape[:-1]) + [4])
