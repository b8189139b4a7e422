"""Deterministic synthetic NeRF weights in the reference's state-dict layout.

The reference's checkpoint stores the bare ``Network.state_dict()``
(/root/reference/train.py:68-82 via src/utils/net_utils.py:325-345), whose keys are
``model.<layer>.{weight,bias}`` for the coarse MLP and ``model_fine.<layer>.*`` for the fine
one, with ``nn.Linear`` ``(out, in)`` weight matrices
(/root/reference/src/models/nerf/network.py:22-47).  There is no dataset or checkpoint in the
build container or on the GPU box, so tests and ``bench.py`` use weights drawn here from a
numpy ``Generator(PCG64(seed))``: a fixture then stores a seed, not 4.8 MB of floats.
"""
from __future__ import annotations

import numpy as np

# (name, out_features, in_features) of one 8x256 NeRF MLP with use_viewdirs=True,
# xyz PE 63 channels, dir PE 27 channels, skip concat after layer 4.
NERF_LAYERS = (
    ("pts_linears.0", 256, 63),
    ("pts_linears.1", 256, 256),
    ("pts_linears.2", 256, 256),
    ("pts_linears.3", 256, 256),
    ("pts_linears.4", 256, 256),
    ("pts_linears.5", 256, 319),
    ("pts_linears.6", 256, 256),
    ("pts_linears.7", 256, 256),
    ("views_linears.0", 128, 283),
    ("feature_linear", 256, 256),
    ("alpha_linear", 1, 256),
    ("rgb_linear", 3, 128),
)

PARAMS_PER_MODEL = sum(o * i + o for _, o, i in NERF_LAYERS)  # 595,844


def make_model_arrays(rng: np.random.Generator, density_gain: float = 1.0,
                      density_bias: float = 0.0) -> dict:
    """One MLP's parameters: uniform(-1/sqrt(fan_in), +1/sqrt(fan_in)) like nn.Linear's default.

    ``density_gain`` / ``density_bias`` rescale ``alpha_linear`` so that synthetic scenes are
    not almost fully transparent (a random-init network emits |sigma| << 1, which would leave
    the transmittance scan numerically trivial).
    """
    out = {}
    for name, fan_out, fan_in in NERF_LAYERS:
        bound = 1.0 / np.sqrt(fan_in)
        w = rng.uniform(-bound, bound, size=(fan_out, fan_in)).astype(np.float32)
        b = rng.uniform(-bound, bound, size=(fan_out,)).astype(np.float32)
        if name == "alpha_linear":
            w = (w * np.float32(density_gain)).astype(np.float32)
            b = (b * np.float32(density_gain) + np.float32(density_bias)).astype(np.float32)
        out[name + ".weight"] = w
        out[name + ".bias"] = b
    return out


def make_state_dict_arrays(seed: int = 0, density_gain: float = 1.0,
                           density_bias: float = 0.0) -> dict:
    """Full ``Network`` state dict (coarse ``model.*`` then fine ``model_fine.*``) as numpy."""
    rng = np.random.Generator(np.random.PCG64(seed))
    sd = {}
    for prefix in ("model", "model_fine"):
        for k, v in make_model_arrays(rng, density_gain, density_bias).items():
            sd[prefix + "." + k] = v
    return sd


def make_state_dict(seed: int = 0, density_gain: float = 1.0, density_bias: float = 0.0):
    """Same as :func:`make_state_dict_arrays` but as torch CPU tensors (for load_state_dict)."""
    import torch

    return {k: torch.from_numpy(v.copy())
            for k, v in make_state_dict_arrays(seed, density_gain, density_bias).items()}


# ---- KiloNeRF: 16^3 tiny MLPs as batched tensors ---------------------------------------------
# (name, shape-after-the-cell-axis) in the reference's state-dict layout
# (/root/reference/src/models/kilonerf/network.py:87-113), fan-in used for the uniform bound
KILO_TENSORS = (
    ("pts_linears_weights.0", (32, 63), 63), ("pts_linears_biases.0", (32,), 63),
    ("pts_linears_weights.1", (32, 32), 32), ("pts_linears_biases.1", (32,), 32),
    ("feature_linear_weights", (32, 32), 32), ("feature_linear_biases", (32,), 32),
    ("sigma_linear_weights", (1, 32), 32), ("sigma_linear_biases", (1,), 32),
    ("view_linears_weights", (32, 59), 59), ("view_linears_biases", (32,), 59),
    ("rgb_linear_weights", (3, 32), 32), ("rgb_linear_biases", (3,), 32),
)
KILO_PARAMS_PER_CELL = sum(int(np.prod(s)) for _, s, _ in KILO_TENSORS)  # 6212


def make_kilo_state_dict_arrays(seed: int = 0, num_cells: int = 4096, density_gain: float = 1.0,
                                density_bias: float = 0.0) -> dict:
    """Deterministic synthetic KiloNeRF weights: uniform(+-1/sqrt(fan_in)) per tensor, every cell
    different; ``model.*`` keys as in the reference's ``Network.state_dict()`` (buffers excluded)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    sd = {}
    for name, shape, fan_in in KILO_TENSORS:
        bound = 1.0 / np.sqrt(fan_in)
        a = rng.uniform(-bound, bound, size=(num_cells,) + shape).astype(np.float32)
        if name == "sigma_linear_weights":
            a = (a * np.float32(density_gain)).astype(np.float32)
        if name == "sigma_linear_biases":
            a = (a * np.float32(density_gain) + np.float32(density_bias)).astype(np.float32)
        sd["model." + name] = a
    return sd
