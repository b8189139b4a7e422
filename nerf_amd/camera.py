"""Synthetic Blender-style cameras for tests and ``bench.py`` (no dataset is available).

The reference reads ``camera_angle_x`` and one 4x4 ``transform_matrix`` per frame from
``transforms_<split>.json`` (/root/reference/src/datasets/nerf/blender.py:47-55, :76-79).
Here the same quantities are generated: the public Blender-lego field of view and poses on
a sphere looking at the origin, in the Blender/OpenGL convention the reference's ray
formula assumes (camera looks down -z, +y up: blender.py:94).
"""
from __future__ import annotations

import numpy as np

LEGO_CAMERA_ANGLE_X = 0.6911112070083618   # transforms_*.json of the Blender lego scene
LEGO_RADIUS = 4.031128874


def focal_from_angle(W: int, camera_angle_x: float = LEGO_CAMERA_ANGLE_X) -> float:
    """``0.5 * W / tan(0.5 * camera_angle_x)`` (blender.py:76), float64."""
    return float(0.5 * W / np.tan(0.5 * camera_angle_x))


def pose_spherical(theta_deg: float, phi_deg: float = -30.0, radius: float = LEGO_RADIUS) -> np.ndarray:
    """Camera-to-world 4x4 (float64) at azimuth ``theta``, elevation ``phi`` on a sphere."""
    th = np.deg2rad(theta_deg)
    ph = np.deg2rad(phi_deg)
    trans = np.eye(4)
    trans[2, 3] = radius
    rot_phi = np.array([[1, 0, 0, 0],
                        [0, np.cos(ph), -np.sin(ph), 0],
                        [0, np.sin(ph), np.cos(ph), 0],
                        [0, 0, 0, 1]], dtype=np.float64)
    rot_theta = np.array([[np.cos(th), 0, -np.sin(th), 0],
                          [0, 1, 0, 0],
                          [np.sin(th), 0, np.cos(th), 0],
                          [0, 0, 0, 1]], dtype=np.float64)
    flip = np.array([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]], dtype=np.float64)
    return flip @ rot_theta @ rot_phi @ trans


def test_set_poses(n_views: int = 200):
    """The 200-view synthetic test orbit of BASELINE config 5: theta = -180 .. +178.2 deg."""
    return [pose_spherical(t) for t in np.linspace(-180.0, 180.0, n_views + 1)[:-1]]
