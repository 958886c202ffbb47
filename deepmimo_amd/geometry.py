"""Host-side geometry utilities of the public API (deepmimo/generator/geometry.py).

The per-path geometry of the hot path (rotation, FoV, array response) runs on the GPU
(csrc/k1_path_prep.hip, csrc/k2_channel_fd*.hip).  What stays on the host is the one-vector
utility ``steering_vec`` (geometry.py:322-339, exported at deepmimo/__init__.py:36-38) that users
call to build beam codebooks, and the element-index helper it needs."""
from __future__ import annotations

import numpy as np


def _ant_indices(panel_size) -> np.ndarray:
    """[M, 3] element indices of an [Mh, Mv] panel: x = 0, y fastest, z slowest (geometry.py:105-120)."""
    mh, mv = int(panel_size[0]), int(panel_size[1])
    m = np.arange(mh * mv)
    return np.stack([np.zeros_like(m), m % mh, m // mh], axis=1)


def steering_vec(array, phi: float = 0, theta: float = 0, spacing: float = 0.5) -> np.ndarray:
    """Normalised array response [M, 1] for a beam towards (phi, theta) degrees.

    Same argument convention as the reference, including its swap: the response is evaluated at
    zenith = phi and azimuth = theta + 90 deg (geometry.py:338)."""
    idx = _ant_indices(array)
    kd = 2 * np.pi * spacing
    zen, az = phi * np.pi / 180, theta * np.pi / 180 + np.pi / 2
    gamma = 1j * kd * np.array([np.sin(zen) * np.cos(az), np.sin(zen) * np.sin(az), np.cos(zen)])
    resp = np.exp(idx @ gamma).reshape(-1, 1)
    return resp / np.linalg.norm(resp)
