"""Index helpers of deepmimo/generator/generator_utils.py that callers of the channel path use to pick users
(``dataset.subset(dataset.get_uniform_idxs([2, 2]))``).  Host-side NumPy; nothing here touches the GPU."""
from __future__ import annotations

from typing import Sequence

import numpy as np


def dbw2watt(val):
    """dBW -> W (generator_utils.py:24-35)."""
    return 10 ** (val / 10)


def get_uniform_idxs(n_ue: int, grid_size: np.ndarray, steps: Sequence[int]) -> np.ndarray:
    """User indices on every steps[0]-th column and steps[1]-th row of an [nx, ny] grid stored row by row
    (index = column + row * nx) - generator_utils.py:37-70.  When nx * ny does not match the number of users the
    reference warns and shrinks BOTH grid dimensions by one until the grid fits ("pseudo-uniform" indices); that
    behaviour, including the in-place shrink of the caller's `grid_size` array, is kept."""
    if list(steps) == [1, 1]:
        return np.arange(n_ue)
    if np.prod(grid_size) != n_ue:
        print(f"Warning. Grid_size: {grid_size} = {np.prod(grid_size)} users != {n_ue} users in rx_pos")
        print("Computing pseudo-uniform indices.")
        while np.prod(grid_size) > n_ue:
            grid_size -= 1
    nx = int(grid_size[0])
    cols = np.arange(0, int(grid_size[0]), steps[0])
    rows = np.arange(0, int(grid_size[1]), steps[1])
    return (rows[:, None] * nx + cols[None, :]).reshape(-1)


_AXIS = {"x": 0, "y": 1, "z": 2}


def get_idxs_with_limits(data_pos: np.ndarray, **limits) -> np.ndarray:
    """Indices of the users inside an axis-aligned box given as x_min / x_max / y_min / y_max / z_min / z_max keyword
    limits (both ends inclusive) - generator_utils.py:148-184."""
    allowed = {f"{a}_{e}" for a in _AXIS for e in ("min", "max")}
    if any(k not in allowed for k in limits):
        raise ValueError(f"Invalid limit key. Supported limits are: {allowed}")
    keep = np.ones(len(data_pos), dtype=bool)
    for name, value in limits.items():
        axis = _AXIS[name[0]]
        if axis >= data_pos.shape[1]:
            raise ValueError(f"Cannot apply {name[0]} limit to {data_pos.shape[1]}D positions")
        col = data_pos[:, axis]
        keep &= (col >= value) if name.endswith("min") else (col <= value)
    return np.flatnonzero(keep)


class LinearPath:
    """Users nearest to equally spaced points on the segment first_pos -> last_pos (generator_utils.py:73-146).

    ``idxs`` are the dataset indices along the path, ``n`` their count.  ``filter_repeated`` True drops consecutive
    repeats, 'hard' keeps each user once (sorted), False keeps every sample.  Without ``n_steps`` the number of
    samples is path length / res, with res raised to the spacing of the first two dataset points when it is finer
    than that (and repeats are being filtered)."""

    def __init__(self, rx_pos: np.ndarray, first_pos, last_pos, res: float = 1, n_steps=None, filter_repeated=True) -> None:
        first_pos, last_pos = np.asarray(first_pos, dtype=float), np.asarray(last_pos, dtype=float)
        if len(first_pos) == 2:                                               # z defaults to 0
            first_pos, last_pos = np.append(first_pos, 0.0), np.append(last_pos, 0.0)
        self.first_pos, self.last_pos = first_pos, last_pos
        if n_steps:
            self.n = n_steps
        else:
            data_res = np.linalg.norm(rx_pos[0] - rx_pos[1])
            if res < data_res and filter_repeated:
                print(f"Changing resolution to {data_res} to eliminate repeated positions")
                res = data_res
            self.n = int(np.linalg.norm(first_pos - last_pos) / res)
        samples = np.linspace(first_pos, last_pos, self.n)                    # [n, 3]
        idxs = np.array([int(np.argmin(np.linalg.norm(rx_pos - p, axis=1))) for p in samples])
        if filter_repeated:
            idxs = idxs[np.concatenate(([True], np.diff(idxs) != 0))]
            if filter_repeated == "hard":
                idxs = np.unique(idxs)
            self.n = len(idxs)
        self.idxs = idxs
