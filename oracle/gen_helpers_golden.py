"""Golden for the user-selection helpers around the channel path, produced by running the REAL reference in the
build container (deepmimo/generator/dataset.py:621-655 interaction views, :702-795 grid info / subset / index
helpers; deepmimo/generator/generator_utils.py:37-184 get_uniform_idxs, LinearPath, get_idxs_with_limits):

    cd /tmp && PYTHONPATH=/root/reference:/root/repo PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg \
        python3 /root/repo/oracle/gen_helpers_golden.py

writes tests/golden/aux_helpers.npz (data only: inputs and the reference's outputs).  The GPU box never runs this
file (it has no /root/reference).
"""
from __future__ import annotations

import io
import os
import sys
from contextlib import redirect_stdout, redirect_stderr

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.oracle_np import synth_rays  # noqa: E402  (input generator only)

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
NX, NY = 12, 9
LIMITS = [dict(x_min=3.0), dict(x_min=2.0, x_max=14.0, y_max=9.0), dict(z_min=1.0, z_max=2.0, y_min=4.0), dict(x_max=-1.0)]
PATHS = [  # first, last, res, n_steps, filter_repeated
    ([0.0, 0.0], [20.0, 14.0], 1.0, None, True), ([1.0, 13.0, 1.5], [19.0, 2.0, 1.5], 0.5, None, True),
    ([0.0, 5.0], [22.0, 5.0], 1.0, 40, "hard"), ([3.0, 3.0], [3.5, 12.0], 1.0, 25, False), ([0.0, 0.0], [22.0, 16.0], 3.0, None, True),
]


def grid_rays():
    """A 12 x 9 user grid (x fastest, 2 m x 1.75 m spacing), 7 paths, interaction codes with up to 4 digits."""
    rays = synth_rays(NX * NY, 7, seed=909)
    xs, ys = np.arange(NX) * 2.0, np.arange(NY) * 1.75
    rays["rx_pos"] = np.stack([np.tile(xs, NY), np.repeat(ys, NX), np.full(NX * NY, 1.5)], axis=1).astype(np.float32)
    rng = np.random.default_rng(910)
    digits = rng.integers(1, 5, size=(NX * NY, 7, 4))
    n_dig = rng.integers(0, 5, size=(NX * NY, 7))                       # 0 digits = LoS code 0
    code = np.zeros((NX * NY, 7))
    for d in range(4):
        code = np.where(n_dig > d, code * 10 + digits[..., d], code)
    rays["inter"] = np.where(np.isnan(rays["power"]), np.nan, code).astype(np.float32)
    for k in ("power", "phase", "delay", "aoa_az", "aoa_el", "aod_az", "aod_el", "inter"):
        rays[k][[5, 40, 77], :] = np.nan                                   # users without any path
    return rays


def main():
    import deepmimo as dm
    rays = grid_rays()
    save = {f"ray_{k}": v for k, v in rays.items()}
    with redirect_stdout(io.StringIO()), redirect_stderr(io.StringIO()):
        # the reference's subset() needs the shared entries to exist (its hasattr raises KeyError otherwise)
        shared = {dm.consts.SCENE_PARAM_NAME: "scene-object", dm.consts.MATERIALS_PARAM_NAME: "materials-object",
                  dm.consts.LOAD_PARAMS_PARAM_NAME: {"max_paths": 7}, dm.consts.RT_PARAMS_PARAM_NAME: {"frequency": 3.5e9}}
        ds = dm.Dataset({**{k: v.copy() for k, v in rays.items()}, **shared})
        save.update(grid_size=np.asarray(ds.grid_size), grid_spacing=np.asarray(ds.grid_spacing),
                    uniform_1_1=ds.get_uniform_idxs([1, 1]), uniform_2_3=ds.get_uniform_idxs([2, 3]),
                    uniform_5_1=ds.get_uniform_idxs([5, 1]), active=ds.get_active_idxs(),
                    num_interactions=np.asarray(ds.num_interactions), inter_int=np.asarray(ds.inter_int),
                    inter_str=np.asarray(ds.inter_str).astype("U8"), distance=np.asarray(ds.distance))
        # a ragged set of users (not a grid any more): the reference shrinks the grid and warns
        keep = np.delete(np.arange(NX * NY), [3, 50, 51, 100])
        rag = dm.Dataset({k: (v[keep].copy() if v.shape[0] == NX * NY else v.copy()) for k, v in rays.items()})
        save.update(ragged_keep=keep, ragged_uniform_2_2=rag.get_uniform_idxs([2, 2]))
        # subset -> channels of the subset
        idxs = ds.get_uniform_idxs([3, 2])
        _ = ds.los, ds.pathloss                                             # computed public attributes travel too
        sub = ds.subset(idxs)
        p = dm.ChannelGenParameters()
        p.bs_antenna.shape = np.array([4, 2])
        p.ofdm.subcarriers = 64
        p.ofdm.selected_subcarriers = np.arange(0, 64, 8)
        h_sub = np.asarray(sub.compute_channels(p)).copy()
        save.update(subset_idxs=idxs, subset_keys=np.array(sorted(k for k in sub.keys() if not k.startswith("_") and k not in ("channel", "ch_params")), dtype="U32"),
                    subset_n_ue=np.array(sub.n_ue), subset_rx_pos=np.asarray(sub.rx_pos), subset_los=np.asarray(sub.los),
                    subset_pathloss=np.asarray(sub.pathloss), subset_channel=h_sub)
        for i, lim in enumerate(LIMITS):
            save[f"limits_{i}"] = dm.get_idxs_with_limits(rays["rx_pos"], **lim)
        for i, (a, b, res, n_steps, filt) in enumerate(PATHS):
            lp = dm.LinearPath(rays["rx_pos"], np.array(a), np.array(b), res=res, n_steps=n_steps, filter_repeated=filt)
            save[f"path_{i}"] = np.asarray(lp.idxs)
            save[f"path_{i}_n"] = np.array(lp.n)
    np.savez_compressed(os.path.join(OUT, "aux_helpers.npz"), **save)
    print("grid", save["grid_size"], save["grid_spacing"], "| uniform_2_3", len(save["uniform_2_3"]), "| ragged", len(save["ragged_uniform_2_2"]),
          "| active", len(save["active"]), "| subset keys", list(save["subset_keys"]), "| inter_str sample", save["inter_str"][0],
          "| paths", [int(save[f"path_{i}_n"]) for i in range(len(PATHS))], "| limits", [len(save[f"limits_{i}"]) for i in range(len(LIMITS))])


if __name__ == "__main__":
    main()
