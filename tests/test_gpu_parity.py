"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the public
Dataset API and therefore through the C-ABI, against (a) the golden vectors the real reference
produced and (b) the NumPy oracle on seeded synthetic rays.

Tolerance (BASELINE.json north_star / SURVEY.md 8c): |dH| <= 5e-5 * max|H_ref[user]| + 1e-12;
LoS, path counts and FoV masks bit-exact; power_linear rtol 1e-6."""
import os

import numpy as np
import torch
import pytest

from tests._cases import (golden_names, load_golden, oracle_params, fov_args, assert_channel_close, TOL_REL)

pytestmark = pytest.mark.gpu


def _dm_params(case, ue_rot):
    import deepmimo_amd as dm
    p = dm.ChannelGenParameters()
    p.bs_antenna.shape = np.array(case["bs_shape"])
    p.ue_antenna.shape = np.array(case["ue_shape"])
    p.bs_antenna.spacing = case["bs_spacing"]
    p.ue_antenna.spacing = case["ue_spacing"]
    p.bs_antenna.rotation = np.array(case["bs_rot"])
    p.ue_antenna.rotation = np.array(ue_rot)
    p.bs_antenna.radiation_pattern = case["bs_pattern"]
    p.ue_antenna.radiation_pattern = case["ue_pattern"]
    p.num_paths = case["num_paths"]
    p.freq_domain = case["freq_domain"]
    p.ofdm.subcarriers = case["subcarriers"]
    p.ofdm.selected_subcarriers = np.array(case["selected"])
    p.ofdm.bandwidth = case["bandwidth"]
    p.ofdm.rx_filter = case["rx_filter"]
    return p


def _dataset(case, rays, with_doppler=False):
    import deepmimo_amd as dm
    d = {k: v.copy() for k, v in rays.items() if with_doppler or not k.startswith("doppler")}
    ds = dm.Dataset(d)
    bs_fov, ue_fov = fov_args(case)
    if bs_fov is not None or ue_fov is not None:
        kw = {}
        if bs_fov is not None:
            kw["bs_fov"] = bs_fov
        if ue_fov is not None:
            kw["ue_fov"] = ue_fov
        ds.apply_fov(**kw)
    return ds


@pytest.mark.parametrize("variant", [1, 0, 4, 5, 12])
@pytest.mark.parametrize("name", golden_names())
def test_golden(name, variant, capsys):
    import deepmimo_amd as dm
    case, rays, ue_rot, ref = load_golden(name)
    if variant == 12:
        from deepmimo_amd.engine import uniform_stride
        pairs = int(np.prod(case["bs_shape"]) * np.prod(case["ue_shape"]))
        if not (case["freq_domain"] and not case["rx_filter"] and pairs <= 128 and uniform_stride(np.array(case["selected"]))[1] > 0):
            pytest.skip("not a case of the folded kernel")
    dm.config("fd_kernel_variant", variant)
    try:
        ds = _dataset(case, rays)
        H = ds.compute_channels(_dm_params(case, ue_rot))
    finally:
        dm.config("fd_kernel_variant", 0)
    assert H.dtype == np.complex64
    if "channel" in ref:
        assert_channel_close(H, ref["channel"], what=name)
    else:
        sub = case["subsample"]
        assert_channel_close(H[:, :, sub["tx"], :][..., sub["k"]], ref["channel_sub"], what=name)
        # size-independent fingerprint of the full tensor
        Hd = H.astype(np.complex128)
        e = float(np.sum(np.abs(Hd) ** 2))
        assert abs(e - ref["channel_checksum"][0]) <= 4 * TOL_REL * ref["channel_checksum"][0]
    np.testing.assert_array_equal(ds.los, ref["los"])
    np.testing.assert_array_equal(ds.num_paths, ref["num_paths"])
    assert ds.los.dtype == ref["los"].dtype and ds.num_paths.dtype == ref["num_paths"].dtype
    if "fov_mask" in ref:
        np.testing.assert_array_equal(ds["_fov_mask"], ref["fov_mask"])
    else:
        assert ds["_fov_mask"] is None
    np.testing.assert_allclose(ds.power_linear, ref["power_linear"], rtol=1e-6, equal_nan=True)
    assert ds.power_linear.dtype == ref["power_linear"].dtype
    g = ds["_power_linear_ant_gain"]
    assert g.dtype == ref["power_linear_ant_gain"].dtype
    # isotropic: the same float32 value as power_linear.  Dipole: the gain cos^2(pi/2 cos t)/sin t is
    # ill-conditioned near its nulls, where the ~1e-7 rad float32-trig difference of the rotated zenith
    # angle is amplified; bound it relative to the un-attenuated path power.
    iso = case["bs_pattern"] == "isotropic" and case["ue_pattern"] == "isotropic"
    if iso:
        np.testing.assert_allclose(g, ref["power_linear_ant_gain"], rtol=2e-6, atol=1e-30, equal_nan=True)
    else:
        err = np.abs(g - ref["power_linear_ant_gain"])
        bound = 2e-6 * 1.643 ** 2 * ref["power_linear"].astype(np.float64) + 5e-5 * np.abs(ref["power_linear_ant_gain"])
        assert np.array_equal(np.isnan(g), np.isnan(ref["power_linear_ant_gain"]))
        assert np.all(err[~np.isnan(err)] <= bound[~np.isnan(err)])
    # Rotated angles: K1 reproduces NumPy's float32 sin/cos bit for bit (np_sincosf), so what is left is
    # float64 libm noise (1 ulp) amplified by 1/sin(zenith) through arccos / atan2.
    for side in ("aod", "aoa"):
        zen_ref, az_ref = ref[side + "_el_rot"], ref[side + "_az_rot"]
        zen, az = ds[f"_{side}_el_rot"], ds[f"_{side}_az_rot"]
        assert np.array_equal(np.isnan(zen), np.isnan(zen_ref)) and np.array_equal(np.isnan(az), np.isnan(az_ref))
        ok = ~np.isnan(zen_ref)
        assert np.all(np.abs(zen - zen_ref)[ok] <= 1e-11)
        assert np.all(np.abs(np.angle(np.exp(1j * (az - az_ref))))[ok] <= 1e-11)
    if case["freq_domain"]:
        warned = "exceed OFDM symbol duration" in capsys.readouterr().out
        assert warned == bool(ref["warned"])


@pytest.mark.parametrize("name", ["g10_doppler_v3", "g13_doppler_lpf_v3"])
def test_doppler_golden(name):
    """Doppler term: golden from the v3 generator (the only Python definition, SURVEY finding 4)."""
    import deepmimo_amd as dm
    case, rays, ue_rot, ref = load_golden(name)
    ds = _dataset(case, rays, with_doppler=True)
    ds["rt_params"] = {"frequency": 3.5e9}
    p = _dm_params(case, ue_rot)
    p.enable_doppler = 1
    H = ds.compute_channels(p)
    assert_channel_close(H, ref["channel_doppler"], what="doppler")
    p.enable_doppler = 0
    assert_channel_close(ds.compute_channels(p), ref["channel"], what="doppler off")


SHAPES = [
    # n_ue, L, bs, ue, N, selected, extra
    (37, 1, [1, 1], [1, 1], 16, list(range(16)), {}),
    (50, 3, [3, 1], [1, 3], 100, [0, 99, 50, 7, 7], {}),                      # unsorted / repeated subcarriers
    (33, 7, [5, 3], [3, 1], 96, list(range(0, 96, 1)), dict(bs_rot=[0, 90, 0])),
    (20, 16, [4, 4], [2, 2], 128, list(range(1, 128, 2)), dict(ue_rot=[45, -30, 170])),
    (12, 25, [8, 8], [2, 2], 512, list(range(0, 512, 3)), dict(max_delay=40e-6)),
    (9, 32, [16, 2], [1, 2], 64, list(range(64)), {}),
    (6, 25, [16, 16], [4, 4], 1024, list(range(0, 1024, 16)), dict(all_valid=True, max_delay=90e-6)),
    (5, 12, [70, 1], [1, 1], 64, list(range(64)), {}),
    (11, 9, [6, 4], [2, 1], 80, list(range(2, 68, 2)), {}),                   # 48 rows, odd K = 33 on the MFMA kernel
    (7, 5, [8, 4], [1, 1], 64, [5], {}),                                      # a single subcarrier
    (6, 4, [3, 3], [3, 3], 32, [31, -1, 32, 95], {}),                         # indices outside 0..N-1 are plain integers
    (5, 25, [8, 4], [1, 1], 8192, list(range(3, 8192, 13)), dict(max_delay=700e-6)),   # subcarrier numbers beyond 4095 (two-term phase reduction)
    (4, 25, [8, 8], [2, 2], 4096, list(range(4096)), dict(all_valid=True, max_delay=150e-6)),   # 4096 subcarriers: 32-KiB rows
]


SHAPES_SMALL = [
    # few selected subcarriers - the small-output kernel's regime (the reference default is ONE subcarrier)
    (70, 25, [8, 8], [1, 1], 512, [0], {}),
    (70, 25, [8, 8], [2, 2], 512, [0, 1], dict(bs_rot=[10, 0, -30])),
    (41, 10, [8, 1], [1, 1], 64, [3, 9, 27], dict(ue_rot=[0, 20, 40])),                 # K = 3: chunk tail at KC = 2
    (33, 25, [4, 4], [2, 1], 512, list(range(0, 512, 103)), {}),                        # K = 5: chunk tail at KC = 4
    (29, 32, [8, 4], [1, 2], 256, list(range(8)), dict(all_valid=True, max_delay=20e-6)),
    (260, 7, [2, 2], [1, 1], 64, list(range(64)), {}),                                  # more users than one grid pass of waves
]


def _small_kernel_fits(shape):
    n, L, bs, ue, N, sel, extra = shape
    return (bs[0] * bs[1] + ue[0] * ue[1] + len(sel)) * min(L, 32) * 8 <= 156 * 1024


SHAPES_FOLD = [
    # few antenna pairs x uniformly spaced subcarriers - the folded matrix-core kernel's regime (variant 12);
    # DeepMIMO's default arrays are BS 8x1 / UE 1x1 (channel.py:36-46)
    (45, 25, [8, 1], [1, 1], 64, list(range(64)), {}),                                  # 8 pairs: 4 blocks per 32-row tile
    (31, 25, [8, 1], [1, 1], 512, list(range(512)), dict(max_delay=40e-6)),             # 32 blocks = one full chunk
    (23, 10, [4, 4], [1, 1], 256, list(range(256)), dict(bs_rot=[5, -20, 60])),         # 16 pairs
    (19, 25, [8, 2], [2, 1], 512, list(range(0, 512, 2)), dict(all_valid=True, max_delay=45e-6)),   # 32 pairs, stride 2
    (17, 7, [4, 3], [1, 1], 100, list(range(100)), {}),                                 # 12 pairs (rows straddle tiles), K % 16 = 4
    (13, 32, [5, 1], [1, 1], 2048, list(range(7, 2048, 3)), dict(max_delay=95e-6)),     # 5 pairs, K = 681: several chunks + tail, first != 0
    (9, 25, [8, 4], [2, 1], 128, list(range(128)), dict(ue_rot=[10, 20, 30])),          # 64 pairs: one table set per workgroup
    (8, 25, [8, 8], [2, 1], 512, list(range(512)), dict(all_valid=True, max_delay=45e-6)),   # 128 pairs: the kernel's upper limit
    (14, 9, [7, 5], [1, 1], 300, list(range(5, 305)), {}),                               # 35 pairs (odd), K % 16 = 12, shared tables
    (12, 25, [2, 1], [1, 1], 1024, list(range(1024)), dict(all_valid=True, max_delay=90e-6)),   # 2 pairs: 16 blocks per tile
    (11, 3, [8, 1], [1, 1], 64, list(range(10, 43)), {}),                               # K = 33: a lone subcarrier in the last block
    (10, 25, [1, 1], [1, 1], 4096, list(range(4096)), dict(all_valid=True, max_delay=300e-6)),  # one pair, 4096 subcarriers
    # round 3: power-of-two pair counts take the scalar-row-offset stores; tiles that hold rows past the chunk or a
    # partial last block take their checked form - every power of two with a tail, the half-wave row term in each regime
    (9, 25, [4, 1], [1, 1], 64, list(range(50)), {}),                                   # 4 pairs, K % 16 = 2: lane >> 5 moves the block
    (9, 12, [2, 1], [1, 1], 64, list(range(3, 24)), {}),                                # 2 pairs, K = 21
    (9, 25, [1, 1], [1, 1], 64, list(range(17)), {}),                                   # 1 pair, K = 17
    (9, 25, [4, 4], [1, 1], 128, list(range(0, 80, 2)), dict(all_valid=True)),          # 16 pairs, K = 40 (tail 8), stride 2
    (7, 25, [8, 4], [1, 1], 64, list(range(53)), {}),                                   # 32 pairs, K = 53 (tail 5): one block per tile
    (6, 25, [8, 4], [2, 1], 128, list(range(72)), dict(bs_rot=[0, 0, 30])),             # 64 pairs (shared tables), K = 72 (tail 8)
    (5, 25, [8, 8], [2, 1], 64, list(range(19)), {}),                                   # 128 pairs, K = 19: a block spans four tiles
]


def _fold_applicable(shape):
    n, L, bs, ue, N, sel, extra = shape
    from deepmimo_amd.engine import uniform_stride
    return uniform_stride(np.asarray(sel))[1] > 0 and bs[0] * bs[1] * ue[0] * ue[1] <= 128


ALL_SHAPES = SHAPES + SHAPES_SMALL + SHAPES_FOLD


# 1 fp32 vector, 0 automatic, 4 / 5 / 10 matrix cores in 4- / 8- / 16-wave workgroups, 9 small-output, 12 folded, and the
# measurement knobs of the C-ABI (3 plain stores, 8 one workgroup per item, 11 resident grid): every variant the header
# documents is parity-checked
@pytest.mark.parametrize("variant", [1, 0, 4, 5, 9, 12, 3, 8, 10, 11])
@pytest.mark.parametrize("shape", ALL_SHAPES, ids=[f"s{i}" for i in range(len(ALL_SHAPES))])
def test_vs_oracle_shapes(shape, variant):
    """Seeded synthetic rays at ragged shapes (K not a multiple of 64, odd panels, 1..32 paths)."""
    import deepmimo_amd as dm
    from oracle import oracle_np as onp
    n, L, bs, ue, N, sel, extra = shape
    if variant == 9 and not _small_kernel_fits(shape):
        pytest.skip("one user's factor tables exceed the LDS of the small-output kernel")
    if variant == 12 and not _fold_applicable(shape):
        pytest.skip("the folded kernel needs uniformly spaced subcarriers and at most 128 antenna pairs")
    if variant in (3, 8, 10, 11) and shape not in SHAPES[2:7] + SHAPES_FOLD[3:5]:
        pytest.skip("measurement knobs are checked on a subset of the shapes")
    rays = onp.synth_rays(n, L, seed=1000 + n, all_valid=extra.get("all_valid", False),
                          max_delay=extra.get("max_delay", 2e-6))
    case = dict(bs_shape=bs, ue_shape=ue, bs_spacing=0.5, ue_spacing=0.37, bs_rot=extra.get("bs_rot", [0, 0, 0]),
                bs_pattern="isotropic", ue_pattern="isotropic", num_paths=L, freq_domain=1, subcarriers=N,
                selected=sel, bandwidth=20e6, rx_filter=0, bs_fov=None, ue_fov=None)
    ue_rot = np.array(extra.get("ue_rot", [0, 0, 0]))
    ref = onp.compute_channels(rays, oracle_params(case, ue_rot))
    dm.config("fd_kernel_variant", variant)
    try:
        ds = dm.Dataset(dict(rays))
        H = ds.compute_channels(_dm_params(case, ue_rot))
    finally:
        dm.config("fd_kernel_variant", 0)
    assert_channel_close(H, ref["channel"], what=str(shape[:5]))
    np.testing.assert_array_equal(ds.los, ref["los"])
    np.testing.assert_array_equal(ds.num_paths, ref["num_paths"])


def test_time_domain_and_lpf_vs_oracle():
    import deepmimo_amd as dm
    from oracle import oracle_np as onp
    rays = onp.synth_rays(40, 9, seed=77)
    for fd, lpf in ((0, 0), (1, 1)):
        case = dict(bs_shape=[4, 2], ue_shape=[2, 1], bs_spacing=0.5, ue_spacing=0.5, bs_rot=[20, 10, -60],
                    bs_pattern="halfwave-dipole", ue_pattern="isotropic", num_paths=9, freq_domain=fd,
                    subcarriers=48, selected=list(range(0, 48, 5)), bandwidth=15e6, rx_filter=lpf,
                    bs_fov=[150, 100], ue_fov=None)
        ue_rot = np.array([0, 15, 0])
        ref = onp.compute_channels(rays, oracle_params(case, ue_rot), bs_fov=np.array([150, 100]),
                                   ue_fov=np.array([360, 180]))
        ds = _dataset(case, rays)
        H = ds.compute_channels(_dm_params(case, ue_rot))
        assert_channel_close(H, ref["channel"], what=f"fd={fd} lpf={lpf}")
        np.testing.assert_array_equal(ds["_fov_mask"], ref["_fov_mask"])


def test_empty_and_all_nan():
    import deepmimo_amd as dm
    from oracle import oracle_np as onp
    rays = onp.synth_rays(0, 5, seed=1)
    ds = dm.Dataset(dict(rays))
    H = ds.compute_channels()
    assert H.shape == (0, 1, 8, 1)
    rays = onp.synth_rays(5, 4, seed=2)
    for k in onp.RAY_KEYS:
        rays[k][:] = np.nan
    ds = dm.Dataset(dict(rays))
    H = ds.compute_channels()
    assert H.shape == (5, 1, 8, 1) and np.all(H == 0)
    assert ds.los.tolist() == [-1] * 5 and ds.num_paths.tolist() == [0] * 5


def test_cache_semantics_and_aliases():
    """dataset.py:197-222, 358-378, 515-535: what is cached, what a rotation / FoV change drops."""
    import deepmimo_amd as dm
    from oracle import oracle_np as onp
    rays = onp.synth_rays(16, 6, seed=3)
    ds = dm.Dataset(dict(rays))
    p = dm.ChannelGenParameters()
    H1 = ds.compute_channels(p)
    assert ds.channel is H1 and ds.ch is H1 and ds["channels"] is H1
    assert "_aod_el_rot" in ds.keys() and "los" in ds.keys()
    ds.apply_fov(bs_fov=np.array([90, 60]))
    assert "channel" not in ds.keys() and "los" not in ds.keys() and "_aod_el_rot" in ds.keys()
    n_before = ds.num_paths.copy()                  # lazy: triggers stage 1 with the new FoV
    assert ds["_fov_mask"].dtype == bool
    p.bs_antenna.rotation = np.array([0, 0, 90])
    H2 = ds.compute_channels(p)
    assert not np.array_equal(n_before, ds.num_paths) or not np.allclose(H1, H2)
    # lazy channel with default params on a fresh dataset
    ds2 = dm.Dataset(dict(rays))
    assert ds2.channel.shape == (16, 1, 8, 1)
    with pytest.raises(KeyError):
        ds2["no_such_matrix"]


def test_side_products_stay_in_hbm_until_read():
    """The reference caches los / num_paths / rotated angles / powers as NumPy arrays inside compute_channels; here
    they are registered under the same keys but copied out of HBM only when read (dataset.py:_DeviceSide)."""
    import deepmimo_amd as dm
    from deepmimo_amd.dataset import _DeviceSide
    from oracle import oracle_np as onp
    rays = onp.synth_rays(32, 7, seed=21)
    ds = dm.Dataset(dict(rays))
    ds.apply_fov(bs_fov=np.array([120, 90]))
    ds.compute_channels(dm.ChannelGenParameters())
    side_keys = ("los", "num_paths", "power_linear", "_power_linear_ant_gain", "_fov_mask",
                 "_aod_el_rot", "_aoa_az_rot", "_aod_el_rot_fov", "_aoa_az_rot_fov")
    assert all(k in ds.keys() for k in side_keys)
    assert all(isinstance(ds._data[k], _DeviceSide) for k in side_keys)
    ref = onp.compute_channels(rays, onp.make_params(), bs_fov=np.array([120, 90]))
    np.testing.assert_array_equal(ds.los, ref["los"])                       # attribute access
    assert isinstance(ds._data["los"], np.ndarray) and isinstance(ds._data["num_paths"], _DeviceSide)
    np.testing.assert_array_equal(ds["n_paths"] if "n_paths" in dm.consts.DATASET_ALIASES else ds["num_paths"],
                                  ref["num_paths"])                          # item access
    np.testing.assert_array_equal(ds.get("_fov_mask"), ref["_fov_mask"])    # Mapping.get
    m = ref["_fov_mask"]
    np.testing.assert_allclose(ds["_aod_el_rot_fov"], np.where(m, ref["_aod_el_rot"], np.nan), atol=1e-11, equal_nan=True)
    d = ds.to_dict()                                                         # everything else lands on export
    assert not any(isinstance(v, _DeviceSide) for v in d.values())
    assert d["power_linear"].dtype == np.float32 and d["_power_linear_ant_gain"].dtype == np.float32
    ds.apply_fov()                                                           # invalidation drops placeholders too
    assert "los" not in ds.keys() and "_fov_mask" not in ds.keys()


def test_dataset_pickles_with_side_products_still_in_hbm():
    """the `_DeviceSide` placeholders are closures - pickling copies them out first; deferred products
    (rotated angles, powers: a second stage-1 pass on first read) come out identical to an eager read."""
    import pickle
    import deepmimo_amd as dm
    from deepmimo_amd.dataset import _DeviceSide
    from oracle import oracle_np as onp
    rays = onp.synth_rays(40, 9, seed=33)
    p = dm.ChannelGenParameters()
    p.bs_antenna.rotation = np.array([10, 20, 30])
    p.ue_antenna.rotation = np.array([[0, 30], [0, 20], [-45, 45]])          # random per-user range, drawn once
    ds = dm.Dataset(dict(rays))
    H = ds.compute_channels(p)
    assert isinstance(ds._data["_aod_el_rot"], _DeviceSide) and isinstance(ds._data["power_linear"], _DeviceSide)
    ds2 = pickle.loads(pickle.dumps(ds))
    assert not any(isinstance(v, _DeviceSide) for v in ds2._data.values())
    ref = onp.compute_channels(rays, onp.make_params(bs_antenna=dict(rotation=np.array([10, 20, 30])),
                                                     ue_antenna=dict(rotation=ds["_ue_rotation_resolved"])))
    np.testing.assert_array_equal(ds2.los, ref["los"])
    np.testing.assert_array_equal(ds2.channel, H)
    for k in ("_aod_el_rot", "_aoa_az_rot"):
        ok = ~np.isnan(ref[k])
        assert np.array_equal(np.isnan(ds2[k]), ~ok) and np.all(np.abs(ds2[k] - ref[k])[ok] <= 1e-11)
    np.testing.assert_allclose(ds2.power_linear, ref["power_linear"], rtol=1e-6, equal_nan=True)
    np.testing.assert_array_equal(ds["_aod_el_rot"], ds2["_aod_el_rot"])      # the original reads the same values later


def test_macro_dataset_fan_out():
    import deepmimo_amd as dm
    from oracle import oracle_np as onp
    a, b = onp.synth_rays(8, 4, seed=5), onp.synth_rays(11, 4, seed=6)
    md = dm.MacroDataset([dm.Dataset(dict(a)), dm.Dataset(dict(b))])
    Hs = md.compute_channels()
    assert isinstance(Hs, list) and Hs[0].shape[0] == 8 and Hs[1].shape[0] == 11
    assert [x.shape[0] for x in md.los] == [8, 11]
    single = dm.MacroDataset([dm.Dataset(dict(a))])
    assert single.compute_channels().shape[0] == 8


def test_random_ue_rotation_matches_reference_rng_order():
    """(3, 2) UE rotation = random range, drawn after np.random.seed(1001) (dataset.py:250, 334-338)."""
    case, rays, ue_rot, ref = load_golden("g09_random_ue_rot")
    ds = _dataset(case, rays)
    H = ds.compute_channels(_dm_params(case, ue_rot))
    assert_channel_close(H, ref["channel"], what="random ue rot")
    np.testing.assert_array_equal(ds["_fov_mask"], ref["fov_mask"])


def test_generate_from_scenario_folder(tmp_path, monkeypatch):
    """dm.generate(scen_name, load_params, ch_gen_params) (core.py:36-61): load + compute_channels, with a
    scenario written in the reference's on-disk layout and two TX points -> MacroDataset fan-out."""
    import json
    import scipy.io
    import deepmimo_amd as dm
    from oracle import oracle_np as onp
    folder = tmp_path / "deepmimo_scenarios" / "toy"
    folder.mkdir(parents=True)
    params = {"rt_params": {"frequency": 28e9}, "scene": {"num_scenes": 1}, "materials": {},
              "txrx_sets": {"txrx_set_0": {"id": 0, "is_tx": True, "is_rx": False, "num_points": 2, "name": "bs"},
                            "txrx_set_1": {"id": 1, "is_tx": False, "is_rx": True, "num_points": 20, "name": "ue"}}}
    (folder / "params.json").write_text(json.dumps(params))
    rays = [onp.synth_rays(20, 6, seed=30 + t) for t in range(2)]
    for t in range(2):
        for k, v in rays[t].items():
            scipy.io.savemat(str(folder / dm.core.get_mat_filename(k, 0, t, 1)), {k: v})
    monkeypatch.chdir(tmp_path)
    p = dm.ChannelGenParameters()
    p.bs_antenna.shape = np.array([4, 2])
    p.ofdm.selected_subcarriers = np.arange(0, 512, 32)
    md = dm.generate("toy", {"max_paths": 5}, p)
    assert isinstance(md, dm.MacroDataset) and len(md) == 2
    for t in range(2):
        sub = {k: (v[:, :5] if v.ndim == 2 and k not in ("rx_pos", "tx_pos") else v) for k, v in rays[t].items()}
        op = onp.make_params(bs_antenna=dict(shape=[4, 2]), ofdm=dict(selected_subcarriers=np.arange(0, 512, 32)))
        ref = onp.compute_channels(sub, op)
        assert_channel_close(md[t].channel, ref["channel"], what=f"generate tx {t}")
        np.testing.assert_array_equal(md[t].los, ref["los"])


def test_errors_are_loud():
    import deepmimo_amd as dm
    from deepmimo_amd._native import NativeError
    from oracle import oracle_np as onp
    rays = onp.synth_rays(4, 3, seed=9)
    ds = dm.Dataset(dict(rays))
    p = dm.ChannelGenParameters()
    p.bs_antenna.radiation_pattern = "patch"
    with pytest.raises((AssertionError, NotImplementedError)):
        ds.compute_channels(p)
    p = dm.ChannelGenParameters()
    p.bs_antenna.rotation = np.array([1, 2])
    with pytest.raises(AssertionError):
        ds.compute_channels(p)
    p = dm.ChannelGenParameters()
    p.ofdm.bandwidth = 0.0
    with pytest.raises(NativeError):
        ds.compute_channels(p)
    wide = dm.Dataset(dict(onp.synth_rays(4, 40, seed=9)))           # beam-space kernel: at most 32 paths, loudly
    p = dm.ChannelGenParameters()
    p.num_paths = 40
    with pytest.raises(NativeError, match="32 paths"):
        wide.compute_beam_channels(np.ones((2, 8)), p)
    dm.config("use_gpu", False)
    try:
        with pytest.raises(RuntimeError):
            dm.Dataset(dict(rays)).compute_channels()
    finally:
        dm.config("use_gpu", True)


def test_device_loader_matches_scipy(tmp_path):
    """load(..., device='cuda'): .mat payload -> HBM -> row-major float32 by the device pass, equal to the
    reference's scipy.io.loadmat + slicing (core.py:241-254), then channels from the device-resident rays."""
    import json
    import scipy.io
    import torch
    import deepmimo_amd as dm
    from oracle import oracle_np as onp
    rays = onp.synth_rays(333, 25, seed=21)
    folder = tmp_path / "scen"
    folder.mkdir()
    params = {"rt_params": {"frequency": 3.5e9}, "scene": {"num_scenes": 1}, "materials": {},
              "txrx_sets": {"txrx_set_0": {"id": 0, "is_tx": True, "is_rx": False, "num_points": 1, "name": "bs"},
                            "txrx_set_1": {"id": 1, "is_tx": False, "is_rx": True, "num_points": 333, "name": "ue"}}}
    (folder / "params.json").write_text(json.dumps(params))
    for k, v in rays.items():
        vv = v.astype(np.float64) if k == "delay" else v                       # one file in another dtype
        scipy.io.savemat(str(folder / dm.core.get_mat_filename(k, 0, 0, 1)), {k: vv}, do_compression=(k == "phase"))
    sel = np.arange(332, -1, -3)
    host = dm.load(str(folder), max_paths=10, rx_sets={1: sel})
    devd = dm.load(str(folder), max_paths=10, rx_sets={1: sel}, device="cuda")
    for k in dm.consts.RAY_FIELDS:
        assert isinstance(devd[k], torch.Tensor) and devd[k].is_cuda and devd[k].dtype == torch.float32
        np.testing.assert_array_equal(devd[k].cpu().numpy(), host[k].astype(np.float32))
    np.testing.assert_array_equal(devd.rx_pos, host.rx_pos)
    from deepmimo_amd import matio
    cap = matio.STAGING_CAP_BYTES
    try:
        matio.STAGING_CAP_BYTES = 1                                             # every file a pipeline run of its own
        dev1 = dm.load(str(folder), max_paths=10, rx_sets={1: sel}, device="cuda")
    finally:
        matio.STAGING_CAP_BYTES = cap
    for k in dm.consts.RAY_FIELDS:
        assert torch.equal(dev1[k].view(torch.int32), devd[k].view(torch.int32))           # bits: NaN padding included
    p = dm.ChannelGenParameters()
    p.bs_antenna.shape = np.array([8, 4])
    Hd, Hh = devd.compute_channels(p), host.compute_channels(p)
    assert np.array_equal(Hd, Hh)
    np.testing.assert_array_equal(devd.los, host.los)


@pytest.mark.parametrize("cfg", [dict(bs=[32, 1], ue=[1, 1], nb=16, L=9, rot=[0, 0, -135]),
                                 dict(bs=[8, 8], ue=[2, 2], nb=64, L=25, rot=[0, 0, 0]),
                                 dict(bs=[4, 2], ue=[3, 1], nb=5, L=4, rot=[10, 20, 30])])
def test_beam_codebook_projection(cfg):
    """Fused consumer: Dataset.compute_beam_channels(F) == F @ compute_channels() (docs/manual.ipynb cell 105),
    for a steering-vector codebook and for an arbitrary un-normalised complex one."""
    import deepmimo_amd as dm
    from oracle import oracle_np as onp
    rays = onp.synth_rays(70, cfg["L"], seed=55)
    p = dm.ChannelGenParameters()
    p.bs_antenna.shape = np.array(cfg["bs"])
    p.ue_antenna.shape = np.array(cfg["ue"])
    p.bs_antenna.rotation = np.array(cfg["rot"])
    p.ofdm.selected_subcarriers = np.arange(0, 512, 7)
    op = onp.make_params(bs_antenna=dict(shape=cfg["bs"], rotation=np.array(cfg["rot"])), ue_antenna=dict(shape=cfg["ue"]),
                         ofdm=dict(selected_subcarriers=np.arange(0, 512, 7)))
    Href = onp.compute_channels(rays, op)["channel"].astype(np.complex128)
    m_tx = cfg["bs"][0] * cfg["bs"][1]
    beams = np.around(np.linspace(-60, 60, cfg["nb"]), 2)
    F1 = np.array([dm.steering_vec(np.array(cfg["bs"]), phi=azi).squeeze() for azi in beams])
    rng = np.random.default_rng(1)
    F2 = (rng.normal(size=(cfg["nb"], m_tx)) + 1j * rng.normal(size=(cfg["nb"], m_tx))) * 37.5
    ds = dm.Dataset(dict(rays))
    for F in (F1, F2):
        Y = ds.compute_beam_channels(F, p)
        Yref = (F @ Href).astype(np.complex64)                      # [n_ue, M_rx, n_beams, K]
        assert Y.shape == Yref.shape == (70, cfg["ue"][0] * cfg["ue"][1], cfg["nb"], 74)
        assert_channel_close(Y, Yref, what="beam-space channel")
    with pytest.raises(ValueError):
        ds.compute_beam_channels(np.ones((4, m_tx + 1)), p)


@pytest.mark.parametrize("cfg", [dict(bs=[32, 1], ue=[1, 1], nb=16, L=9, rot=[0, 0, -135], sel=np.arange(0, 512, 7)),   # the notebook's sweep
                                 dict(bs=[8, 8], ue=[2, 2], nb=64, L=25, rot=[0, 0, 0], sel=np.arange(0, 512, 5)),     # 256 rows: 8 tiles
                                 dict(bs=[4, 2], ue=[3, 1], nb=5, L=4, rot=[10, 20, 30], sel=np.array([3])),          # 15 rows, one subcarrier
                                 dict(bs=[8, 4], ue=[4, 3], nb=33, L=32, rot=[0, 0, 0], sel=np.arange(100, 400)),       # 396 rows: two row blocks
                                 dict(bs=[8, 1], ue=[1, 1], nb=3, L=12, rot=[0, 0, 0], sel=np.arange(512))])           # default arrays
def test_beam_power_fused_reduction(cfg):
    """dmx_beam_power / Dataset.compute_beam_power: np.abs(F @ H).mean(axis=1).mean(axis=-1) -> dBm -> argmax
    (docs/manual.ipynb cells 105, 110, 112) without H or F @ H ever being written; against the oracle's H reduced in
    float64.  Mean amplitudes: 1e-5 of the user's strongest beam (the contraction's error is relative to the user's peak,
    as for the channel itself: a beam 60 dB down cannot be met to rtol 1e-5 of ITS value by any fp32 result), and rtol
    1e-5 element-wise for every beam within 30 dB of the strongest; dBm within one rounding step; best beam exact except
    near-ties."""
    import deepmimo_amd as dm
    from deepmimo_amd.dataset import _engine
    from oracle import oracle_np as onp
    n = 70
    rays = onp.synth_rays(n, cfg["L"], seed=56)
    p = dm.ChannelGenParameters()
    p.bs_antenna.shape, p.ue_antenna.shape = np.array(cfg["bs"]), np.array(cfg["ue"])
    p.bs_antenna.rotation = np.array(cfg["rot"])
    p.num_paths = cfg["L"]
    p.ofdm.selected_subcarriers = cfg["sel"]
    op = onp.make_params(bs_antenna=dict(shape=cfg["bs"], rotation=np.array(cfg["rot"])), ue_antenna=dict(shape=cfg["ue"]),
                         num_paths=cfg["L"], ofdm=dict(selected_subcarriers=cfg["sel"]))
    ref = onp.compute_channels(rays, op)
    Href = ref["channel"].astype(np.complex128)
    m_tx = cfg["bs"][0] * cfg["bs"][1]
    F1 = np.array([dm.steering_vec(np.array(cfg["bs"]), phi=azi).squeeze() for azi in np.around(np.linspace(-60, 60, cfg["nb"]), 2)])
    F1 = F1.reshape(cfg["nb"], m_tx)
    rng = np.random.default_rng(2)
    F2 = (rng.normal(size=(cfg["nb"], m_tx)) + 1j * rng.normal(size=(cfg["nb"], m_tx))) * 11.0
    for F in (F1, F2):
        want_amp = np.abs(F @ Href).mean(axis=1).mean(axis=-1)                  # [n, nb] float64
        ds = dm.Dataset(dict(rays))
        pwr, best = ds.compute_beam_power(F, p, return_best=True)
        amp = ds["beam_mean_amplitude"]
        assert amp.shape == (n, cfg["nb"]) and amp.dtype == np.float32 and pwr.dtype == np.float64
        np.testing.assert_array_equal(ds.los, ref["los"])
        has = ref["los"] != -1
        assert has.sum() > 0
        peak = want_amp[has].max(axis=1, keepdims=True)
        assert np.all(np.abs(amp[has] - want_amp[has]) <= 1e-5 * peak)
        strong = want_amp[has] >= peak * 10 ** (-30 / 20)
        np.testing.assert_allclose(amp[has][strong], want_amp[has][strong], rtol=1e-5, atol=0)
        assert np.all(amp[~has] == 0)
        # cells 105 / 110 / 112 on the reference amplitudes, in the notebook's float32 (channel is complex64)
        want_pwr = np.zeros((n, cfg["nb"])) * np.nan
        want_pwr[has] = np.around(20 * np.log10(want_amp[has].astype(np.float32)) + 30, 1)
        assert np.array_equal(np.isnan(pwr), np.isnan(want_pwr))
        assert np.nanmax(np.abs(pwr - want_pwr)) <= 0.1 + 1e-4                  # a value on a rounding boundary may step
        assert np.mean(np.abs(pwr[has] - want_pwr[has]) < 1e-4) > 0.98
        want_best = np.argmax(want_pwr, axis=1).astype(float)
        want_best[~has] = np.nan
        assert np.array_equal(np.isnan(best), np.isnan(want_best))
        for u in np.nonzero(has & (best != want_best))[0]:                       # only (near-)ties may differ
            assert abs(want_pwr[u, int(best[u])] - want_pwr[u, int(want_best[u])]) <= 0.1 + 1e-4
        # the kernel's own argmax (first maximum of the un-rounded means)
        eng = _engine()
        prep = eng.prepare(eng.upload_rays(rays), p.validate(n))
        amp_d, best_d = eng.beam_power(prep, F)
        np.testing.assert_array_equal(amp_d.cpu().numpy(), amp)
        bd = best_d.cpu().numpy()
        assert np.all(bd[~has] == -1) and np.array_equal(bd[has], np.argmax(amp[has], axis=1))
        # a user sub-range is bit-identical to the same rows of the full call (what sharding relies on)
        part, _ = eng.beam_power(prep, F, user_begin=11, user_count=30)
        np.testing.assert_array_equal(part.cpu().numpy(), amp[11:41])
    with pytest.raises(ValueError):
        ds.compute_beam_power(np.ones((4, m_tx + 1)), p)


def test_pathloss_matches_reference_formula():
    """Dataset.compute_pathloss (dataset.py:541-566), restated inline in NumPy with the reference's dtypes."""
    import deepmimo_amd as dm
    from oracle import oracle_np as onp
    rays = onp.synth_rays(500, 25, seed=91)
    rays["power"][7, 3] = np.nan                                   # NaN in the middle of a row
    ds = dm.Dataset(dict(rays))
    for coherent in (True, False):
        with np.errstate(invalid="ignore", divide="ignore"):
            g = np.sqrt(10 ** (rays["power"] / 10)).astype(np.complex64)
            if coherent:
                g = g * np.exp(1j * np.deg2rad(rays["phase"]))
            tp = np.abs(np.nansum(g, axis=1)) ** 2
            want = np.full_like(tp, np.nan)
            want[tp > 0] = -10 * np.log10(tp[tp > 0])
        got = ds.compute_pathloss(coherent)
        assert got.dtype == np.float32 and got.shape == (500,)
        assert np.array_equal(np.isnan(got), np.isnan(want))
        np.testing.assert_allclose(got[~np.isnan(want)], want[~np.isnan(want)], rtol=0, atol=2e-4)   # dB
    assert ds.pl is ds.pathloss


@pytest.mark.parametrize("N,K", [(512, 512), (256, 100), (96, 96)])
def test_rx_filter_fft_and_mfma_path(N, K):
    """rx_filter = 1 at shapes where the FFT form of the gains (power-of-two N) and the MFMA contraction with
    table-loaded gains are used (M = 128 rows), plus a non-power-of-two N on the direct kernel; with Doppler."""
    import deepmimo_amd as dm
    from oracle import oracle_np as onp
    rays = onp.synth_rays(24, 25, seed=N + K, max_delay=N / 10e6 * 1.1, with_doppler=True)
    sel = np.arange(K) if K == N else np.sort(np.random.default_rng(K).choice(N, K, replace=False))
    case = dict(bs_shape=[8, 4], ue_shape=[2, 2], bs_spacing=0.5, ue_spacing=0.5, bs_rot=[0, 10, 45],
                bs_pattern="isotropic", ue_pattern="isotropic", num_paths=25, freq_domain=1, subcarriers=N,
                selected=list(sel), bandwidth=10e6, rx_filter=1, bs_fov=None, ue_fov=None)
    ue_rot = np.array([0, 0, 0])
    op = oracle_params(case, ue_rot)
    for dop in (0, 1):
        op["enable_doppler"] = dop
        ref = onp.compute_channels(rays, op, doppler=dict(vel=rays["doppler_vel"], acc=rays["doppler_acc"], carrier_freq=28e9))
        ds = dm.Dataset(dict(rays))
        ds["rt_params"] = {"frequency": 28e9}
        p = _dm_params(case, ue_rot)
        p.enable_doppler = dop
        H = ds.compute_channels(p)
        assert_channel_close(H, ref["channel"], what=f"lpf N={N} K={K} doppler={dop}")


@pytest.mark.parametrize("arrays", ["mfma", "valu", "dma256", "dma320"])
@pytest.mark.parametrize("selection", ["all512", "first200", "random100", "offset512", "even256", "wrap600"])
def test_rx_filter_fft512_variants(selection, arrays):
    """Every instantiation of the N = 512 wave-per-user FFT (k3_lpf_fft512): selected subcarriers 0..K-1 stored from
    registers (K = 512 unguarded, K = 200 guarded, 512..1023 = the same bins through the stride promise) or any selection
    through the buffer; packed f16 table for the matrix-core contraction or float table for the vector kernel; Doppler on
    and off; users with 0, 1 and all paths; delays that are whole samples (np.sinc(0) = 1 taps, channel.py:166-168).
    dma256 / dma320: 256 and 320 antenna pairs with K a multiple of 16 take the LDS-DMA form of the contraction
    (k2_fd_mfma<.., GSRC = 3>): whole and ragged row blocks (8 and 2 tiles per strip: both counted waits), the prefetch
    chain across work items broken by the user without paths."""
    import deepmimo_amd as dm
    from oracle import oracle_np as onp
    N = 512
    rays = onp.synth_rays(21, 25, seed=512 + len(selection) + len(arrays), max_delay=N / 10e6 * 1.1, with_doppler=True)
    rays["delay"][3, :6] = (np.arange(6) * 37 / 10e6).astype(rays["delay"].dtype)      # dn = 0, 37, 74, ... samples
    rays["delay"][4, 0] = 511 / 10e6
    for k in rays:                                                                       # user 5: one path, user 6: none
        if rays[k].ndim == 2 and rays[k].shape[1] == 25:
            rays[k][5, 1:] = np.nan
            rays[k][6, :] = np.nan
    sel = {"all512": np.arange(512), "first200": np.arange(200), "offset512": np.arange(512, 1024),
           "even256": np.arange(0, 512, 2), "wrap600": np.arange(600),        # more subcarriers than bins: the generic FFT kernel
           "random100": np.sort(np.random.default_rng(7).choice(N, 100, replace=False))}[selection]
    bs, ue = {"mfma": ([8, 4], [2, 2]), "valu": ([2, 1], [1, 1]), "dma256": ([8, 8], [2, 2]), "dma320": ([10, 8], [2, 2])}[arrays]
    case = dict(bs_shape=bs, ue_shape=ue, bs_spacing=0.5, ue_spacing=0.5, bs_rot=[0, 10, 45],
                bs_pattern="isotropic", ue_pattern="isotropic", num_paths=25, freq_domain=1, subcarriers=N,
                selected=list(sel), bandwidth=10e6, rx_filter=1, bs_fov=None, ue_fov=None)
    ue_rot = np.array([0, 0, 0])
    op = oracle_params(case, ue_rot)
    for dop in (0, 1):
        op["enable_doppler"] = dop
        ref = onp.compute_channels(rays, op, doppler=dict(vel=rays["doppler_vel"], acc=rays["doppler_acc"], carrier_freq=28e9))
        ds = dm.Dataset(dict(rays))
        ds["rt_params"] = {"frequency": 28e9}
        p = _dm_params(case, ue_rot)
        p.enable_doppler = dop
        H = ds.compute_channels(p)
        assert_channel_close(H, ref["channel"], what=f"lpf512 {selection} {arrays} doppler={dop}")


@pytest.mark.parametrize("arrays", ["mfma", "valu"])
@pytest.mark.parametrize("selection", ["all", "first_half", "random", "offset"])
@pytest.mark.parametrize("N", [64, 128, 256, 1024])
def test_rx_filter_fft_pow2_variants(N, selection, arrays):
    """k3_lpf_fft_pow2: the register FFT of the other default OFDM sizes - 8, 4 and 2 paths per wave (N = 64, 128, 256:
    the last lane group runs past a user's path count) and 16 points per lane with a radix-16 last pass (N = 1024);
    selections stored from registers or through the buffer, packed or float table, Doppler on and off, users with 0, 1, 2
    and all paths, whole-sample delays."""
    import deepmimo_amd as dm
    from oracle import oracle_np as onp
    n_ue = 9 if N == 1024 else 14
    rays = onp.synth_rays(n_ue, 25, seed=N + len(selection) + len(arrays), max_delay=N / 10e6 * 1.1, with_doppler=True)
    rays["delay"][3, :6] = (np.arange(6) * (N // 7) / 10e6).astype(rays["delay"].dtype)     # whole samples, 0 included
    rays["delay"][4, 0] = (N - 1) / 10e6
    for k in rays:                                                   # user 5: one path, user 6: none, user 7: two
        if rays[k].ndim == 2 and rays[k].shape[1] == 25:
            rays[k][5, 1:] = np.nan
            rays[k][6, :] = np.nan
            rays[k][7, 2:] = np.nan
    rays["power"][8, ~np.isnan(rays["power"][8])] = -80.0            # equal powers: a lost path (the last one sits alone in its
    #                                                                  lane-group iteration) is an error of 1 / n_paths, not of its luck
    sel = {"all": np.arange(N), "first_half": np.arange(N // 2), "offset": np.arange(N, 2 * N),
           "random": np.sort(np.random.default_rng(N).choice(N, max(5, N // 5), replace=False))}[selection]
    bs, ue = ([8, 4], [2, 2]) if arrays == "mfma" else ([2, 1], [1, 1])
    case = dict(bs_shape=bs, ue_shape=ue, bs_spacing=0.5, ue_spacing=0.5, bs_rot=[0, 10, 45],
                bs_pattern="isotropic", ue_pattern="isotropic", num_paths=25, freq_domain=1, subcarriers=N,
                selected=list(sel), bandwidth=10e6, rx_filter=1, bs_fov=None, ue_fov=None)
    ue_rot = np.array([0, 0, 0])
    op = oracle_params(case, ue_rot)
    for dop in (0, 1):
        op["enable_doppler"] = dop
        ref = onp.compute_channels(rays, op, doppler=dict(vel=rays["doppler_vel"], acc=rays["doppler_acc"], carrier_freq=28e9))
        ds = dm.Dataset(dict(rays))
        ds["rt_params"] = {"frequency": 28e9}
        p = _dm_params(case, ue_rot)
        p.enable_doppler = dop
        H = ds.compute_channels(p)
        assert_channel_close(H, ref["channel"], what=f"lpf N={N} {selection} {arrays} doppler={dop}")


def test_sionna_export_of_time_domain_channels():
    """(a, tau) samples in Sionna's layout (reference: integrations/sionna_adapter.py:174-200) from TD channels."""
    import deepmimo_amd as dm
    from deepmimo_amd.sionna_adapter import DeepMIMOSionnaAdapter
    from oracle import oracle_np as onp
    rays = [onp.synth_rays(12, 6, seed=40 + b) for b in range(2)]
    p = dm.ChannelGenParameters()
    p.bs_antenna.shape, p.ue_antenna.shape, p.freq_domain, p.num_paths = np.array([4, 1]), np.array([2, 1]), 0, 6
    md = dm.MacroDataset([dm.Dataset(dict(r)) for r in rays])
    md.compute_channels(p)
    ad = DeepMIMOSionnaAdapter(md, bs_idx=np.array([[0, 1]]), ue_idx=np.array([[0, 1, 2], [3, 4, 5]]))
    assert len(ad) == 2 and ad.ch_shape == (3, 2, 2, 4, 6, 1) and ad.t_shape == (3, 2, 6)
    samples = list(ad())
    assert len(samples) == 2
    a, tau = samples[1]
    assert a.dtype == np.complex64 and tau.dtype == np.float32
    np.testing.assert_array_equal(a[2, :, 1, :, :, 0], md[1].channel[5])
    valid = ~np.isnan(rays[1]["power"][5])
    np.testing.assert_array_equal(tau[2, 1, :valid.sum()], rays[1]["delay"][5][valid])
    assert np.all(tau[2, 1, valid.sum():] == 0)
    p.freq_domain = 1
    md.compute_channels(p)
    with pytest.raises(ValueError):
        DeepMIMOSionnaAdapter(md)


def _aux(name):
    from tests._cases import GOLDEN_DIR
    return np.load(os.path.join(GOLDEN_DIR, name), allow_pickle=False)


def test_pathloss_reference_golden():
    """Dataset.compute_pathloss against what the REAL reference returned (tests/golden/aux_pathloss.npz, produced by
    oracle/gen_aux_golden.py from deepmimo/generator/dataset.py:541-566), coherent and incoherent."""
    import deepmimo_amd as dm
    z = _aux("aux_pathloss.npz")
    n, L = z["ray_power"].shape
    rays = dict(power=z["ray_power"], phase=z["ray_phase"])
    for k in dm.consts.RAY_FIELDS:
        rays.setdefault(k, np.zeros((n, L), np.float32))
    ds = dm.Dataset(rays)
    for coherent, key in ((True, "ref_coherent"), (False, "ref_incoherent")):
        want = z[key]
        got = ds.compute_pathloss(coherent)
        assert np.array_equal(np.isnan(got), np.isnan(want)) and np.isnan(want).sum() > 0
        np.testing.assert_allclose(got[~np.isnan(want)], want[~np.isnan(want)], rtol=0, atol=2e-4)   # dB


def test_sionna_adapter_reference_golden():
    """(a, tau) of this package's adapter on GPU-generated TD channels against the samples the REFERENCE's
    DeepMIMOSionnaAdapter (integrations/sionna_adapter.py:22-200) produced from the reference's own TD channels of
    the same rays (tests/golden/aux_sionna.npz)."""
    import deepmimo_amd as dm
    from deepmimo_amd.sionna_adapter import DeepMIMOSionnaAdapter
    z = _aux("aux_sionna.npz")
    rays = [{k[len(f"bs{b}_ray_"):]: z[k] for k in z.files if k.startswith(f"bs{b}_ray_")} for b in range(2)]
    p = dm.ChannelGenParameters()
    p.bs_antenna.shape, p.ue_antenna.shape = np.array([4, 2]), np.array([2, 1])
    p.bs_antenna.rotation = np.array([0, 10, -20])
    p.num_paths, p.freq_domain = 5, 0
    md = dm.MacroDataset([dm.Dataset(dict(r)) for r in rays])
    H = md.compute_channels(p)
    for b in range(2):
        assert_channel_close(H[b], z[f"td_channel_bs{b}"], what=f"TD channel of BS {b}")
    ad = DeepMIMOSionnaAdapter(md, bs_idx=z["bs_idx"], ue_idx=z["ue_idx"])
    assert len(ad) == int(z["n_samples"])
    samples = list(ad())
    a, tau = np.stack([s[0] for s in samples]), np.stack([s[1] for s in samples])
    assert a.shape == z["a"].shape and a.dtype == z["a"].dtype and tau.dtype == z["tau"].dtype
    np.testing.assert_array_equal(tau, z["tau"])
    peak = np.abs(z["a"]).max()
    assert np.abs(a - z["a"]).max() <= TOL_REL * peak
    ad1 = DeepMIMOSionnaAdapter(md)                                           # defaults: BS 0, every user
    assert len(ad1) == int(z["n_samples_default"])
    s1 = list(ad1())
    np.testing.assert_array_equal(np.stack([s[1] for s in s1]), z["tau_default"])
    assert np.abs(np.stack([s[0] for s in s1]) - z["a_default"]).max() <= TOL_REL * np.abs(z["a_default"]).max()


def test_only_radiation_pattern_changes_reference_golden():
    """The reference keeps its cached `_power_linear_ant_gain` when only a radiation pattern changes (its cache is
    dropped by rotation / FoV changes only, dataset.py:213-220): on the same Dataset the 'dipole' call returns the
    ISOTROPIC channel again (tests/golden/aux_stale_cache.npz records that, and the fresh dipole result).  This
    package recomputes by default - a documented deviation, asserted here - and reproduces the reference's sequence
    exactly under config('strict_reference_cache', True)."""
    import deepmimo_amd as dm
    z = _aux("aux_stale_cache.npz")
    rays = {k[4:]: z[k] for k in z.files if k.startswith("ray_")}
    assert np.array_equal(z["ref_iso"], z["ref_dipole_same_dataset"])         # what the reference does
    assert not np.array_equal(z["ref_dipole_fresh"], z["ref_iso"])

    def params(pattern):
        p = dm.ChannelGenParameters()
        p.bs_antenna.shape, p.ue_antenna.shape = np.array([4, 2]), np.array([2, 1])
        p.bs_antenna.rotation = np.array([10, 20, 30])
        p.bs_antenna.radiation_pattern = pattern
        p.ofdm.subcarriers = 64
        p.ofdm.selected_subcarriers = np.arange(0, 64, 4)
        return p

    # default: the new pattern takes effect at once (deviation)
    ds = dm.Dataset({k: v.copy() for k, v in rays.items()})
    assert_channel_close(ds.compute_channels(params("isotropic")), z["ref_iso"], what="isotropic")
    assert_channel_close(ds.compute_channels(params("halfwave-dipole")), z["ref_dipole_fresh"], what="dipole, recomputed")
    # strict: the reference's sequence, stale value included
    dm.config("strict_reference_cache", True)
    try:
        ds = dm.Dataset({k: v.copy() for k, v in rays.items()})
        assert_channel_close(ds.compute_channels(params("isotropic")), z["ref_iso"], what="strict: isotropic")
        assert_channel_close(ds.compute_channels(params("halfwave-dipole")), z["ref_dipole_same_dataset"], what="strict: stale")
        np.testing.assert_allclose(ds["_power_linear_ant_gain"], z["ref_gain_same_dataset"], rtol=2e-6, equal_nan=True)
        ds.apply_fov()
        assert_channel_close(ds.compute_channels(params("halfwave-dipole")), z["ref_dipole_after_apply_fov"], what="strict: after apply_fov")
    finally:
        dm.config("strict_reference_cache", False)


@pytest.mark.parametrize("mode", ["fd", "td", "lpf", "beams"])
def test_numpy_output_pipeline_equals_resident_tensor(mode):
    """The default NumPy return value comes through the chunked generate -> PCIe -> host-copy pipeline
    (ChannelEngine.channels_to_host); it must hold exactly the bits of the HBM-resident tensor, for every stage-2
    entry point, with a ragged last chunk and more chunks than pipeline stages."""
    import deepmimo_amd as dm
    from deepmimo_amd.dataset import _engine
    from oracle.oracle_np import synth_rays
    rays = synth_rays(1037, 9, seed=4242)
    p = dm.ChannelGenParameters()
    p.bs_antenna.shape, p.ue_antenna.shape = np.array([4, 2]), np.array([2, 1])
    p.ofdm.subcarriers = 64
    p.ofdm.selected_subcarriers = np.arange(0, 64, 2)
    if mode == "td":
        p.freq_domain = 0
    if mode == "lpf":
        p.ofdm.rx_filter = 1
    F = np.stack([dm.steering_vec(np.array([4, 2]), phi=a).ravel() for a in (-40.0, 0.0, 25.0)]) if mode == "beams" else None
    eng = _engine()
    old = eng.HOST_CHUNK_BYTES
    try:
        dm.config("channel_output", "torch")
        ds = dm.Dataset({k: v.copy() for k, v in rays.items()})
        want = (ds.compute_beam_channels(F, p) if mode == "beams" else ds.compute_channels(p)).cpu().numpy()
        dm.config("channel_output", "numpy")
        eng.HOST_CHUNK_BYTES = 100 * int(np.prod(want.shape[1:])) * 8          # 100 users per chunk: 11 chunks, 37 in the last
        ds2 = dm.Dataset({k: v.copy() for k, v in rays.items()})
        got = ds2.compute_beam_channels(F, p) if mode == "beams" else ds2.compute_channels(p)
        assert isinstance(got, np.ndarray) and got.dtype == np.complex64 and got.shape == want.shape
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
        got2 = ds2.compute_beam_channels(F, p) if mode == "beams" else ds2.compute_channels(p)   # staging buffers reused
        assert np.array_equal(got2.view(np.uint32), want.view(np.uint32))
        if mode != "beams":                                   # iter_channels: the same pipeline inside every yielded chunk
            seen = 0
            for b, chunk in ds2.iter_channels(p, chunk_users=450):
                assert np.array_equal(chunk.view(np.uint32), want[b:b + len(chunk)].view(np.uint32))
                seen += len(chunk)
            assert seen == len(want)
    finally:
        eng.HOST_CHUNK_BYTES = old
        dm.config.reset()


@pytest.mark.parametrize("kind", ["fold_16x256", "fold_shared_64x256", "mfma_256x512", "mfma_256x512_v3", "mfma_256x512_v4",
                                  "mfma_256x512_v5", "mfma_256x512_v8", "mfma_256x512_v10", "mfma_256x512_v11", "mfma_rt_64x256",
                                  "mfma_rt_16x512", "beam_power", "beam_power_1rx", "rx_filter"])
def test_launches_are_bit_reproducible(kind):
    """The same launch three times gives the same bits, on enough users that a one-in-a-thousand hazard shows.  A
    round-2 build of the folded kernel (K-steps guarded at run time, the accumulator first read in another basic block
    behind `s_waitcnt; ds_read; s_nop 8`) returned one corrupted 16-subcarrier block - the accumulator registers the
    MFMA writes last - for ~0.5 % of the users, differently on every launch, while every parity test on a few hundred
    users stayed green; the full-size shard-invariance check caught it, this is the direct test."""
    import deepmimo_amd as dm
    from deepmimo_amd.engine import ChannelEngine
    from oracle import oracle_np as onp
    cfg = {"fold_16x256": (150_000, [4, 4], [1, 1], 256), "fold_shared_64x256": (60_000, [8, 8], [1, 1], 256),
           "mfma_256x512": (30_000, [8, 8], [2, 2], 512), "beam_power": (30_000, [8, 8], [2, 2], 512),
           "mfma_256x512_v3": (6_000, [8, 8], [2, 2], 512), "mfma_256x512_v4": (6_000, [8, 8], [2, 2], 512),
           "mfma_256x512_v5": (6_000, [8, 8], [2, 2], 512), "mfma_256x512_v8": (6_000, [8, 8], [2, 2], 512),
           "mfma_256x512_v10": (6_000, [8, 8], [2, 2], 512), "mfma_256x512_v11": (6_000, [8, 8], [2, 2], 512),
           "mfma_rt_64x256": (20_000, [8, 8], [1, 1], 256), "mfma_rt_16x512": (40_000, [4, 4], [1, 1], 512),
           "beam_power_1rx": (20_000, [8, 8], [1, 1], 256),
           "rx_filter": (10_000, [8, 8], [2, 2], 512)}[kind]
    n, bs, ue, N = cfg
    rays = onp.synth_rays(n, 25, seed=77)
    p = dm.ChannelGenParameters()
    p.bs_antenna.shape, p.ue_antenna.shape = np.array(bs), np.array(ue)
    p.ofdm.subcarriers = N
    p.ofdm.selected_subcarriers = np.arange(N)
    if kind == "rx_filter":
        p.ofdm.rx_filter = 1
    eng = ChannelEngine(0)
    prep = eng.prepare(eng.upload_rays(rays), p, want_side=False)
    if kind.startswith("beam_power"):
        F = np.stack([dm.steering_vec(np.array(bs), phi=a).ravel() for a in np.linspace(-60, 60, 64)])
        runs = [eng.beam_power(prep, F)[0].clone() for _ in range(3)]
        for r in runs[1:]:
            assert torch.equal(r, runs[0])
        return
    variant = 2 if kind.startswith("mfma") else 0                  # mfma_rt_*: the run-time-guarded tile body of small row blocks
    if "_v" in kind:
        variant = int(kind.rsplit("_v", 1)[1])                     # the other forms of the matrix-core kernel the ABI exposes
    first = eng.channels(prep, variant=variant).clone()
    for _ in range(2):
        again = eng.channels(prep, variant=variant)
        bad = (torch.view_as_real(again) != torch.view_as_real(first)).reshape(n, -1).any(dim=1)
        if bool(bad.any()):
            from tests._repro_dump import dump_mismatch
            where = dump_mismatch(f"parity_{kind}", p, rays, first, again, bad, variant=variant)
            raise AssertionError(f"{int(bad.sum())} of {n} users differ between two identical launches ({kind}); "
                                 f"differing tiles saved to {where}")


def test_adaptive_precision_weak_tail_worst_case():
    """The matrix-core kernel takes the products of a user's LAST 8-path K-step as one f16 term when every path in it
    is >= 66.2 dB (2^-11 in amplitude) below the user's strongest path (k2_channel_fd_mfma.hip, stage_item).  Worst
    case for that rule: 8 equal strong paths and 8 tail paths sitting just under the threshold (and, other users, just
    over it: the rule must not fire).  Error against the float64 oracle stays within 1e-5 of each user's peak (stated
    tolerance 5e-5); the default (three terms for every path, `adaptive_precision` off) must give different bits exactly
    for the users the rule fires on, which is what shows that it fired - and only when asked to."""
    import deepmimo_amd as dm
    from oracle import oracle_np as onp
    n, L = 64, 16
    rays = onp.synth_rays(n, L, seed=99, all_valid=True)
    rng = np.random.default_rng(3)
    p = rays["power"].astype(np.float64)
    fires = np.zeros(n, bool)
    for u in range(n):
        p[u, :8] = -70.0 + rng.uniform(-0.5, 0.0, 8)                  # strong group, strongest is <= -70 dBW
        if u % 2 == 0:
            p[u, 8:] = p[u, :8].max() - 66.5 - rng.uniform(0, 0.3, 8)  # just under the threshold: the rule fires
            fires[u] = True
        else:
            p[u, 8:] = p[u, :8].max() - 65.0 + rng.uniform(0, 0.5, 8)  # just over: three terms
        order = rng.permutation(L)                                     # stage 1 has to find the order itself
        for k in onp.RAY_KEYS:
            if rays[k].ndim == 2 and rays[k].shape[1] == L:
                rays[k][u] = rays[k][u, order]
        p[u] = p[u, order]
    rays["power"] = p.astype(np.float32)
    case = dict(bs_shape=[8, 8], ue_shape=[2, 2], bs_spacing=0.5, ue_spacing=0.5, bs_rot=[0, 0, 0], bs_pattern="isotropic",
                ue_pattern="isotropic", num_paths=L, freq_domain=1, subcarriers=256, selected=list(range(256)),
                bandwidth=10e6, rx_filter=0, bs_fov=None, ue_fov=None)
    ue_rot = np.array([0, 0, 0])
    ref = onp.compute_channels(rays, oracle_params(case, ue_rot))
    dm.config("fd_kernel_variant", 2)
    try:
        H3 = dm.Dataset(dict(rays)).compute_channels(_dm_params(case, ue_rot))
        dm.config("adaptive_precision", True)
        H = dm.Dataset(dict(rays)).compute_channels(_dm_params(case, ue_rot))
    finally:
        dm.config("adaptive_precision", False)
        dm.config("fd_kernel_variant", 0)
    peak = np.abs(ref["channel"]).reshape(n, -1).max(axis=1)
    err = np.abs(H - ref["channel"]).reshape(n, -1).max(axis=1) / peak
    err3 = np.abs(H3 - ref["channel"]).reshape(n, -1).max(axis=1) / peak
    assert err.max() < 1e-5 and err3.max() < 3e-6, (err.max(), err3.max())
    # With the flag stage 1 also orders the kept paths by amplitude, so on these shuffled rays the two results differ in
    # summation order for every user.  Where the rule fired shows on rays that are ALREADY in that order (strongest path
    # first: stage 1's ranking is then the identity): the bits differ exactly for the users the rule fires on.
    srt = {k: v.copy() for k, v in rays.items()}
    for u in range(n):
        order = np.argsort(-srt["power"][u].astype(np.float64), kind="stable")
        for k in onp.RAY_KEYS:
            if srt[k].ndim == 2 and srt[k].shape[1] == L:
                srt[k][u] = srt[k][u, order]
    dm.config("fd_kernel_variant", 2)
    try:
        S3 = dm.Dataset(dict(srt)).compute_channels(_dm_params(case, ue_rot))
        dm.config("adaptive_precision", True)
        S1 = dm.Dataset(dict(srt)).compute_channels(_dm_params(case, ue_rot))
    finally:
        dm.config("adaptive_precision", False)
        dm.config("fd_kernel_variant", 0)
    same = np.array([np.array_equal(S1[u].view(np.uint32), S3[u].view(np.uint32)) for u in range(n)])
    assert not same[fires].any() and same[~fires].all(), (same[fires].sum(), (~same[~fires]).sum())


def test_subset_and_active_users_reference_golden():
    """`dataset.subset(dataset.get_uniform_idxs(...))` then `compute_channels`, and `get_active_idxs`, against the REAL
    reference on the same rays (tests/golden/aux_helpers.npz from oracle/gen_helpers_golden.py; dataset.py:739-795):
    the public entries that travel, their user-axis indexing, and the channels / LoS / pathloss of the subset."""
    import deepmimo_amd as dm
    z = _aux("aux_helpers.npz")
    rays = {k[4:]: z[k] for k in z.files if k.startswith("ray_")}
    shared = {dm.consts.SCENE_PARAM_NAME: "scene-object", dm.consts.MATERIALS_PARAM_NAME: "materials-object",
              dm.consts.LOAD_PARAMS_PARAM_NAME: {"max_paths": 7}, dm.consts.RT_PARAMS_PARAM_NAME: {"frequency": 3.5e9}}
    ds = dm.Dataset({**{k: v.copy() for k, v in rays.items()}, **shared})
    assert np.array_equal(ds.get_active_idxs(), z["active"])
    _ = (ds.grid_size, ds.num_interactions, ds.inter_int, ds.inter_str, ds.distance, ds.los, ds.pathloss)   # as the generator did
    idxs = ds.get_uniform_idxs([3, 2])
    assert np.array_equal(idxs, z["subset_idxs"])
    sub = ds.subset(idxs)
    keys = sorted(k for k in sub.keys() if not k.startswith("_") and k not in ("channel", "ch_params"))
    assert keys == [str(k) for k in z["subset_keys"]]
    assert sub.n_ue == int(z["subset_n_ue"]) and sub.scene == "scene-object" and sub.rt_params is ds.rt_params
    assert np.array_equal(sub.los, z["subset_los"]) and np.array_equal(sub.rx_pos, z["subset_rx_pos"])
    want = z["subset_pathloss"]
    assert np.array_equal(np.isnan(sub.pathloss), np.isnan(want))
    np.testing.assert_allclose(sub.pathloss[~np.isnan(want)], want[~np.isnan(want)], rtol=0, atol=2e-4)      # dB
    p = dm.ChannelGenParameters()
    p.bs_antenna.shape = np.array([4, 2])
    p.ofdm.subcarriers = 64
    p.ofdm.selected_subcarriers = np.arange(0, 64, 8)
    assert_channel_close(sub.compute_channels(p), z["subset_channel"], what="subset channels vs reference")
    assert np.array_equal(sub.num_paths, ds.num_paths[idxs])
    # device-resident rays stay on the device through subset()
    import torch
    dev = torch.device("cuda", 0)
    dds = dm.Dataset({k: (torch.from_numpy(v).to(dev) if k in dm.consts.RAY_FIELDS else v.copy()) for k, v in rays.items()})
    dsub = dds.subset(idxs)
    assert dsub.power.is_cuda and dsub.power.shape[0] == len(idxs)
    assert_channel_close(dsub.compute_channels(p), z["subset_channel"], what="device-resident subset vs reference")


def test_steering_codebook_reference_golden():
    """dm.steering_vec against the reference's vectors (tests/golden/aux_steering.npz, geometry.py:322-339), and a
    codebook of those REFERENCE vectors through the fused beam-space kernel against F @ H of the oracle."""
    import deepmimo_amd as dm
    from oracle import oracle_np as onp
    z = _aux("aux_steering.npz")
    rows = {}
    for i, (mh, mv, phi, theta, spacing) in enumerate(z["cases"]):
        v = dm.steering_vec(np.array([int(mh), int(mv)]), phi=phi, theta=theta, spacing=spacing)
        assert v.shape == z[f"v{i}"].shape
        assert np.abs(v - z[f"v{i}"]).max() <= 1e-12
        rows.setdefault((int(mh), int(mv)), []).append(z[f"v{i}"].squeeze())
    F = np.array(rows[(8, 1)])                                               # three reference beams of the 8x1 default array
    rays = onp.synth_rays(50, 12, seed=77)
    p = dm.ChannelGenParameters()
    p.ofdm.selected_subcarriers = np.arange(0, 512, 16)
    Href = onp.compute_channels(rays, onp.make_params(ofdm=dict(selected_subcarriers=np.arange(0, 512, 16))))["channel"]
    Y = dm.Dataset(dict(rays)).compute_beam_channels(F, p)
    assert_channel_close(Y, (F @ Href.astype(np.complex128)).astype(np.complex64), what="reference steering codebook")


def test_iter_channels_chunks_equal_full_tensor():
    import deepmimo_amd as dm
    from oracle import oracle_np as onp
    rays = onp.synth_rays(205, 9, seed=17)
    p = dm.ChannelGenParameters()
    p.bs_antenna.shape = np.array([8, 4])
    p.ofdm.selected_subcarriers = np.arange(0, 512, 11)
    ds = dm.Dataset(dict(rays))
    H = ds.compute_channels(p)
    seen = 0
    for b, chunk in ds.iter_channels(p, chunk_users=64):
        assert b == seen and np.array_equal(chunk, H[b:b + chunk.shape[0]])
        seen += chunk.shape[0]
    assert seen == 205


@pytest.mark.parametrize("L,bs,ue,lpf", [(40, [4, 2], [1, 1], 0), (70, [8, 8], [2, 1], 0), (45, [8, 4], [1, 2], 1)])
def test_more_than_32_paths(L, bs, ue, lpf):
    """Beyond DeepMIMO's MAX_PATHS = 25: path slots past 32 are added by accumulate passes (k2_channel_fd.hip),
    on top of either kernel and of the rx_filter table."""
    import deepmimo_amd as dm
    from oracle import oracle_np as onp
    rays = onp.synth_rays(30, L, seed=L)
    case = dict(bs_shape=bs, ue_shape=ue, bs_spacing=0.5, ue_spacing=0.5, bs_rot=[0, 0, 30], bs_pattern="isotropic",
                ue_pattern="isotropic", num_paths=L, freq_domain=1, subcarriers=64, selected=list(range(0, 64, 3)),
                bandwidth=10e6, rx_filter=lpf, bs_fov=None, ue_fov=None)
    ue_rot = np.array([0, 0, 0])
    ref = onp.compute_channels(rays, oracle_params(case, ue_rot))
    ds = dm.Dataset(dict(rays))
    H = ds.compute_channels(_dm_params(case, ue_rot))
    assert_channel_close(H, ref["channel"], what=f"{L} paths")
    np.testing.assert_array_equal(ds.num_paths, ref["num_paths"])
    case["freq_domain"] = 0
    ref = onp.compute_channels(rays, oracle_params(case, ue_rot))
    assert_channel_close(ds.compute_channels(_dm_params(case, ue_rot)), ref["channel"], what=f"{L} paths, time domain")


def test_cache_plumbing_sequence_matches_reference():
    """The same SEQUENCE of Dataset API calls that oracle/gen_sequence_golden.py ran on the real reference
    (lazy attribute before compute with a random UE-rotation range, rotation change, repeated compute, FoV reset
    with lazy .channel, switch to time domain): every step must reproduce what the reference returned
    (dataset.py:144-222, 250, 327-338, 358-378, 515-535)."""
    import io
    from contextlib import redirect_stdout
    import deepmimo_amd as dm
    from oracle.gen_sequence_golden import run_sequence
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "seq_cache_plumbing.npz"), allow_pickle=False)
    rays = {k[4:]: z[k] for k in z.files if k.startswith("ray_")}
    with redirect_stdout(io.StringIO()):
        got = run_sequence(dm, rays)
    for k in got:
        ref = z["ref_" + k]
        if k.endswith("channel"):
            assert_channel_close(got[k], ref, what=k)
        else:
            np.testing.assert_array_equal(got[k], ref, err_msg=k)
    # the three frequency-domain steps really differ from each other (the sequence is not vacuous)
    assert not np.allclose(z["ref_s1_channel"], z["ref_s2_channel"])
    assert np.array_equal(z["ref_s2_channel"], z["ref_s3_channel"])


@pytest.mark.parametrize("bs,ue,sel", [([8, 4], [2, 1], np.arange(0, 512, 8)),      # 64 rows: 18 KB of LDS
                                       ([8, 8], [2, 2], np.arange(512)),            # headline shape: 256 rows, 74 KB of LDS ->
                                                                                    # hipFuncSetAttribute + occupancy query run inside the capture
                                       ([8, 1], [1, 1], np.arange(512))])           # default arrays: the folded kernel
def test_hip_graph_replay(bs, ue, sel):
    """The C-ABI calls only enqueue kernels (no allocation, no sync): stage 1 + stage 2 captured in a HIP graph and
    replayed on new ray data give the same tensor as fresh eager calls."""
    import torch
    import deepmimo_amd as dm
    from deepmimo_amd.engine import ChannelEngine
    from oracle import oracle_np as onp
    eng = ChannelEngine(0)
    p = dm.ChannelGenParameters()
    p.bs_antenna.shape, p.ue_antenna.shape = np.array(bs), np.array(ue)
    p.ofdm.selected_subcarriers = sel
    p.validate(64)
    a, b = onp.synth_rays(64, 10, seed=1), onp.synth_rays(64, 10, seed=2)
    rays = eng.upload_rays(a)
    prep = eng.prepare(rays, p, want_side=True)
    out = torch.empty(eng.channel_shape(prep), dtype=torch.complex64, device="cuda")
    eng.relaunch(prep, out)                                   # warm-up outside the capture
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        eng.relaunch(prep, out)
    want_a = eng.channels(eng.prepare(eng.upload_rays(a), p, want_side=False)).clone()
    want_b = eng.channels(eng.prepare(eng.upload_rays(b), p, want_side=False)).clone()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(torch.view_as_real(out), torch.view_as_real(want_a))
    for k in dm.consts.RAY_FIELDS:                            # new batch lands in the SAME device buffers
        rays.fields[k].copy_(torch.from_numpy(b[k]))
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(torch.view_as_real(out), torch.view_as_real(want_b))
    ref_b = onp.compute_channels(b, onp.make_params(bs_antenna=dict(shape=bs), ue_antenna=dict(shape=ue),
                                                   ofdm=dict(selected_subcarriers=sel)))
    assert np.array_equal(prep.side["los"].cpu().numpy(), ref_b["los"])
    assert_channel_close(out.cpu().numpy(), ref_b["channel"], what="graph replay vs oracle")


def test_reference_patch_gpu_path():
    """The body reference_patch installs into the reference's Dataset (the real package cannot travel to the GPU
    box): run it here on a Dataset-shaped object and compare with the goldens of the real reference."""
    import deepmimo_amd as dm
    from deepmimo_amd import reference_patch as rp
    case, rays, ue_rot, ref = load_golden("g03_rot_fov")
    ds = _dataset(case, rays)
    p = _dm_params(case, ue_rot)
    ds.set_channel_params(p)
    np.random.seed(1001)
    H = rp._gpu_compute_channels(ds, p, 0)
    assert_channel_close(H, ref["channel"], what="reference_patch body")
    np.testing.assert_array_equal(ds["los"], ref["los"])
    np.testing.assert_array_equal(ds["_fov_mask"], ref["fov_mask"])
    case, rays, ue_rot, ref = load_golden("g09_random_ue_rot")       # random UE-rotation range: same RNG order
    ds = _dataset(case, rays)
    p = _dm_params(case, ue_rot)
    ds.set_channel_params(p)
    np.random.seed(1001)
    assert_channel_close(rp._gpu_compute_channels(ds, p, 0), ref["channel"], what="reference_patch random rotation")


def test_c_abi_demo_client_matches_python_host():
    """examples/c_abi_demo.cpp: a C++ program with hipMalloc'ed buffers and plain structs - no Python, no PyTorch -
    calling the C-ABI.  Its inputs come from a fixed LCG; the same inputs through the Python host must give the same
    tensor fingerprints, LoS and path-count sums."""
    import json
    import subprocess
    import deepmimo_amd as dm
    from deepmimo_amd._native import LIB_PATH
    exe = os.path.join(os.path.dirname(LIB_PATH), "dmx_demo")
    if not os.path.exists(exe):
        pytest.skip("demo client not built (make -C deepmimo_amd/csrc)")
    n, L, K = 2000, 10, 64
    r = subprocess.run([exe, str(n)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    got = json.loads(r.stdout.strip().splitlines()[-1])
    assert got["abi"] == 3 and got["shape"] == [n, 2, 16, K]
    # the demo's LCG, vectorised: s_{i+1} = a s_i + c (mod 2^32)
    total = 8 * n * L
    state = np.empty(total, dtype=np.uint64)
    s_ = 12345
    for i in range(total):
        s_ = (s_ * 1664525 + 1013904223) & 0xFFFFFFFF
        state[i] = s_
    x = (state >> np.uint64(8)).astype(np.float32)
    lo = np.array([-140, -180, 1e-8, -180, 0, -180, 0, 0], dtype=np.float32)
    hi = np.array([-60, 180, 2e-6, 180, 180, 180, 180, 4.999], dtype=np.float32)
    keys = ("power", "phase", "delay", "aoa_az", "aoa_el", "aod_az", "aod_el", "inter")
    rays = {}
    for f, k in enumerate(keys):
        t = ((hi[f] - lo[f]) * x[f * n * L:(f + 1) * n * L]).astype(np.float32) * np.float32(1.0 / 16777216.0)
        rays[k] = (lo[f] + t.astype(np.float32)).astype(np.float32).reshape(n, L)
    rays["inter"] = np.floor(rays["inter"])
    for k in keys:
        rays[k][::7, 3:] = np.nan
    rays["rx_pos"] = np.zeros((n, 3), np.float32)
    rays["tx_pos"] = np.zeros((1, 3), np.float32)
    p = dm.ChannelGenParameters()
    p.bs_antenna.shape, p.ue_antenna.shape = np.array([8, 2]), np.array([2, 1])
    p.bs_antenna.rotation = np.array([0, 0, 30])
    p.ofdm.selected_subcarriers = np.arange(K)
    ds = dm.Dataset(rays)
    H = ds.compute_channels(p)
    h = np.ascontiguousarray(H).view(np.float32).astype(np.float64).ravel()
    energy, wsum = float(np.sum(h * h)), float(np.sum(h * np.cos(0.37 * np.arange(h.size, dtype=np.float64))))
    assert got["energy"] == pytest.approx(energy, rel=1e-9)
    assert got["wsum"] == pytest.approx(wsum, rel=1e-6, abs=1e-9 * np.sqrt(energy))
    assert got["los_sum"] == int(ds.los.sum()) and got["num_paths_sum"] == int(ds.num_paths.sum())


@pytest.mark.parametrize("variant", [0, 1, 4, 5, 9, 12])
@pytest.mark.parametrize("bs,ue,K", [([9, 5], [1, 1], 24), ([8, 6], [1, 1], 40), ([3, 3], [1, 1], 33), ([8, 1], [1, 1], 100),
                                    ([4, 3], [1, 1], 7)])
def test_output_guard_regions_stay_untouched(bs, ue, K, variant):
    """Stores of partial tiles are dropped by the buffer range check (row offsets past the block, masked lanes of a
    partial subcarrier block): nothing may land outside the caller's tensor.  The output sits between two
    sentinel-filled guard regions of one allocation; M = 45 / 48 / 9 / 8 / 12 rows, K not a multiple of 16."""
    import torch
    import deepmimo_amd as dm
    from deepmimo_amd.engine import ChannelEngine
    from oracle import oracle_np as onp
    n, L = 37, 11
    M = bs[0] * bs[1] * ue[0] * ue[1]
    if variant == 9 and (bs[0] * bs[1] + ue[0] * ue[1] + K) * L * 8 > 156 * 1024:
        pytest.skip("tables exceed the LDS of the small-output kernel")
    rays = onp.synth_rays(n, L, seed=400 + K)
    p = dm.ChannelGenParameters()
    p.bs_antenna.shape, p.ue_antenna.shape = np.array(bs), np.array(ue)
    p.num_paths = L
    p.ofdm.selected_subcarriers = np.arange(3, 3 + K)
    p.validate(n)
    eng = ChannelEngine(0)
    prep = eng.prepare(eng.upload_rays(rays), p, want_side=False)
    guard, size = 1 << 16, n * M * K
    sentinel = complex(-12345.5, 54321.25)
    big = torch.full((guard + size + guard,), sentinel, dtype=torch.complex64, device="cuda")
    out = big[guard:guard + size].view(n, ue[0] * ue[1], bs[0] * bs[1], K)
    eng.channels(prep, out=out, variant=variant)
    torch.cuda.synchronize()
    assert bool((big[:guard] == sentinel).all()) and bool((big[guard + size:] == sentinel).all()), "write outside the output tensor"
    ref = onp.compute_channels(rays, onp.make_params(bs_antenna=dict(shape=bs), ue_antenna=dict(shape=ue), num_paths=L,
                                                     ofdm=dict(selected_subcarriers=np.arange(3, 3 + K))))
    assert_channel_close(out.cpu().numpy(), ref["channel"], what=f"guarded output, variant {variant}")
    # the same for a user sub-range written into the middle of the tensor: its neighbours stay as they are
    big.fill_(sentinel)
    eng.channels(prep, out=out[5:20], user_begin=5, user_count=15, variant=variant)
    torch.cuda.synchronize()
    assert bool((out[:5] == sentinel).all()) and bool((out[20:] == sentinel).all())
    assert_channel_close(out[5:20].cpu().numpy(), ref["channel"][5:20], what="sub-range")


@pytest.mark.parametrize("L,per_user", [(40, True), (25, False), (7, True)])
def test_prepare_without_side_products_matches_oracle(L, per_user):
    """eng.prepare(..., want_side=False) - what bench.py times - takes stage 1's branch that skips arccos / atan2
    (array-response steps straight from the rotated direction).  Against the oracle with rotated BS and UE, NaN-padded
    rows with holes, per-user rotation and more than 32 paths."""
    import deepmimo_amd as dm
    from deepmimo_amd.engine import ChannelEngine
    from oracle import oracle_np as onp
    n = 53
    rays = onp.synth_rays(n, L, seed=900 + L)
    rng = np.random.default_rng(L)
    hole = rng.uniform(size=(n, L)) < 0.15
    for k in onp.RAY_KEYS:
        rays[k][hole] = np.nan
    ue_rot = rng.uniform(-70, 70, (n, 3)) if per_user else np.array([25, -40, 110])
    bs_rot = np.array([-15, 35, 200])
    p = dm.ChannelGenParameters()
    p.bs_antenna.shape, p.ue_antenna.shape = np.array([6, 3]), np.array([2, 2])
    p.bs_antenna.rotation, p.ue_antenna.rotation = bs_rot, ue_rot
    p.num_paths = L
    p.ofdm.selected_subcarriers = np.arange(0, 512, 9)
    p.validate(n)
    op = onp.make_params(bs_antenna=dict(shape=[6, 3], rotation=bs_rot), ue_antenna=dict(shape=[2, 2], rotation=ue_rot),
                         num_paths=L, ofdm=dict(selected_subcarriers=np.arange(0, 512, 9)))
    ref = onp.compute_channels(rays, op)
    eng = ChannelEngine(0)
    dr = eng.upload_rays(rays)
    lean = eng.channels(eng.prepare(dr, p, ue_rotation_per_user=ue_rot if per_user else None, want_side=False))
    full = eng.channels(eng.prepare(dr, p, ue_rotation_per_user=ue_rot if per_user else None, want_side=True))
    assert_channel_close(lean.cpu().numpy(), ref["channel"], what="want_side=False")
    assert_channel_close(full.cpu().numpy(), ref["channel"], what="want_side=True")


@pytest.mark.parametrize("variant", [0, 1, 2, 12])
@pytest.mark.parametrize("bs,ue,K", [([8, 8], [2, 2], 128), ([8, 1], [1, 1], 128)])
def test_extreme_power_spread(bs, ue, K, variant):
    """VERDICT r1 weak point 8: the split-precision kernels scale each user's operands by ONE power of two taken from its
    strongest path.  Users whose strongest path is 60 ... 150 dB above the rest, users that are uniformly very weak
    (-190 dBW) or very strong (+20 dBW), and a user whose paths span 170 dB evenly: the error stays relative to the
    user's peak (tolerance 5e-5 of it), nothing overflows or flushes."""
    import deepmimo_amd as dm
    from oracle import oracle_np as onp
    n, L = 48, 25
    if variant == 12 and bs[0] * bs[1] * ue[0] * ue[1] > 128:
        pytest.skip("folded kernel: at most 128 antenna pairs")
    rays = onp.synth_rays(n, L, seed=4242, all_valid=True)
    rng = np.random.default_rng(7)
    p = rays["power"]
    for u in range(n):
        kind = u % 6
        if kind == 0:                                   # one dominant path, the rest 60-150 dB below
            p[u] = rng.uniform(-190, -100, L); p[u, rng.integers(L)] = -40.0
        elif kind == 1:                                 # uniformly very weak
            p[u] = rng.uniform(-195, -185, L)
        elif kind == 2:                                 # uniformly very strong
            p[u] = rng.uniform(10, 20, L)
        elif kind == 3:                                 # an even 170-dB ladder
            p[u] = np.linspace(-20, -190, L)
        elif kind == 4:                                 # two equal dominant paths in near-opposite phase + weak floor
            p[u] = rng.uniform(-180, -150, L); p[u, :2] = -50.0
            rays["phase"][u, 0], rays["phase"][u, 1] = 10.0, -169.5
    rays["power"] = p.astype(np.float32)
    case = dict(bs_shape=bs, ue_shape=ue, bs_spacing=0.5, ue_spacing=0.5, bs_rot=[0, 0, 0], bs_pattern="isotropic",
                ue_pattern="isotropic", num_paths=L, freq_domain=1, subcarriers=512, selected=list(range(0, 512, 512 // K)),
                bandwidth=10e6, rx_filter=0, bs_fov=None, ue_fov=None)
    ue_rot = np.array([0, 0, 0])
    ref = onp.compute_channels(rays, oracle_params(case, ue_rot))
    dm.config("fd_kernel_variant", variant)
    try:
        H = dm.Dataset(dict(rays)).compute_channels(_dm_params(case, ue_rot))
    finally:
        dm.config("fd_kernel_variant", 0)
    assert np.isfinite(H).all()
    worst = assert_channel_close(H, ref["channel"], what=f"power spread, variant {variant}")
    assert worst < 2e-5
    peak = np.abs(ref["channel"]).reshape(n, -1).max(axis=1)
    assert peak.min() > 0 and peak.max() / peak.min() > 1e9          # the users themselves span > 180 dB


def test_codebook_with_huge_dynamic_range():
    """A TX codebook whose rows span 9 orders of magnitude (and one all-zero beam): the projected responses are scaled
    per user (k2b_beam_project's exponent), so each (user, beam) row keeps 5e-5 of the USER's peak over all beams -
    the tolerance the channel itself has; beams more than ~90 dB below a user's strongest one lose relative precision,
    as any fp32 result of the full product would."""
    import deepmimo_amd as dm
    from oracle import oracle_np as onp
    rays = onp.synth_rays(40, 12, seed=99, all_valid=True)
    p = dm.ChannelGenParameters()
    p.bs_antenna.shape, p.ue_antenna.shape = np.array([8, 4]), np.array([2, 1])
    p.num_paths = 12
    p.ofdm.selected_subcarriers = np.arange(0, 512, 8)
    Href = onp.compute_channels(rays, onp.make_params(bs_antenna=dict(shape=[8, 4]), ue_antenna=dict(shape=[2, 1]), num_paths=12,
                                                     ofdm=dict(selected_subcarriers=np.arange(0, 512, 8))))["channel"].astype(np.complex128)
    rng = np.random.default_rng(3)
    nb = 12
    F = (rng.normal(size=(nb, 32)) + 1j * rng.normal(size=(nb, 32))) * (10.0 ** np.linspace(-6, 3, nb))[:, None]
    F[5] = 0
    ds = dm.Dataset(dict(rays))
    Y = ds.compute_beam_channels(F, p)
    Yref = F @ Href
    assert np.isfinite(Y).all() and np.all(Y[:, :, 5] == 0)
    peak = np.abs(Yref).reshape(40, -1).max(axis=1)
    err = np.abs(Y - Yref).reshape(40, -1).max(axis=1)
    assert np.all(err <= TOL_REL * peak)
    strong = np.abs(Yref) >= 1e-3 * peak[:, None, None, None]          # within 60 dB of the user's strongest entry
    assert np.all(np.abs(Y - Yref)[strong] <= 1e-3 * np.abs(Yref)[strong])
    pwr = ds.compute_beam_power(F, p)
    amp = ds["beam_mean_amplitude"]
    want = np.abs(Yref).mean(axis=1).mean(axis=-1)
    assert np.all(np.abs(amp - want) <= 1e-5 * want.max(axis=1, keepdims=True)) and np.all(amp[:, 5] == 0)
    assert np.all(np.isneginf(pwr[:, 5]) | np.isnan(pwr[:, 5]))          # 20 log10(0): the notebook's formula gives -inf


def test_array_response_product_matches_oracle():
    """The public lazy attribute `array_response_product` (dataset.py:849, 398-417) for a bounded user count: built on
    the host from the rotated / FoV-filtered angles of GPU stage 1, against the oracle's restatement of
    `_array_response_batch` (geometry.py:38-82; NaN angle -> zero column)."""
    import deepmimo_amd as dm
    from oracle import oracle_np as onp
    rays = onp.synth_rays(40, 12, seed=5)
    p = dm.ChannelGenParameters()
    p.bs_antenna.shape, p.ue_antenna.shape = np.array([4, 2]), np.array([2, 1])
    p.bs_antenna.rotation = np.array([10, -20, 30])
    ds = dm.Dataset(dict(rays))
    ds.apply_fov(bs_fov=np.array([150, 120]))
    ds.set_channel_params(p)
    got = ds.array_response_product
    op = onp.make_params(bs_antenna=dict(shape=[4, 2], rotation=np.array([10, -20, 30])), ue_antenna=dict(shape=[2, 1]))
    prep = onp.prepare_paths(rays, op, bs_fov=np.array([150, 120]))
    a_tx = onp.array_response_batch(op["bs_antenna"]["shape"], 0.5, prep["_aod_el_rot_fov"], prep["_aod_az_rot_fov"])
    a_rx = onp.array_response_batch(op["ue_antenna"]["shape"], 0.5, prep["_aoa_el_rot_fov"], prep["_aoa_az_rot_fov"])
    want = a_rx[:, :, None, :] * a_tx[:, None, :, :]
    assert got.shape == want.shape == (40, 2, 8, 12) and got.dtype == np.complex128
    assert np.abs(got - want).max() < 1e-9
    assert np.array_equal(got == 0, want == 0)


@pytest.mark.parametrize("adaptive", [False, True])
@pytest.mark.parametrize("kind", ["fold_8x512", "fold_shared_64x128", "mfma_256x256", "beam_power"])
def test_precision_flag_parity(kind, adaptive):
    """dmx_params.flags (ABI 3): both arithmetic modes of the matrix-core kernels against the float64 oracle, on rays whose
    powers span 80 dB (the opt-in one-term rule fires for most users).  Default: <= 3e-6 of a user's peak; with
    DMX_FLAG_ADAPTIVE_TERMS: <= 1e-5 (stated tolerance 5e-5).  The flag is a field of the preparation's parameter block:
    two engines' worth of state in one process, no environment involved."""
    import deepmimo_amd as dm
    from deepmimo_amd.engine import ChannelEngine
    from oracle import oracle_np as onp
    n, bs, ue, N = {"fold_8x512": (300, [8, 1], [1, 1], 512), "fold_shared_64x128": (120, [8, 8], [1, 1], 128),
                    "mfma_256x256": (100, [8, 8], [2, 2], 256), "beam_power": (100, [8, 8], [2, 2], 256)}[kind]
    rays = onp.synth_rays(n, 25, seed=123, all_valid=True)
    p = dm.ChannelGenParameters()
    p.bs_antenna.shape, p.ue_antenna.shape = np.array(bs), np.array(ue)
    p.ofdm.subcarriers = N
    p.ofdm.selected_subcarriers = np.arange(N)
    op = onp.make_params(bs_antenna=dict(shape=bs), ue_antenna=dict(shape=ue), ofdm=dict(subcarriers=N, selected_subcarriers=np.arange(N)))
    ref = onp.compute_channels(rays, op)["channel"].astype(np.complex128)
    eng = ChannelEngine(0)
    prep = eng.prepare(eng.upload_rays(rays), p, want_side="light", adaptive_terms=adaptive)
    assert prep.params_struct.flags == (1 if adaptive else 0)
    lim = 1e-5 if adaptive else 3e-6
    if kind == "beam_power":
        F = np.stack([dm.steering_vec(np.array(bs), phi=a).ravel() for a in np.linspace(-60, 60, 16)])
        amp = eng.beam_power(prep, F)[0].cpu().numpy()
        want = np.abs(np.einsum("bt,nrtk->nrbk", F, ref)).mean(axis=1).mean(axis=-1)
        assert np.all(np.abs(amp - want) <= lim * want.max(axis=1, keepdims=True))
        return
    H = eng.channels(prep).cpu().numpy()
    peak = np.abs(ref).reshape(n, -1).max(axis=1)
    err = np.abs(H - ref).reshape(n, -1).max(axis=1) / peak
    assert err.max() < lim, (kind, adaptive, err.max())
    assert np.array_equal(prep.side["num_paths"].cpu().numpy(), np.full(n, 25))
