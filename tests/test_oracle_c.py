"""CPU tests: the plain-C restatement (oracle/oracle_c.c, the dmx_cpu_* twins of the product's C-ABI) against
the golden vectors of the real reference and against the NumPy oracle on seeded random configurations.

It is a second, independently written restatement (scalar loops + libm).  Where it can differ from NumPy it does
so by rounding only: float64 sin/cos/arccos/arctan2 (SIMD vs glibc, ~1e-16) and `10 ** x` in float32, where
NumPy's vector pow and glibc's powf differ by one ulp (1.2e-7) - hence 1e-6 on channels and 2.5e-7 on powers
instead of the 1e-12 the NumPy oracle meets.  Integer side products and FoV masks are exact."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle_c as oc
from oracle import oracle_np as onp
from tests._cases import (golden_names, load_golden, oracle_params, fov_args, assert_channel_close, random_case)

TOL_H = 1e-6          # of each user's peak
TOL_P = 2.5e-7        # relative, linear powers (one float32 ulp)
TOL_ANG = 1e-14       # radians


def _check_side(res, ref_los, ref_np, ref_mask, ref_pl, ref_pag, ref_angles):
    np.testing.assert_array_equal(res["los"], ref_los)
    np.testing.assert_array_equal(res["num_paths"], ref_np)
    if ref_mask is None:
        assert res["_fov_mask"] is None
    else:
        np.testing.assert_array_equal(res["_fov_mask"], ref_mask)
    np.testing.assert_allclose(res["power_linear"], ref_pl, rtol=TOL_P, atol=0, equal_nan=True)
    np.testing.assert_allclose(res["_power_linear_ant_gain"], ref_pag, rtol=5e-7, atol=1e-30, equal_nan=True)
    for k, want in ref_angles.items():
        np.testing.assert_allclose(res[k], want, rtol=0, atol=TOL_ANG, equal_nan=True)


@pytest.mark.parametrize("name", golden_names())
def test_c_oracle_matches_reference(name):
    case, rays, ue_rot, ref = load_golden(name)
    params = oracle_params(case, ue_rot)
    bs_fov, ue_fov = fov_args(case)
    res = oc.compute_channels(rays, params, bs_fov, ue_fov)
    H = res["channel"]
    if "channel" in ref:
        assert H.dtype == np.complex64
        assert_channel_close(H, ref["channel"], tol_rel=TOL_H, what=name)
    else:
        sub = case["subsample"]
        assert_channel_close(H[:, :, sub["tx"], :][..., sub["k"]], ref["channel_sub"], tol_rel=TOL_H, what=name)
    _check_side(res, ref["los"], ref["num_paths"], ref.get("fov_mask"), ref["power_linear"],
                ref["power_linear_ant_gain"],
                {"_" + k: ref[k] for k in ("aod_el_rot", "aod_az_rot", "aoa_el_rot", "aoa_az_rot")})
    if case["freq_domain"]:
        assert res["delay_exceeds_symbol"] == bool(ref["warned"])


@pytest.mark.parametrize("name", ["g10_doppler_v3", "g13_doppler_lpf_v3"])
def test_c_oracle_doppler_matches_v3(name):
    case, rays, ue_rot, ref = load_golden(name)
    params = oracle_params(case, ue_rot)
    params["enable_doppler"] = 1
    dop = dict(vel=rays["doppler_vel"], acc=rays["doppler_acc"], carrier_freq=3.5e9)
    res = oc.compute_channels(rays, params, doppler=dop)
    assert_channel_close(res["channel"], ref["channel_doppler"], tol_rel=2e-6, what=name)


@pytest.mark.parametrize("seed", range(40))
def test_c_oracle_matches_numpy_oracle_random(seed):
    c, rays, ue_rot, op, bs_fov, ue_fov = random_case(seed)
    a = oc.compute_channels(rays, op, bs_fov=bs_fov, ue_fov=ue_fov, threads=2)
    b = onp.compute_channels(rays, op, bs_fov=bs_fov, ue_fov=ue_fov)
    assert_channel_close(a["channel"], b["channel"], tol_rel=TOL_H, what=f"seed {seed}: {c}")
    _check_side(a, b["los"], b["num_paths"], b["_fov_mask"], b["power_linear"], b["_power_linear_ant_gain"],
                {k: b[k] for k in ("_aod_el_rot", "_aod_az_rot", "_aoa_el_rot", "_aoa_az_rot")})


def test_c_oracle_user_range_and_threads():
    """user_begin / user_count of the twin select the same rows as the full run; threads do not change a bit."""
    rays = onp.synth_rays(37, 9, seed=11)
    op = onp.make_params(bs_antenna=dict(shape=[4, 2], rotation=np.array([5, 10, 20])), ue_antenna=dict(shape=[2, 1]),
                         ofdm=dict(subcarriers=64, selected_subcarriers=np.arange(0, 64, 3)))
    full = oc.compute_channels(rays, op, threads=1)["channel"]
    part = oc.compute_channels(rays, op, users=slice(10, 29), threads=3)["channel"]
    np.testing.assert_array_equal(part, full[10:29])


def test_c_oracle_empty_and_all_nan():
    res = oc.compute_channels(onp.synth_rays(0, 5, seed=1), onp.make_params())
    assert res["channel"].shape == (0, 1, 8, 1)
    rays = onp.synth_rays(3, 4, seed=2)
    for k in onp.RAY_KEYS:
        rays[k][:] = np.nan
    res = oc.compute_channels(rays, onp.make_params())
    assert np.all(res["channel"] == 0)
    assert res["los"].tolist() == [-1, -1, -1] and res["num_paths"].tolist() == [0, 0, 0]


def test_twin_structs_have_the_product_layout():
    """The twins take include/deepmimo_amd.h's structs: the oracle's ctypes mirrors and the product's must agree."""
    from deepmimo_amd import _native as nat
    for mine, theirs in ((oc.Rays, nat.DmxRays), (oc.Params, nat.DmxParams), (oc.Side, nat.DmxSide)):
        assert C.sizeof(mine) == C.sizeof(theirs)
        assert [(n, getattr(mine, n).offset) for n, *_ in mine._fields_] == \
               [(n, getattr(theirs, n).offset) for n, *_ in theirs._fields_]
    lib = oc.load()
    for sym in oc.SYMBOLS:
        getattr(lib, sym)
    assert lib.dmx_cpu_version() == nat.ABI_VERSION
