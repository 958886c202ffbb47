"""CPU tests: the NumPy oracle (oracle/oracle_np.py) against the golden vectors produced by the
real reference (oracle/gen_golden.py).  This is what "pins" the oracle: channels must agree to
1e-12 absolute (the oracle follows the reference's dtype flow), integer side products exactly."""
import os

import numpy as np
import pytest

from oracle import oracle_np as onp
from tests._cases import (golden_names, load_golden, oracle_params, fov_args, assert_channel_close)


def _run(name, style):
    case, rays, ue_rot, ref = load_golden(name)
    params = oracle_params(case, ue_rot)
    bs_fov, ue_fov = fov_args(case)
    res = onp.compute_channels(rays, params, bs_fov, ue_fov, style=style)
    return case, rays, ref, res, params


@pytest.mark.parametrize("name", golden_names())
def test_oracle_matches_reference(name):
    case, rays, ref, res, params = _run(name, "batched")
    H = res["channel"]
    if "channel" in ref:
        assert H.dtype == ref["channel"].dtype == np.complex64
        assert_channel_close(H, ref["channel"], tol_rel=1e-7, tol_abs=1e-12, what=name)
        # near bit-level: the oracle reproduces the dtype flow, only summation order may differ
        fin = np.isfinite(ref["channel"])
        assert np.max(np.abs(np.where(fin, H - ref["channel"], 0))) <= 1e-10
    else:
        sub = case["subsample"]
        Hs = H[:, :, sub["tx"], :][..., sub["k"]]
        assert_channel_close(Hs, ref["channel_sub"], tol_rel=1e-7, what=name)
    np.testing.assert_array_equal(res["los"], ref["los"])
    np.testing.assert_array_equal(res["num_paths"], ref["num_paths"])
    if "fov_mask" in ref:
        np.testing.assert_array_equal(res["_fov_mask"], ref["fov_mask"])
    else:
        assert res["_fov_mask"] is None
    np.testing.assert_array_equal(res["power_linear"], ref["power_linear"])
    np.testing.assert_array_equal(res["_power_linear_ant_gain"], ref["power_linear_ant_gain"])
    for k in ("aod_el_rot", "aod_az_rot", "aoa_el_rot", "aoa_az_rot"):
        np.testing.assert_array_equal(res["_" + k], ref[k])
    if case["freq_domain"]:
        assert res["delay_exceeds_symbol"] == bool(ref["warned"])


@pytest.mark.parametrize("name", ["g01_plumbing", "g03_rot_fov", "g06_dipole", "g08_num_paths_nan"])
def test_reference_style_loop_equals_batched(name):
    """The per-user broadcast+nansum loop (what bench.py times as the CPU baseline) and the
    batched einsum are the same arithmetic."""
    _, _, ref, res_b, _ = _run(name, "batched")
    _, _, _, res_r, _ = _run(name, "reference")
    assert_channel_close(res_r["channel"], ref["channel"], tol_rel=1e-7, what=name)
    assert_channel_close(res_r["channel"], res_b["channel"], tol_rel=1e-7, what=name)


@pytest.mark.parametrize("name", ["g10_doppler_v3", "g13_doppler_lpf_v3"])
def test_doppler_term_matches_v3(name):
    """v4 has no Doppler implementation; the v3 generator's term is the oracle (SURVEY finding 4).
    g13 is the rx_filter branch, where the Doppler phase depends on the tap delay Ts*d."""
    case, rays, ue_rot, ref = load_golden(name)
    params = oracle_params(case, ue_rot)
    params["enable_doppler"] = 1
    dop = dict(vel=rays["doppler_vel"], acc=rays["doppler_acc"], carrier_freq=3.5e9)
    res = onp.compute_channels(rays, params, doppler=dop)
    assert float(ref["v3_v4_maxdiff"]) < 1e-10
    assert_channel_close(res["channel"], ref["channel_doppler"], tol_rel=1e-6, what="doppler")
    # and it is not a no-op
    assert np.max(np.abs(ref["channel_doppler"] - ref["channel"])) > 1e-3 * np.max(np.abs(ref["channel"]))


def test_dipole_known_answers():
    """reference test idea: test/test_ant_patterns.py:72-78 (max at 90 deg, nulls at 0/180)."""
    g = onp.pattern_gain("halfwave-dipole", np.array([np.pi / 2, 0.0, np.pi, np.pi / 4, np.nan]))
    assert g[0] == pytest.approx(1.643)
    assert g[1] == 0 and g[2] == 0 and g[4] == 0
    rel = (g[3] / g[0]) ** 2          # tx*rx relative gain at 45 deg, v4 formula (/sin, not /sin^2)
    assert rel == pytest.approx(0.0777, abs=2e-4)          # SURVEY.md 8(c) G6 / test_ant_patterns.py:72-78
    assert onp.pattern_gain("isotropic", np.zeros(3)) == 1.0
    with pytest.raises(NotImplementedError):
        onp.pattern_gain("patch", np.zeros(3))


def test_ant_indices_order():
    idx = onp.ant_indices([4, 2])
    assert idx[:, 0].tolist() == [0] * 8
    assert idx[:, 1].tolist() == [0, 1, 2, 3, 0, 1, 2, 3]
    assert idx[:, 2].tolist() == [0, 0, 0, 0, 1, 1, 1, 1]


def test_array_response_nan_column_and_batch_vs_scalar():
    """reference test idea: test/test_array_response.py (batch == per-user, NaN edge cases)."""
    rng = np.random.default_rng(5)
    th = rng.uniform(0, np.pi, (6, 7))
    ph = rng.uniform(-np.pi, np.pi, (6, 7))
    th[2, 3:] = np.nan
    ph[2, 3:] = np.nan
    a = onp.array_response_batch([4, 2], 0.5, th, ph)
    assert a.shape == (6, 8, 7)
    assert np.all(a[2, :, 3:] == 0)
    idx = onp.ant_indices([4, 2])
    for i in (0, 2):
        for l in range(3):
            want = np.exp(1j * np.pi * (idx[:, 1] * np.sin(th[i, l]) * np.sin(ph[i, l]) + idx[:, 2] * np.cos(th[i, l])))
            np.testing.assert_allclose(a[i, :, l], want, rtol=1e-12, atol=1e-12)


def test_empty_and_all_nan_inputs():
    rays = onp.synth_rays(0, 5, seed=1)
    res = onp.compute_channels(rays, onp.make_params())
    assert res["channel"].shape == (0, 1, 8, 1)
    rays = onp.synth_rays(3, 4, seed=2)
    for k in onp.RAY_KEYS:
        rays[k][:] = np.nan
    res = onp.compute_channels(rays, onp.make_params())
    assert np.all(res["channel"] == 0)
    assert res["los"].tolist() == [-1, -1, -1]
    assert res["num_paths"].tolist() == [0, 0, 0]


def test_numpy_f32_sincos_model():
    """K1 reproduces NumPy's float32 sin/cos routine (csrc/k1_path_prep.hip: np_sincosf).  This pins the
    model of that routine - same constants, same operation order - to np.sin / np.cos bit for bit, so a
    NumPy whose float32 loops changed would be noticed here rather than as drifting rotated angles."""
    f32 = np.float32

    def fma(a, b, c):      # float32 FMA emulated through float64 (exact product + one rounding)
        return (a.astype(np.float64) * np.float64(b) + np.asarray(c, dtype=np.float64)).astype(f32)

    def H(s):
        return f32(float.fromhex(s))

    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(-np.pi, np.pi, 400_000), rng.uniform(0, np.pi, 100_000),
                        np.deg2rad(np.arange(0, 181, dtype=f32))]).astype(f32)
    q = (x * H("0x1.45f306p-1")).astype(f32)
    q = ((q + H("0x1.8p+23")).astype(f32) - H("0x1.8p+23")).astype(f32)
    r = fma(q, H("-0x1.921fb0p+00"), x)
    r = fma(q, H("-0x1.5110b4p-22"), r)
    r = fma(q, H("-0x1.846988p-48"), r)
    r2 = (r * r).astype(f32)
    sp = fma(r2, H("0x1.7d3bbcp-19"), H("-0x1.a06bbap-13"))
    for cst in ("0x1.11119ap-07", "-0x1.555556p-03"):
        sp = (sp.astype(np.float64) * r2.astype(np.float64) + np.float64(H(cst))).astype(f32)
    sp = (sp.astype(np.float64) * r2.astype(np.float64)).astype(f32)
    sp = (sp.astype(np.float64) * r.astype(np.float64) + r.astype(np.float64)).astype(f32)
    cp = fma(r2, H("0x1.98e616p-16"), H("-0x1.6c06dcp-10"))
    for cst in ("0x1.55553cp-05", "-0x1.000000p-01", "0x1.000000p+00"):
        cp = (cp.astype(np.float64) * r2.astype(np.float64) + np.float64(H(cst))).astype(f32)
    iq = q.astype(np.int64)
    sin_m = np.where(iq & 2, -np.where(iq & 1, cp, sp), np.where(iq & 1, cp, sp)).astype(f32)
    iqc = iq + 1
    cos_m = np.where(iqc & 2, -np.where(iqc & 1, cp, sp), np.where(iqc & 1, cp, sp)).astype(f32)
    assert np.array_equal(sin_m, np.sin(x))
    assert np.array_equal(cos_m, np.cos(x))


def test_batch_forms_equal_scalar_forms():
    """The reference's own test idea (test/test_rotate_angles.py:10-127, test/test_fov.py:10-155,
    test/test_array_response.py:8-50): vectorised == per-user, including per-user rotations and NaN padding."""
    rng = np.random.default_rng(11)
    n, L = 9, 7
    th = rng.uniform(0, 180, (n, L)).astype(np.float32)
    ph = rng.uniform(-180, 180, (n, L)).astype(np.float32)
    th[3, 4:] = np.nan
    ph[3, 4:] = np.nan
    rot = rng.uniform(-90, 90, (n, 3))
    tb, pb = onp.rotate_angles_batch(rot, th, ph)
    for i in range(n):
        ts, ps = onp.rotate_angles_scalar(rot[i], th[i], ph[i])
        np.testing.assert_array_equal(tb[i], ts)
        np.testing.assert_array_equal(pb[i], ps)
    tb1, pb1 = onp.rotate_angles_batch(np.array([30, 40, 30]), th, ph)
    ts1, ps1 = onp.rotate_angles_scalar(np.array([30, 40, 30]), th[2], ph[2])
    np.testing.assert_array_equal(tb1[2], ts1)
    np.testing.assert_array_equal(pb1[2], ps1)
    with np.errstate(invalid="ignore"):
        mb = onp.fov_mask_batch(np.array([140, 120]), tb, pb)
        for i in range(n):
            np.testing.assert_array_equal(mb[i], onp.fov_mask_scalar(np.array([140, 120]), tb[i], pb[i]))
    a = onp.array_response_batch([4, 2], 0.5, tb, pb)
    for i in (0, 3, 8):
        for l in range(L):
            if np.isnan(tb[i, l]):
                assert np.all(a[i, :, l] == 0)
            else:
                np.testing.assert_allclose(a[i, :, l], onp.array_response_scalar([4, 2], 0.5, tb[i, l], pb[i, l]),
                                           rtol=1e-10, atol=1e-12)          # test_array_response.py: rtol 1e-10


# ---- the committed recipes must keep running (crosscheck_reference.py had rotted unnoticed) -------------
ORACLE_SCRIPTS = ("gen_golden", "gen_p2m_golden", "gen_sequence_golden", "gen_aux_golden", "gen_helpers_golden", "crosscheck_reference",
                  "certify_baseline", "oracle_np", "oracle_c")
REFERENCE = "/root/reference"


def test_every_oracle_script_imports():
    """Importing executes no reference code (each script imports `deepmimo` inside main()), so this runs anywhere."""
    import importlib
    import glob
    here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")
    found = sorted(os.path.basename(p)[:-3] for p in glob.glob(os.path.join(here, "*.py")) if not p.endswith("__init__.py"))
    assert found == sorted(ORACLE_SCRIPTS), f"update ORACLE_SCRIPTS: {found}"
    for name in ORACLE_SCRIPTS:
        mod = importlib.import_module("oracle." + name)
        assert name in ("oracle_np", "oracle_c") or callable(getattr(mod, "main"))


def _run_recipe(script, *args, timeout=600):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=f"{REFERENCE}:{root}", PYTHONDONTWRITEBYTECODE="1", MPLBACKEND="Agg")
    return subprocess.run([sys.executable, os.path.join(root, "oracle", script), *args], cwd="/tmp", env=env,
                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=timeout)


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="the reference exists in the build container only")
def test_crosscheck_against_imported_reference():
    """oracle_np vs the imported reference on the first random configurations of the GPU sweep: bit-identical."""
    r = _run_recipe("crosscheck_reference.py", "10")
    assert r.returncode == 0, r.stdout[-2000:]
    assert "oracle == reference on 10 random configurations; worst |dH|/peak = 0.00e+00" in r.stdout


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="the reference exists in the build container only")
def test_aux_goldens_regenerate_bit_identically(tmp_path):
    """tests/golden/aux_*.npz are what the reference returns today (the generator is deterministic)."""
    import shutil
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gold = os.path.join(root, "tests", "golden")
    names = ["aux_steering.npz", "aux_pathloss.npz", "aux_sionna.npz", "aux_stale_cache.npz", "aux_helpers.npz"]
    keep = {n: dict(np.load(os.path.join(gold, n), allow_pickle=False)) for n in names}
    backup = tmp_path / "backup"
    backup.mkdir()
    for n in names:
        shutil.copy(os.path.join(gold, n), backup / n)
    try:
        for script in ("gen_aux_golden.py", "gen_helpers_golden.py"):
            r = _run_recipe(script)
            assert r.returncode == 0, r.stdout[-2000:]
        for n in names:
            new = np.load(os.path.join(gold, n))
            assert sorted(new.files) == sorted(keep[n])
            for k in new.files:
                assert np.array_equal(new[k], keep[n][k], equal_nan=new[k].dtype.kind in "fc"), (n, k)
    finally:
        for n in names:                                            # leave the committed bytes untouched
            shutil.copy(backup / n, os.path.join(gold, n))
