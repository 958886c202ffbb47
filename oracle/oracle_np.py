"""CPU oracle for the DeepMIMO channel-generation hot path (NumPy restatement).

TEST INFRASTRUCTURE ONLY.  Nothing under ``deepmimo_amd/`` may import this file;
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg use it, and there only as the checker / the timed CPU baseline.

Parity status: PINNED.  ``oracle/gen_golden.py`` imports the real reference
(``/root/reference``, DeepMIMO v4.0.0a3 + its bundled v3 generator) in the build
container and writes ``tests/golden/*.npz``; ``tests/test_oracle_golden.py``
checks this restatement against those vectors (channels <=1e-12 abs, masks /
LoS / path counts exact).

Every function cites the reference lines it restates (paths relative to
``/root/reference``).  The dtype flow of the reference (which intermediate is
float32, which is float64, which is complex64) is reproduced on purpose: it
decides the result at the 1e-5 level ("numerics").

Array conventions: N users, L paths, rays are float32 ``[N, L]`` with NaN = "no
path"; angles in degrees, elevation = zenith angle; channel is complex64
``[N, M_rx, M_tx, K]`` (frequency domain) or ``[N, M_rx, M_tx, L]`` (time domain).
"""
from __future__ import annotations

import copy
import numpy as np

LIGHTSPEED = 299792458.0  # deepmimo_v3/consts.py:112
MAX_PATHS = 25            # deepmimo/consts.py:180

RAY_KEYS = ("power", "phase", "delay", "aoa_az", "aoa_el", "aod_az", "aod_el", "inter")


# --------------------------------------------------------------------------------------
# parameter block (deepmimo/generator/channel.py:33-63)
# --------------------------------------------------------------------------------------
def default_params() -> dict:
    """Plain-dict mirror of ChannelGenParameters.DEFAULT_PARAMS (channel.py:33-63)."""
    return {
        "bs_antenna": {"shape": np.array([8, 1]), "spacing": 0.5,
                       "rotation": np.array([0, 0, 0]), "radiation_pattern": "isotropic"},
        "ue_antenna": {"shape": np.array([1, 1]), "spacing": 0.5,
                       "rotation": np.array([0, 0, 0]), "radiation_pattern": "isotropic"},
        "enable_doppler": 0,
        "enable_dual_polar": 0,
        "num_paths": MAX_PATHS,
        "freq_domain": 1,
        "ofdm": {"subcarriers": 512, "selected_subcarriers": np.arange(1),
                 "bandwidth": 10e6, "rx_filter": 0},
    }


def make_params(**kw) -> dict:
    """default_params() with nested overrides, e.g. make_params(bs_antenna={'shape': [8, 8]})."""
    p = default_params()
    for k, v in kw.items():
        if isinstance(v, dict):
            p[k].update(v)
        else:
            p[k] = v
    for side in ("bs_antenna", "ue_antenna"):
        p[side]["shape"] = np.asarray(p[side]["shape"])
        p[side]["rotation"] = np.asarray(p[side]["rotation"])
    p["ofdm"]["selected_subcarriers"] = np.asarray(p["ofdm"]["selected_subcarriers"])
    return p


# --------------------------------------------------------------------------------------
# geometry (deepmimo/generator/geometry.py)
# --------------------------------------------------------------------------------------
def ant_indices(shape) -> np.ndarray:
    """Element index triples of an [Mh, Mv] panel (geometry.py:105-120).

    x is always 0, y runs fastest over 0..Mh-1, z slowest over 0..Mv-1, so the
    flat element index is m = y + Mh*z.
    """
    mh, mv = int(shape[0]), int(shape[1])
    m = np.arange(mh * mv)
    return np.stack([np.zeros_like(m), m % mh, m // mh], axis=1)


def rotate_angles_batch(rotation_deg, theta_deg, phi_deg):
    """Array-rotation of (zenith, azimuth) pairs (geometry.py:244-319).

    rotation_deg: [3] or [N,3] degrees (about x, y, z); theta/phi: float32 [N, L] degrees.
    dtype flow (measured against the reference): deg2rad keeps float32; sin/cos of the
    zenith angle are evaluated in float32; everything that touches the rotation
    (float64) is float64.  Returns float64 radians; NaN in -> NaN out.
    """
    rot = np.asarray(rotation_deg)
    if rot.ndim == 1:
        rot = rot[None, :]
    rot = np.deg2rad(rot)                      # int/float64 -> float64 (geometry.py:286)
    th = np.deg2rad(theta_deg)                 # float32 stays float32 (geometry.py:284)
    ph = np.deg2rad(phi_deg)
    rx, ry, rz = rot[:, 0:1], rot[:, 1:2], rot[:, 2:3]
    d = ph - rz                                # float32 - float64 -> float64 (geometry.py:294)
    sd, cd = np.sin(d), np.cos(d)
    sy, cy = np.sin(ry), np.cos(ry)
    sx, cx = np.sin(rx), np.cos(rx)
    st, ct = np.sin(th), np.cos(th)            # float32 (geometry.py:301-302)
    th_rot = np.arccos(cy * cx * ct + st * (sy * cx * cd - sx * sd))            # :305-306
    ph_rot = np.angle(cy * st * cd - sy * ct +
                      1j * (cy * sx * ct + st * (sy * sx * cd + cx * sd)))      # :308-310
    return th_rot, ph_rot


def fov_mask_batch(fov_deg, theta, phi) -> np.ndarray:
    """In-field-of-view test on rotated radians (geometry.py:162-195)."""
    th = np.mod(theta, 2 * np.pi)
    ph = np.mod(phi, 2 * np.pi)
    fov = np.deg2rad(fov_deg)
    az_ok = np.logical_or(ph <= 0 + fov[0] / 2, ph >= 2 * np.pi - fov[0] / 2)
    el_ok = np.logical_and(th <= np.pi / 2 + fov[1] / 2, th >= np.pi / 2 - fov[1] / 2)
    return np.logical_and(az_ok, el_ok)


def is_full_fov(fov) -> bool:
    """dataset.py:450-459"""
    return fov[0] >= 360 and fov[1] >= 180


def array_response_batch(shape, spacing, theta, phi) -> np.ndarray:
    """complex128 [N, M, L] array response; NaN zenith -> zero column (geometry.py:38-102)."""
    kd = 2 * np.pi * spacing                   # dataset.py:393
    idx = ant_indices(shape)                   # [M, 3]
    ok = ~np.isnan(theta)
    gy = 1j * kd * np.sin(theta) * np.sin(phi)  # geometry.py:99-101 (x index is always 0)
    gz = 1j * kd * np.cos(theta)
    with np.errstate(invalid="ignore"):
        a = np.exp(idx[None, :, 1, None] * gy[:, None, :] + idx[None, :, 2, None] * gz[:, None, :])
    return np.where(ok[:, None, :], a, 0.0 + 0.0j)


# scalar (per-user) twins: the reference keeps both forms and its tests compare them
# (test/test_rotate_angles.py, test/test_fov.py, test/test_array_response.py); so does tests/test_oracle_golden.py
def rotate_angles_scalar(rotation_deg, theta_deg, phi_deg):
    """One user's paths, one rotation triple (geometry.py:198-241).  rotation None = degrees to radians only."""
    th, ph = np.deg2rad(theta_deg), np.deg2rad(phi_deg)
    if rotation_deg is None:
        return th, ph
    rx, ry, rz = np.deg2rad(rotation_deg)
    d = ph - rz
    st, ct = np.sin(th), np.cos(th)
    th_rot = np.arccos(np.cos(ry) * np.cos(rx) * ct + st * (np.sin(ry) * np.cos(rx) * np.cos(d) - np.sin(rx) * np.sin(d)))
    ph_rot = np.angle(np.cos(ry) * st * np.cos(d) - np.sin(ry) * ct +
                      1j * (np.cos(ry) * np.sin(rx) * ct + st * (np.sin(ry) * np.sin(rx) * np.cos(d) + np.cos(rx) * np.sin(d))))
    return th_rot, ph_rot


def fov_mask_scalar(fov_deg, theta, phi) -> np.ndarray:
    """geometry.py:123-159 (same arithmetic as the batch form, 1-D inputs)."""
    return fov_mask_batch(fov_deg, np.asarray(theta), np.asarray(phi))


def array_response_scalar(shape, spacing, theta: float, phi: float) -> np.ndarray:
    """[M] response of one path (geometry.py:19-35)."""
    idx = ant_indices(shape)
    kd = 2 * np.pi * spacing
    gamma = 1j * kd * np.array([np.sin(theta) * np.cos(phi), np.sin(theta) * np.sin(phi), np.cos(theta)])
    return np.exp(idx @ gamma)


def steering_vec(shape, phi=0.0, theta=0.0, spacing=0.5) -> np.ndarray:
    """Normalised beam-steering vector [M, 1] (geometry.py:322-339).

    The reference feeds (phi, theta + 90 deg) into the (theta, phi) slots of the array
    response; that swap is reference behaviour and is kept.
    """
    th = np.array([[phi * np.pi / 180]])
    ph = np.array([[theta * np.pi / 180 + np.pi / 2]])
    a = array_response_batch(shape, spacing, th, ph)[0]
    return a / np.linalg.norm(a)


# --------------------------------------------------------------------------------------
# antenna patterns (deepmimo/generator/ant_patterns.py)
# --------------------------------------------------------------------------------------
def pattern_gain(name: str, theta):
    """Element power gain: python float 1.0 (isotropic, :21-31) or the half-wave dipole
    1.643*cos^2(pi/2 cos t)/sin t where |sin t| > 1e-10 else 0 (:34-71; NaN -> 0)."""
    if name == "isotropic":
        return 1.0
    if name != "halfwave-dipole":
        raise NotImplementedError(f"The given '{name}' antenna radiation pattern is not applicable.")
    theta = np.asarray(theta)
    g = np.zeros_like(theta, dtype=np.float64)
    with np.errstate(invalid="ignore"):
        ok = np.abs(np.sin(theta)) > 1e-10
    t = theta[ok]
    g[ok] = 1.643 * (np.cos(np.pi / 2 * np.cos(t)) ** 2 / np.sin(t))
    return g


def dbw2watt(p):
    """generator_utils.py:23-35 (float32 in -> float32 out)."""
    return 10 ** (p / 10)


# --------------------------------------------------------------------------------------
# path -> subcarrier gains (deepmimo/generator/channel.py:141-198 + v3 Doppler)
# --------------------------------------------------------------------------------------
def ofdm_path_gains(power, delay, phase, ofdm: dict, doppler=None):
    """[L', K] complex path gains of ONE user's valid paths (channel.py:170-198).

    power: linear W (float32 isotropic / float64 dipole), delay: s, phase: deg.
    doppler: None or (vel, acc, carrier_freq) -> v3 term construct_deepmimo.py:267-280.
    """
    n_sc = ofdm["subcarriers"]
    sc = np.asarray(ofdm["selected_subcarriers"])
    ts = 1 / ofdm["bandwidth"]                               # channel.py:223
    pw = np.array(power, copy=True).reshape(-1, 1)
    dn = delay.reshape(-1, 1) / ts                            # float32 / weak python float
    ph = phase.reshape(-1, 1)
    over = dn >= n_sc                                         # channel.py:187-189
    pw[over] = 0
    dn[over] = n_sc
    c = np.sqrt(pw / n_sc) * np.exp(1j * np.deg2rad(ph))      # complex64 for float32 power
    if ofdm["rx_filter"]:
        d = np.arange(n_sc)
        w = np.exp(-1j * 2 * np.pi / n_sc * np.outer(d, sc))  # channel.py:166-168
        taps = c * np.sinc(d - dn)                             # [L', N]
        if doppler is not None:
            vel, acc, fc = doppler
            tau = ts * d[None, :]
            taps = taps * np.exp(-1j * 2 * np.pi * fc * (vel.reshape(-1, 1) * tau / LIGHTSPEED +
                                                         acc.reshape(-1, 1) * tau ** 2 / (2 * LIGHTSPEED)))
        return taps @ w
    g = c * np.exp(-1j * (2 * np.pi / n_sc) * np.outer(dn.ravel(), sc))   # channel.py:196-197
    if doppler is not None:
        vel, acc, fc = doppler
        tau = delay.reshape(-1, 1)
        g = g * np.exp(-1j * 2 * np.pi * fc * (vel.reshape(-1, 1) * tau / LIGHTSPEED +
                                               acc.reshape(-1, 1) * tau ** 2 / (2 * LIGHTSPEED)))
    return g


# --------------------------------------------------------------------------------------
# the whole path: Dataset.compute_channels (deepmimo/generator/dataset.py:224-268)
# --------------------------------------------------------------------------------------
def resolve_ue_rotation(rot, n_ue: int) -> np.ndarray:
    """[N,3] per-user UE rotation (dataset.py:327-338).  A (3,2) array is a [lo, hi] range:
    drawn with the *global* NumPy RNG exactly like the reference (caller seeds 1001)."""
    rot = np.asarray(rot)
    if rot.ndim == 1 and rot.shape[0] == 3:
        return np.tile(rot, (n_ue, 1))
    if rot.ndim == 2 and rot.shape == (3, 2):
        return np.random.uniform(rot[:, 0], rot[:, 1], (n_ue, 3))
    return rot


def prepare_paths(rays: dict, params: dict, bs_fov=None, ue_fov=None) -> dict:
    """Everything before the paths x antennas x subcarriers sum, for all users at once.

    Restates Dataset._compute_rotated_angles (dataset.py:310-356), _compute_fov (:461-512),
    _compute_power_linear_ant_gain (:665-696), _compute_num_paths (:613-619) and
    _compute_los (:569-611).  Operates on ALL loaded paths (the num_paths slice is applied
    by the caller, dataset.py:258-261).
    """
    n_ue = rays["power"].shape[0]
    bs, ue = params["bs_antenna"], params["ue_antenna"]
    ue_rot = resolve_ue_rotation(ue["rotation"], n_ue)
    aod_el, aod_az = rotate_angles_batch(bs["rotation"], rays["aod_el"], rays["aod_az"])
    aoa_el, aoa_az = rotate_angles_batch(ue_rot, rays["aoa_el"], rays["aoa_az"])
    out = {"_aod_el_rot": aod_el, "_aod_az_rot": aod_az, "_aoa_el_rot": aoa_el, "_aoa_az_rot": aoa_az,
           "ue_rotation": ue_rot}

    # Dataset.apply_fov (dataset.py:423-448) always stores both FoVs (defaults [360, 180]); a side that
    # was never set while the other one was is therefore the full sphere.
    if bs_fov is not None and ue_fov is None:
        ue_fov = np.array([360, 180])
    if ue_fov is not None and bs_fov is None:
        bs_fov = np.array([360, 180])
    bs_full = bs_fov is not None and is_full_fov(bs_fov)
    ue_full = ue_fov is not None and is_full_fov(ue_fov)
    if (bs_fov is None and ue_fov is None) or (bs_full and ue_full):   # dataset.py:484
        mask = None
    else:
        mask = np.ones(aod_el.shape, dtype=bool)
        with np.errstate(invalid="ignore"):
            if not bs_full:                                             # dataset.py:497
                mask &= fov_mask_batch(bs_fov, aod_el, aod_az)
            if not ue_full:
                mask &= fov_mask_batch(ue_fov, aoa_el, aoa_az)
        aod_el, aod_az = np.where(mask, aod_el, np.nan), np.where(mask, aod_az, np.nan)
        aoa_el, aoa_az = np.where(mask, aoa_el, np.nan), np.where(mask, aoa_az, np.nan)
    out.update({"_fov_mask": mask, "_aod_el_rot_fov": aod_el, "_aod_az_rot_fov": aod_az,
                "_aoa_el_rot_fov": aoa_el, "_aoa_az_rot_fov": aoa_az})

    out["power_linear"] = dbw2watt(rays["power"])
    gain = pattern_gain(bs["radiation_pattern"], aod_el) * pattern_gain(ue["radiation_pattern"], aoa_el)
    out["_power_linear_ant_gain"] = out["power_linear"] * gain        # ant_patterns.py:167-168

    out["num_paths"] = (~np.isnan(aoa_az)).sum(axis=1)
    inter = rays["inter"]
    los = np.full(n_ue, -1)
    if mask is not None:
        has = mask.any(axis=1)
        first = np.full(n_ue, -1.0)
        j = np.argmax(mask, axis=1)                                     # first in-FoV path (dataset.py:594-598)
        first[has] = inter[np.arange(n_ue)[has], j[has]]
    else:
        has = out["num_paths"] > 0
        first = inter[:, 0]
    los[has] = 0
    with np.errstate(invalid="ignore"):
        los[(first == 0) & has] = 1
    out["los"] = los
    return out


def compute_channels(rays: dict, params: dict, bs_fov=None, ue_fov=None, doppler=None,
                     style: str = "batched", users=None) -> dict:
    """Restatement of Dataset.compute_channels (dataset.py:224-268) on raw arrays.

    rays: dict of float32 [N, L] arrays (RAY_KEYS); params: make_params() dict;
    bs_fov/ue_fov: what Dataset.apply_fov stored (None = never called);
    doppler: None or dict(vel=[N,L], acc=[N,L], carrier_freq=float) - the v3 term
    (construct_deepmimo.py:267-280) applied on the v4 path when params['enable_doppler'];
    style: 'reference' = the reference's own per-user loop with a complex128 broadcast
    product + nansum (channel.py:264-287; this is what bench.py times as cpu_baseline),
    'batched' = same arithmetic as one einsum per user block (fast, for tests).
    users: optional slice/index array restricting the final sum (the prep is cheap).
    Returns dict(channel, los, num_paths, _fov_mask, rotated angles, powers ...).
    """
    params = copy.deepcopy(params)
    np.random.seed(1001)                                                # dataset.py:250
    prep = prepare_paths(rays, params, bs_fov, ue_fov)
    P = int(params["num_paths"])
    bs, ue, ofdm = params["bs_antenna"], params["ue_antenna"], params["ofdm"]
    sel = slice(None) if users is None else users

    a_tx = array_response_batch(bs["shape"], bs["spacing"],
                                prep["_aod_el_rot_fov"][sel], prep["_aod_az_rot_fov"][sel])[..., :P]
    a_rx = array_response_batch(ue["shape"], ue["spacing"],
                                prep["_aoa_el_rot_fov"][sel], prep["_aoa_az_rot_fov"][sel])[..., :P]
    power = prep["_power_linear_ant_gain"][sel][..., :P]
    delay = rays["delay"][sel][..., :P]
    phase = rays["phase"][sel][..., :P]
    use_dop = bool(params["enable_doppler"]) and doppler is not None
    if use_dop:
        vel, acc = doppler["vel"][sel][..., :P], doppler["acc"][sel][..., :P]

    n, m_rx, m_tx = power.shape[0], a_rx.shape[1], a_tx.shape[1]
    fd = bool(params["freq_domain"])
    last = len(ofdm["selected_subcarriers"]) if fd else power.shape[1]
    H = np.zeros((n, m_rx, m_tx, last), dtype=np.csingle)               # channel.py:257
    if fd:
        with np.errstate(invalid="ignore"):
            prep["max_delay"] = np.nanmax(delay) if delay.size and not np.all(np.isnan(delay)) else np.nan
        prep["delay_exceeds_symbol"] = bool(prep["max_delay"] > ofdm["subcarriers"] / ofdm["bandwidth"])
    valid = ~np.isnan(power)                                            # channel.py:260
    for i in range(n):
        v = valid[i]
        nv = int(v.sum())
        if nv == 0:
            continue
        if fd:
            dop = (vel[i, v], acc[i, v], doppler["carrier_freq"]) if use_dop else None
            g = ofdm_path_gains(power[i, v], delay[i, v], phase[i, v], ofdm, dop)     # [L', K]
            if style == "reference":
                arp = a_rx[i][:, None, :] * a_tx[i][None, :, :]        # dataset.py:417
                arp = arp[..., v]
                H[i] = np.nansum(arp[..., None, :] * g.T[None, None, :, :], axis=-1)  # channel.py:283-284
            else:
                t = a_rx[i][:, None, v] * a_tx[i][None, :, v]           # [M_rx, M_tx, L']
                prod_ok = ~(np.isnan(t).any(axis=(0, 1)) | np.isnan(g).any(axis=1))
                # nansum semantics: a NaN product contributes 0.  Path-level NaN (angle/delay/phase)
                # poisons every element of that path's slice, so dropping the path is identical.
                H[i] = np.einsum("rtl,lk->rtk", t[..., prod_ok], g[prod_ok])
        else:
            pg = np.sqrt(power[i, v]) * np.exp(1j * np.deg2rad(phase[i, v]))           # channel.py:286
            H[i, ..., :nv] = a_rx[i][:, None, v] * a_tx[i][None, :, v] * pg[None, None, :]
    prep["channel"] = H
    return prep


# --------------------------------------------------------------------------------------
# synthetic rays (the generator every test and bench shares)
# --------------------------------------------------------------------------------------
def synth_rays(n_ue: int, n_paths: int, seed: int = 0, all_valid: bool = False,
               max_delay: float = 2e-6, with_doppler: bool = False) -> dict:
    """Synthetic ray matrices with the reference's storage conventions (core.py:209-219):
    float32 [N, L], trailing-NaN padded, per-user valid count uniform in 0..L."""
    rng = np.random.default_rng(seed)
    shp = (n_ue, n_paths)
    r = {
        "power": rng.uniform(-140, -60, shp), "phase": rng.uniform(-180, 180, shp),
        "delay": rng.uniform(1e-8, max_delay, shp),
        "aoa_az": rng.uniform(-180, 180, shp), "aoa_el": rng.uniform(0, 180, shp),
        "aod_az": rng.uniform(-180, 180, shp), "aod_el": rng.uniform(0, 180, shp),
        "inter": rng.integers(0, 5, shp).astype(np.float64),
    }
    if with_doppler:
        r["doppler_vel"] = rng.uniform(-30, 30, shp)
        r["doppler_acc"] = rng.uniform(-1, 1, shp)
    nvalid = np.full(n_ue, n_paths) if all_valid else rng.integers(0, n_paths + 1, n_ue)
    pad = np.arange(n_paths)[None, :] >= nvalid[:, None]
    out = {}
    for k, v in r.items():
        v = v.astype(np.float32)
        v[pad] = np.nan
        out[k] = v
    out["rx_pos"] = rng.uniform(-100, 100, (n_ue, 3)).astype(np.float32)
    out["tx_pos"] = np.zeros((1, 3), np.float32)
    return out
