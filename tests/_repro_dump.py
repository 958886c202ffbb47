"""What a bit-reproducibility test saves when two identical launches differ (tests/test_gpu_parity.py,
tests/test_gpu_fullsize.py, tools/repro_stress.py).  Round 2 lost the evidence: four builds of the folded kernel
returned corrupted 32-row tiles once per 200 ... 10^9 users and only indices and max |d| were printed, so "one MFMA's
contribution missing" against "another tile's / K-step's data" was inferred, never shown.  One failure now
classifies itself: for every differing user (up to four) the file holds

  * both launches' values of the whole user block and the user's ray records,
  * the float64 oracle's PER-PATH contributions H_l[p, k] over the bounding box of the differing elements widened to the
    32-row (antenna pair, 16-subcarrier block) tiles of the folded kernel and one tile either side - their partial sums
    over the kernels' 8-path K-steps (stage 1 keeps the path order unless adaptive precision is on) are the terms a lost
    matrix-core instruction would take with it,
  * and a verdict per user: which launch is the wrong one, and whether its error is (a) minus one K-step's partial sum,
    (b) minus / a multiple of one path's contribution, (c) the values of a neighbouring tile or of another launch
    position ("foreign data"), or (d) none of these.

The oracle is the checker here, as everywhere under tests/.
"""
from __future__ import annotations

import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT_DIR = os.path.join(os.environ.get("GRAFT_REPO_ROOT", ROOT), "gpurun_out")


def _oracle_params(dm_params):
    from oracle import oracle_np as onp
    bs, ue, ofdm = dm_params["bs_antenna"], dm_params["ue_antenna"], dm_params["ofdm"]
    return onp.make_params(
        bs_antenna=dict(shape=np.asarray(bs["shape"]), spacing=float(bs["spacing"]), rotation=np.asarray(bs["rotation"]),
                        radiation_pattern=bs["radiation_pattern"]),
        ue_antenna=dict(shape=np.asarray(ue["shape"]), spacing=float(ue["spacing"]), rotation=np.asarray(ue["rotation"]),
                        radiation_pattern=ue["radiation_pattern"]),
        num_paths=int(dm_params["num_paths"]), freq_domain=int(bool(dm_params["freq_domain"])),
        ofdm=dict(subcarriers=int(ofdm["subcarriers"]), selected_subcarriers=np.asarray(ofdm["selected_subcarriers"]),
                  bandwidth=float(ofdm["bandwidth"]), rx_filter=int(bool(ofdm["rx_filter"]))))


def per_path_contributions(rays_u: dict, oparams: dict):
    """float64 contributions of ONE user's paths: [L_valid, M_rx * M_tx, K] complex128 (channel.py:281-284 before the
    sum), in path order, and the indices of those paths."""
    from oracle import oracle_np as onp
    prep = onp.prepare_paths(rays_u, oparams)
    P = int(oparams["num_paths"])
    bs, ue, ofdm = oparams["bs_antenna"], oparams["ue_antenna"], oparams["ofdm"]
    a_tx = onp.array_response_batch(bs["shape"], bs["spacing"], prep["_aod_el_rot_fov"], prep["_aod_az_rot_fov"])[0][..., :P]
    a_rx = onp.array_response_batch(ue["shape"], ue["spacing"], prep["_aoa_el_rot_fov"], prep["_aoa_az_rot_fov"])[0][..., :P]
    power = prep["_power_linear_ant_gain"][0][:P]
    v = ~np.isnan(power)
    g = onp.ofdm_path_gains(power[v], rays_u["delay"][0][:P][v], rays_u["phase"][0][:P][v], ofdm, None)   # [L', K]
    t = (a_rx[:, None, v] * a_tx[None, :, v]).reshape(-1, int(v.sum()))                                  # [M, L']
    ok = ~(np.isnan(t).any(axis=0) | np.isnan(g).any(axis=1) | (np.abs(g).max(axis=1) == 0))
    contrib = t.T[ok][:, :, None] * g[ok][:, None, :]
    return contrib.astype(np.complex128), np.nonzero(v)[0][ok]


def classify(first_u: np.ndarray, again_u: np.ndarray, contrib: np.ndarray, rows: slice, cols: slice, tol=2e-5):
    """first_u / again_u: [M, K] complex64 of one user; contrib: [L', M, K].  Returns a dict."""
    ref = contrib.sum(axis=0)
    peak = np.abs(ref).max() + 1e-300
    e_first = np.abs(first_u - ref).max() / peak
    e_again = np.abs(again_u - ref).max() / peak
    wrong, good = (first_u, again_u) if e_first > e_again else (again_u, first_u)
    res = {"err_first_launch": float(e_first), "err_repeat_launch": float(e_again),
           "wrong_launch": "first" if e_first > e_again else "repeat"}
    diff = (wrong.astype(np.complex128) - ref)[rows, cols]
    box_peak = np.abs(diff).max() + 1e-300
    cands = []
    nl = contrib.shape[0]
    for s in range((nl + 7) // 8):
        part = contrib[8 * s:8 * s + 8].sum(axis=0)[rows, cols]
        cands.append((f"minus the partial sum of K-step {s} (kept paths {8 * s}..{min(nl, 8 * s + 8) - 1})", -part))
    for l in range(nl):
        cands.append((f"minus the contribution of kept path {l}", -contrib[l][rows, cols]))
    best = None
    for name, c in cands:
        r = np.abs(diff - c).max() / box_peak
        if best is None or r < best[1]:
            best = (name, float(r))
    res["best_missing_term"] = {"what": best[0], "residual_over_error": best[1]}
    # a scaled single path (a stale table entry multiplies ONE path's contribution by a wrong unit phasor)
    sc_best = None
    for l in range(nl):
        c = contrib[l][rows, cols].ravel()
        z = np.vdot(c, diff.ravel()) / (np.vdot(c, c) + 1e-300)
        r = np.abs(diff.ravel() - z * c).max() / box_peak
        if sc_best is None or r < sc_best[2]:
            sc_best = (l, complex(z), float(r))
    res["best_scaled_path"] = {"kept_path": int(sc_best[0]), "factor": [sc_best[1].real, sc_best[1].imag],
                               "residual_over_error": sc_best[2]}
    # foreign data: the wrong values equal the CORRECT values of the same box shifted by whole tiles (32 rows / 16 columns)
    foreign = None
    M, K = ref.shape
    r0, r1, c0, c1 = rows.start, rows.stop, cols.start, cols.stop
    for dr in (-64, -32, 0, 32, 64):
        for dc in (-32, -16, 0, 16, 32):
            if (dr, dc) == (0, 0) or r0 + dr < 0 or r1 + dr > M or c0 + dc < 0 or c1 + dc > K:
                continue
            r = np.abs(wrong[r0:r1, c0:c1] - good[r0 + dr:r1 + dr, c0 + dc:c1 + dc]).max() / (np.abs(good).max() + 1e-300)
            if foreign is None or r < foreign[2]:
                foreign = (dr, dc, float(r))
    if foreign is not None:
        res["best_foreign_tile"] = {"row_shift": foreign[0], "col_shift": foreign[1], "mismatch_over_peak": foreign[2]}
    if best[1] < 0.05:
        res["verdict"] = "MISSING TERM: " + best[0]
    elif sc_best[2] < 0.05:
        res["verdict"] = f"ONE PATH SCALED: kept path {sc_best[0]} times (1 + {sc_best[1]:.3g})"
    elif foreign is not None and foreign[2] < tol:
        res["verdict"] = f"FOREIGN DATA: the values of the tile {foreign[0]} rows / {foreign[1]} subcarriers away"
    else:
        res["verdict"] = "unclassified (see arrays)"
    res["error_over_user_peak"] = float(box_peak / peak)
    return res


def dump_from_checksums(tag: str, eng, prep, dm_params, rays: dict, H, bad_users, variant=0, max_users: int = 4) -> str:
    """For tests that compare per-user checksums against a first launch whose values are gone: the differing users'
    current values against a fresh single-user launch of each (classify() decides by the oracle which one is wrong)."""
    import torch
    ids = [int(u) for u in list(bad_users)[:max_users]]
    cur = torch.stack([H[u].clone() for u in ids])
    fresh = torch.stack([eng.channels(prep, user_begin=u, user_count=1, variant=variant)[0] for u in ids])
    sub = {k: np.asarray(v)[ids] for k, v in rays.items() if hasattr(v, "shape") and len(v.shape) == 2}
    return dump_mismatch(tag, dm_params, sub, fresh, cur, torch.ones(len(ids), dtype=torch.bool), variant=variant,
                         user_ids=ids, total=len(list(bad_users)))


def dump_mismatch(tag: str, dm_params, rays: dict, first, again, bad_mask, variant=0, max_users: int = 4, user_ids=None,
                  total=None) -> str:
    """first / again: torch complex64 [N, M_rx, M_tx, K] (device or host); bad_mask: bool [N] of differing users."""
    import torch
    os.makedirs(OUT_DIR, exist_ok=True)
    bad = torch.nonzero(bad_mask).flatten().cpu().numpy()[:max_users]
    oparams = _oracle_params(dm_params)
    arrays, report = {}, {"tag": tag, "variant": int(variant),
                          "differing_users_total": int(total if total is not None else bad_mask.sum()), "users": []}
    for u in bad:
        u = int(u)
        f = first[u].reshape(-1, first.shape[-1]).cpu().numpy()
        a = again[u].reshape(-1, again.shape[-1]).cpu().numpy()
        d = np.nonzero(f.view(np.uint32) != a.view(np.uint32))
        rr, cc = d[0], d[1] // 2
        M, K = f.shape
        if len(rr) == 0:                                        # (checksum form) the launch whose values are gone was the odd one
            report["users"].append({"user": int(user_ids[u]) if user_ids is not None else u,
                                    "note": "current values equal a fresh launch: the FIRST launch differed (values not kept)"})
            continue
        # rows of the folded kernel's tiles are (16-subcarrier block, antenna pair); report both views
        r0, r1 = int(rr.min()), int(rr.max()) + 1
        c0, c1 = int(cc.min()) // 16 * 16, min(K, (int(cc.max()) // 16 + 1) * 16)
        rays_u = {k: np.asarray(v[u:u + 1]) for k, v in rays.items() if hasattr(v, "shape") and len(v.shape) == 2}
        entry = {"user": int(user_ids[u]) if user_ids is not None else u, "pairs": [r0, r1], "subcarriers": [c0, c1], "elements_differing": int(len(rr)),
                 "fold_tile_rows": f"blocks {c0 // 16}..{(c1 - 1) // 16} x pairs {r0}..{r1 - 1}"}
        try:
            contrib, kept = per_path_contributions(rays_u, oparams)
            entry["kept_paths"] = kept.tolist()
            entry.update(classify(f, a, contrib, slice(r0, r1), slice(c0, c1)))
            lo, hi = max(0, c0 - 32), min(K, c1 + 32)
            arrays[f"u{u}_contrib_box"] = contrib[:, :, lo:hi].astype(np.complex64)
            arrays[f"u{u}_contrib_box_cols"] = np.array([lo, hi])
        except Exception as exc:                               # the dump must not hide the original failure
            entry["classification_error"] = repr(exc)
        arrays[f"u{u}_first"], arrays[f"u{u}_repeat"] = f, a
        for k, v in rays_u.items():
            arrays[f"u{u}_ray_{k}"] = v
        report["users"].append(entry)
    base = os.path.join(OUT_DIR, f"repro_dump_{tag}")
    np.savez_compressed(base + ".npz", **arrays)
    with open(base + ".json", "w") as fh:
        json.dump(report, fh, indent=1)
    print("REPRODUCIBILITY MISMATCH", json.dumps(report, indent=1))
    return base + ".{npz,json}"
