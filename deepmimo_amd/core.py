"""``generate`` / ``load`` - the outer entry points of the channel path.

``generate(scen_name, load_params, ch_gen_params)`` keeps the reference's name and signature
(deepmimo/generator/core.py:36-61) with its evidently intended behaviour: load the scenario, then
``compute_channels``.  (At the reference snapshot the function calls a method that does not exist,
``dataset._compute_channels`` - core.py:59 - and raises KeyError; SURVEY.md finding 3.)

``load`` reads a converted scenario folder in the reference's on-disk format
(docs/resources/specs.md:10-69): ``params.json`` plus one MATLAB-v5 ``.mat`` file per matrix and
(TX set, TX index, RX set) triple named ``{key}_t{tx_set:03}_tx{tx_idx:03}_r{rx_set:03}.mat``
(general_utils.py:296-323), with the ``max_paths`` / ``tx_sets`` / ``rx_sets`` / ``matrices``
selection rules of core.py:139-258, 261-338.  There is no network here, so a missing scenario is
an error instead of a download prompt (core.py:103-109).  Scene / material objects (plot-only
metadata) are left as the raw dictionaries.

``load(..., device='cuda')`` (extension) keeps the eight per-path matrices out of
host NumPy altogether: each file is memory-mapped, its payload copied to HBM as stored and laid out
row-major by a device pass (deepmimo_amd/matio.py); the Dataset then holds torch tensors for those
fields and ``compute_channels`` uses them in place.
"""
from __future__ import annotations

import json
import os
from typing import Any, Dict, List

import numpy as np

from . import consts as c
from .channel import ChannelGenParameters
from .config import config
from .dataset import Dataset, MacroDataset

_MATRIX_KEYS = (c.AOA_AZ_PARAM_NAME, c.AOA_EL_PARAM_NAME, c.AOD_AZ_PARAM_NAME, c.AOD_EL_PARAM_NAME,
                c.POWER_PARAM_NAME, c.PHASE_PARAM_NAME, c.DELAY_PARAM_NAME, c.RX_POS_PARAM_NAME,
                c.TX_POS_PARAM_NAME, c.INTERACTIONS_PARAM_NAME, c.INTERACTIONS_POS_PARAM_NAME)


def get_mat_filename(key: str, tx_set_idx: int, tx_idx: int, rx_set_idx: int) -> str:
    """general_utils.py:296-323"""
    return f"{key}_t{tx_set_idx:03}_tx{tx_idx:03}_r{rx_set_idx:03}.mat"


def get_scenario_folder(scen_name: str) -> str:
    """general_utils.py:48-73: <cwd>/<config scenarios_folder>/<name>"""
    return os.path.join(os.getcwd(), config.get("scenarios_folder"), scen_name)


def generate(scen_name: str, load_params: Dict[str, Any] = {}, ch_gen_params: Dict[str, Any] = {}):
    """Load a scenario and compute its channels (fan-out over all TX/RX pairs of a MacroDataset)."""
    dataset = load(scen_name, **load_params)
    ch_params = ch_gen_params if ch_gen_params else ChannelGenParameters()
    if not isinstance(ch_params, ChannelGenParameters):
        ch_params = ChannelGenParameters(dict(ch_params))
    _ = dataset.compute_channels(ch_params)
    return dataset


def load(scen_name: str, **load_params):
    """core.py:63-137 without the download prompt.  Extra keyword: device=None | 'cuda' | 'cuda:N'."""
    if os.path.isabs(scen_name):
        folder, scen_name = scen_name, os.path.basename(scen_name.rstrip(os.sep))
    else:
        folder = get_scenario_folder(scen_name)
    if not os.path.exists(folder):
        raise ValueError(f"Scenario {scen_name} not found")
    params_file = os.path.join(folder, "params.json")
    if not os.path.exists(params_file):
        raise ValueError(f"Parameters file not found in {folder}")
    with open(params_file) as f:
        params = json.load(f)
    n_snap = params.get(c.SCENE_PARAM_NAME, {}).get("num_scenes", 1)
    if n_snap > 1:
        raise NotImplementedError("Dynamic scenarios not implemented yet")
    dataset = _load_raytracing_scene(folder, params[c.TXRX_PARAM_NAME], **load_params)
    dataset["name"] = scen_name
    dataset[c.LOAD_PARAMS_PARAM_NAME] = load_params
    dataset[c.RT_PARAMS_PARAM_NAME] = params.get(c.RT_PARAMS_PARAM_NAME, {})
    dataset[c.SCENE_PARAM_NAME] = params.get(c.SCENE_PARAM_NAME)
    dataset[c.MATERIALS_PARAM_NAME] = params.get(c.MATERIALS_PARAM_NAME)
    return dataset


def _validate_txrx_sets(sets, txrx_dict: Dict[str, Any], tx_or_rx: str) -> Dict[int, np.ndarray]:
    """core.py:261-338: 'all' | [set ids] | {set id: 'all' | indices}."""
    flag = "is_tx" if tx_or_rx == "tx" else "is_rx"
    valid = [txrx_dict[k]["id"] for k in sorted(txrx_dict.keys()) if txrx_dict[k][flag]]
    name = "Tx" if tx_or_rx == "tx" else "Rx"
    hint = "To see supported TX/RX sets and indices run dm.info(<scenario_name>)"

    def n_points(set_id):
        return txrx_dict[f"txrx_set_{set_id}"]["num_points"]

    if isinstance(sets, dict):
        out = {}
        for set_id, idxs in sets.items():
            if set_id not in valid:
                raise Exception(f"{name} set {set_id} not in allowed sets {valid}\n" + hint)
            avail = np.arange(n_points(set_id))
            if isinstance(idxs, str):
                if idxs != "all":
                    raise Exception(f"String '{idxs}' not recognized for tx/rx indices ")
                idxs = avail
            elif isinstance(idxs, (list, np.ndarray)):
                idxs = np.asarray(idxs)
            else:
                raise Exception("Only <list> of <np.ndarray> allowed as tx/rx indices")
            if not set(idxs.tolist()).issubset(set(avail.tolist())):
                raise Exception(f"Some indices of {idxs} are not in {avail}. " + hint)
            out[set_id] = idxs
        return out
    if isinstance(sets, list):
        for set_id in sets:
            if set_id not in valid:
                raise Exception(f"{name} set {set_id} not in allowed sets {valid}\n" + hint)
        return {s: np.arange(n_points(s)) for s in sets}
    if isinstance(sets, str):
        if sets != "all":
            raise Exception(f"String '{sets}' not understood. Only string allowed is 'all' to generate all "
                            "available sets and indices")
        return {s: np.arange(n_points(s)) for s in valid}
    raise Exception("tx_sets / rx_sets must be a dict, a list or 'all'")


def _load_raytracing_scene(folder: str, txrx_dict: dict, max_paths: int = c.MAX_PATHS, tx_sets="all",
                           rx_sets="all", matrices="all", device=None):
    tx_sets = _validate_txrx_sets(tx_sets, txrx_dict, "tx")
    rx_sets = _validate_txrx_sets(rx_sets, txrx_dict, "rx")
    out: List[Dict[str, Any]] = []
    for tx_set_id, tx_idxs in tx_sets.items():
        for rx_set_id, rx_idxs in rx_sets.items():
            for tx_idx in tx_idxs:
                print(f"Loading TXRX PAIR: TXset {tx_set_id} (tx_idx {tx_idx}) & RXset {rx_set_id} "
                      f"(rx_idxs {len(rx_idxs)})")
                d = _load_tx_rx_raydata(folder, tx_set_id, rx_set_id, int(tx_idx), rx_idxs, max_paths, matrices, device)
                d["txrx"] = {"tx_set_id": tx_set_id, "rx_set_id": rx_set_id, "tx_idx": int(tx_idx)}
                out.append(d)
    if len(out) > 1:
        return MacroDataset([Dataset(d) for d in out])
    return Dataset(out[0])


def _load_tx_rx_raydata(folder: str, tx_set_id: int, rx_set_id: int, tx_idx: int, rx_idxs, max_paths: int,
                        matrices="all", device=None) -> Dict[str, Any]:
    """core.py:186-258"""
    import scipy.io
    if isinstance(matrices, str) and matrices == "all":
        wanted = list(_MATRIX_KEYS)
    else:
        wanted = [] if matrices is None else list(matrices)
        bad = set(wanted) - set(_MATRIX_KEYS)
        if bad:
            raise ValueError(f"Invalid matrix names: {bad}. Valid names are: {set(_MATRIX_KEYS)}")
    d: Dict[str, Any] = {k: None for k in _MATRIX_KEYS}
    on_device = []
    for key in _MATRIX_KEYS:
        if key not in wanted:
            continue
        path = os.path.join(folder, get_mat_filename(key, tx_set_id, tx_idx, rx_set_id))
        if not os.path.exists(path):
            print(f"File {path} could not be found")
            continue
        if device is not None and key in c.RAY_FIELDS:
            on_device.append((path, key))                       # all ray matrices of the pair in one pipeline, below
            continue
        m = None
        if device is not None:                                  # beside the device pipeline: the library's parser, not loadmat's 0.45 ms per file
            try:
                from .matio import read_matrix_host
                m = read_matrix_host(path, key)
            except Exception:
                m = None                                        # anything the parser does not cover: as the reference reads it
        if m is None:
            m = scipy.io.loadmat(path)[key]
        if key != c.TX_POS_PARAM_NAME:
            m = m[rx_idxs]
        if key not in (c.RX_POS_PARAM_NAME, c.TX_POS_PARAM_NAME):
            m = m[:, :max_paths, ...]
        d[key] = m
    if on_device:
        from .matio import load_matrices_to_device
        d.update(load_matrices_to_device(on_device, device, rx_idxs=rx_idxs, max_paths=max_paths))
    return d
