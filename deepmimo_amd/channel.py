"""Channel-generation parameter block: ``ChannelGenParameters`` with the reference's defaults
(deepmimo/generator/channel.py:33-63) and ``validate`` rules (:78-139).  The arithmetic that
channel.py holds in the reference (OFDM_PathGenerator, _generate_MIMO_channel) lives in the HIP
kernels; see deepmimo_amd/engine.py for the driver."""
from __future__ import annotations

from copy import deepcopy
from typing import Dict, Optional

import numpy as np

from . import consts as c
from .general_utils import DotDict, compare_two_dicts


class ChannelGenParameters(DotDict):
    """Dot/dict-accessible channel parameters, e.g. ``p.bs_antenna.shape = np.array([8, 8])``."""

    DEFAULT_PARAMS = {
        c.PARAMSET_ANT_BS: {
            c.PARAMSET_ANT_SHAPE: np.array([8, 1]),
            c.PARAMSET_ANT_SPACING: 0.5,
            c.PARAMSET_ANT_ROTATION: np.array([0, 0, 0]),
            c.PARAMSET_ANT_RAD_PAT: c.PARAMSET_ANT_RAD_PAT_VALS[0],
        },
        c.PARAMSET_ANT_UE: {
            c.PARAMSET_ANT_SHAPE: np.array([1, 1]),
            c.PARAMSET_ANT_SPACING: 0.5,
            c.PARAMSET_ANT_ROTATION: np.array([0, 0, 0]),
            c.PARAMSET_ANT_RAD_PAT: c.PARAMSET_ANT_RAD_PAT_VALS[0],
        },
        c.PARAMSET_DOPPLER_EN: 0,
        c.PARAMSET_POLAR_EN: 0,
        c.PARAMSET_NUM_PATHS: c.MAX_PATHS,
        c.PARAMSET_FD_CH: 1,
        c.PARAMSET_OFDM: {
            c.PARAMSET_OFDM_SC_NUM: 512,
            c.PARAMSET_OFDM_SC_SAMP: np.arange(1),
            c.PARAMSET_OFDM_BANDWIDTH: 10e6,
            c.PARAMSET_OFDM_LPF: 0,
        },
    }

    def __init__(self, data: Optional[Dict] = None):
        super().__init__(deepcopy(self.DEFAULT_PARAMS))
        if data is not None:
            self.update(data)

    def validate(self, n_ues: int) -> "ChannelGenParameters":
        """Same checks, messages and exception type (AssertionError) as channel.py:78-139."""
        extra = compare_two_dicts(self, ChannelGenParameters())
        if len(extra):
            print("The following parameters seem unnecessary:")
            print(extra)

        bs, ue = self[c.PARAMSET_ANT_BS], self[c.PARAMSET_ANT_UE]
        if c.PARAMSET_ANT_ROTATION in bs.keys():
            shp = np.shape(bs[c.PARAMSET_ANT_ROTATION])
            assert len(shp) == 1 and shp[0] == 3, "The BS antenna rotation must be a 3D vector"
        else:
            bs[c.PARAMSET_ANT_ROTATION] = None

        if c.PARAMSET_ANT_ROTATION in ue.keys() and ue[c.PARAMSET_ANT_ROTATION] is not None:
            shp = np.shape(ue[c.PARAMSET_ANT_ROTATION])
            ok = ((len(shp) == 1 and shp[0] == 3) or (len(shp) == 2 and shp[0] == 3 and shp[1] == 2) or
                  (len(shp) >= 1 and shp[0] == n_ues))
            assert ok, ("The UE antenna rotation must either be a 3D vector for "
                        "constant values or 3 x 2 matrix for random values")
        else:
            ue[c.PARAMSET_ANT_ROTATION] = np.array([0, 0, 0])

        for side, name in ((bs, "BS"), (ue, "UE")):
            if c.PARAMSET_ANT_RAD_PAT in side.keys() and side[c.PARAMSET_ANT_ROTATION] is not None:
                assert side[c.PARAMSET_ANT_RAD_PAT] in c.PARAMSET_ANT_RAD_PAT_VALS, (
                    f"The {name} antenna radiation pattern must have one of the following values: "
                    f"{str(c.PARAMSET_ANT_RAD_PAT_VALS)}")
            else:
                side[c.PARAMSET_ANT_RAD_PAT] = c.PARAMSET_ANT_RAD_PAT_VALS[0]
        return self
