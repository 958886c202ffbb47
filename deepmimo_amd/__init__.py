"""deepmimo_amd - MI355X-native channel generation behind DeepMIMO's API.

Drop-in for ONE path of jmoraispk/DeepMIMO: ``dm.generate()`` / ``Dataset.compute_channels()`` /
``Dataset.channel`` (deepmimo/__init__.py:7-62 lists the reference's full public surface; the
names kept here are the ones that path needs).  The arithmetic runs in hand-written HIP kernels
for gfx950 behind a C-ABI shared library (include/deepmimo_amd.h); this package is the Python
host side that mirrors the reference's objects.

    import deepmimo_amd as dm
    ds = dm.Dataset({...float32 [n_ue, n_paths] ray matrices...})
    p = dm.ChannelGenParameters(); p.bs_antenna.shape = np.array([8, 8])
    H = ds.compute_channels(p)          # complex64 [n_ue, M_rx, M_tx, K]
"""
from . import consts
from .config import config
from .general_utils import DotDict
from .channel import ChannelGenParameters
from .dataset import Dataset, MacroDataset
from .geometry import steering_vec
from .core import generate, load
from .generator_utils import get_idxs_with_limits, LinearPath

__version__ = consts.VERSION

__all__ = ["generate", "load", "Dataset", "MacroDataset", "ChannelGenParameters", "config", "steering_vec",
           "get_idxs_with_limits", "LinearPath", "DotDict", "consts"]
