"""Sionna-layout export of time-domain channels (reference:
deepmimo/integrations/sionna_adapter.py:22-200, which still reads the v3 dict layout).

Pure reshaping on the consumer side: a generator that yields ``(a, tau)`` samples with
``a``  complex64 ``[num_rx, num_rx_ant, num_tx, num_tx_ant, num_paths, num_time_steps=1]`` and
``tau`` float32 ``[num_rx, num_tx, num_paths]`` from the time-domain channels
(``freq_domain = 0``; valid paths compacted to the front - exactly what ``dmx_channels_td`` writes) and the
path delays of one or several ``Dataset`` objects (one per basestation).  Index conventions
(``bs_idx`` / ``ue_idx`` as int, vector or 2-D matrix of samples x elements) follow the reference adapter.
"""
from __future__ import annotations

from typing import Iterator, List, Tuple

import numpy as np

from . import consts as c


def _as_index_matrix(idx) -> np.ndarray:
    """sionna_adapter.py:101-166: int -> [[i]], vector -> column, matrix stays."""
    if isinstance(idx, (int, np.integer)):
        idx = np.array([[int(idx)]])
    elif isinstance(idx, (list, range)):
        idx = np.array(idx)
    elif not isinstance(idx, np.ndarray):
        raise TypeError("The index input type must be an integer, list, or numpy array!")
    if idx.ndim == 1:
        idx = idx.reshape((-1, 1))
    elif idx.ndim != 2:
        raise ValueError("The index input must be integer, vector or 2D matrix!")
    return idx


class DeepMIMOSionnaAdapter:
    """``adapter = DeepMIMOSionnaAdapter(dataset_or_list, bs_idx, ue_idx); for a, tau in adapter(): ...``

    dataset: a ``Dataset``, a ``MacroDataset`` or a list of ``Dataset`` (one per basestation) whose ``channel``
    holds time-domain channels [n_ue, M_rx, M_tx, P] (NumPy or torch)."""

    def __init__(self, dataset, bs_idx=None, ue_idx=None) -> None:
        if hasattr(dataset, "datasets"):
            dataset = dataset.datasets
        self.datasets: List = list(dataset) if isinstance(dataset, (list, tuple)) else [dataset]
        self._ch = [self._host(d[c.CHANNEL_PARAM_NAME]) for d in self.datasets]
        for d in self.datasets:
            if int(d[c.CH_PARAMS_PARAM_NAME][c.PARAMSET_FD_CH]) != 0:
                raise ValueError("the Sionna export needs time-domain channels: compute with params.freq_domain = 0")
        self.bs_idx = _as_index_matrix(np.array([[0]]) if bs_idx is None else bs_idx)
        self.ue_idx = _as_index_matrix(np.arange(self._ch[0].shape[0]) if ue_idx is None else ue_idx)
        self.num_rx_ant, self.num_tx_ant, self.num_paths = self._ch[0].shape[1:4]
        self.num_samples_bs, self.num_tx = self.bs_idx.shape
        self.num_samples_ue, self.num_rx = self.ue_idx.shape
        self.num_samples = self.num_samples_bs * self.num_samples_ue
        self.num_time_steps = 1
        self.ch_shape = (self.num_rx, self.num_rx_ant, self.num_tx, self.num_tx_ant, self.num_paths, 1)
        self.t_shape = (self.num_rx, self.num_tx, self.num_paths)

    @staticmethod
    def _host(x) -> np.ndarray:
        return x.detach().cpu().numpy() if hasattr(x, "detach") else np.asarray(x)

    def _delays(self, i_bs: int, i_ue: int) -> np.ndarray:
        """ToA of the user's valid paths in slot order (the TD kernel compacts valid = non-NaN-power paths)."""
        d = self.datasets[i_bs]
        P = self.num_paths
        power = self._host(d[c.PWR_LINEAR_ANT_GAIN_PARAM_NAME])[i_ue, :P]
        delay = self._host(d[c.DELAY_PARAM_NAME])[i_ue, :P]
        return delay[~np.isnan(power)]

    def __len__(self) -> int:
        return self.num_samples

    def __call__(self) -> Iterator[Tuple[np.ndarray, np.ndarray]]:
        for i in range(self.num_samples_ue):
            for j in range(self.num_samples_bs):
                a = np.zeros(self.ch_shape, dtype=np.csingle)
                tau = np.zeros(self.t_shape, dtype=np.single)
                for i_ch in range(self.num_rx):
                    for j_ch in range(self.num_tx):
                        i_ue, i_bs = int(self.ue_idx[i][i_ch]), int(self.bs_idx[j][j_ch])
                        a[i_ch, :, j_ch, :, :, 0] = self._ch[i_bs][i_ue]
                        toa = self._delays(i_bs, i_ue)
                        tau[i_ch, j_ch, :len(toa)] = toa
                yield a, tau
