"""Global configuration singleton, API-compatible with deepmimo/config.py:36-165:
``config('use_gpu', True)``, ``config('use_gpu')``, ``config(use_gpu=True)``, ``config()``
(print), ``config.get/set/reset/get_all``.

The reference declares ``use_gpu`` / ``gpu_device_id`` (config.py:58-59) but nothing reads them;
here they select the device the MI355X kernels run on.  This package has no CPU channel path:
``use_gpu`` defaults to True and switching it off makes ``compute_channels`` raise.
Extra key: ``channel_output`` = 'numpy' (reference-compatible return type) or 'torch' (keep the
complex64 tensor resident in HBM - required when it does not fit host memory); ``strict_reference_cache``
reproduces the reference's cache staleness when only a radiation pattern changes (Dataset._params_for_prep);
``host_copy_guard`` makes ``compute_channels`` raise instead of copying a tensor larger than the free host memory;
``adaptive_precision`` opts into the one-term rule for weak path groups (default off: three product terms for every
path, the arithmetic closest to the reference's complex128 sum, channel.py:283-284)."""
from __future__ import annotations

from typing import Any


class _Config:
    _DEFAULTS = {
        "use_gpu": True,
        "gpu_device_id": 0,
        "scenarios_folder": "deepmimo_scenarios",
        "channel_output": "numpy",
        "fd_kernel_variant": 0,      # 0 auto, 1 fp32 vector kernel, 2 split-precision MFMA kernel, 9 small-output kernel, 12 folded
        "strict_reference_cache": False,   # True: keep the reference's stale `_power_linear_ant_gain` (dataset.py:213-220)
        "host_copy_guard": True,     # refuse a NumPy copy of the channel tensor that exceeds the free host memory
        "array_response_max_bytes": 2 << 30,   # largest `array_response_product` ([N, M_rx, M_tx, L] complex128) built on request
        "adaptive_precision": False,  # True: DMX_FLAG_ADAPTIVE_TERMS - a user's weak last path group in ONE f16 product term
                                      # (<= 7.6e-6 of the strongest path instead of ~2e-6; 3-5 % faster at 25 paths)
    }

    def __init__(self):
        self._cfg = dict(self._DEFAULTS)

    def set(self, key: str, value: Any) -> None:
        if key not in self._cfg:
            print(f"Warning: Configuration key '{key}' does not exist. Adding as new key.")
        self._cfg[key] = value

    def get(self, key: str, default: Any = None) -> Any:
        return self._cfg.get(key, default)

    def reset(self) -> None:
        self._cfg = dict(self._DEFAULTS)

    def get_all(self) -> dict:
        return dict(self._cfg)

    def print_config(self) -> None:
        print("\nDeepMIMO Configuration:")
        print("-" * 50)
        for k, v in self._cfg.items():
            print(f"{k}: {v}")
        print("-" * 50)

    def __call__(self, *args, **kwargs):
        if not args and not kwargs:
            self.print_config()
            return None
        if len(args) == 1 and not kwargs:
            return self.get(args[0])
        if len(args) == 2 and not kwargs:
            self.set(args[0], args[1])
            return None
        if kwargs and not args:
            for k, v in kwargs.items():
                self.set(k, v)
            return None
        raise ValueError("Invalid arguments. Use config(), config('key'), config('key', value) or config(key=value).")


config = _Config()
