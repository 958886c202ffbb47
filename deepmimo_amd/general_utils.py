"""Container utility mirrored from the reference's API shape: ``DotDict``
(deepmimo/general_utils.py:124-261) - a mapping with attribute access whose nested dicts are
DotDicts too.  Behaviour (not code) follows the reference: item/attr get/set, ``update``,
``get``, ``keys/values/items``, ``to_dict``, ``deepcopy`` (copies ndarrays), AttributeError on a
missing attribute."""
from __future__ import annotations

from collections.abc import Mapping
from pprint import pformat
from typing import Any, Dict, Optional

import numpy as np


def _wrap(v):
    return DotDict(v) if isinstance(v, dict) and not isinstance(v, DotDict) else v


class DotDict(Mapping):
    def __init__(self, data: Optional[Dict[str, Any]] = None):
        object.__setattr__(self, "_data", {})
        for k, v in (data or {}).items():
            self._data[k] = _wrap(v)

    # attribute protocol ------------------------------------------------------------
    def __getattr__(self, key: str) -> Any:
        if key == "_data" or key.startswith("__"):       # an instance pickle has created but not yet filled, protocol probes
            raise AttributeError(key)
        try:
            return self._data[key]
        except KeyError:
            raise AttributeError(key) from None

    def __getstate__(self):
        return {"_data": self._data}

    def __setstate__(self, state):
        object.__setattr__(self, "_data", state["_data"])

    def __setattr__(self, key: str, value: Any) -> None:
        if key == "_data":
            object.__setattr__(self, key, value)
        else:
            self[key] = value

    # mapping protocol --------------------------------------------------------------
    def __getitem__(self, key: str) -> Any:
        return self._data[key]

    def __setitem__(self, key: str, value: Any) -> None:
        self._data[key] = _wrap(value)

    def __delitem__(self, key: str) -> None:
        del self._data[key]

    def __len__(self) -> int:
        return len(self._data)

    def __iter__(self):
        return iter(self._data)

    def __dir__(self):
        return sorted(set(list(super().__dir__()) + list(self._data.keys())))

    def keys(self):
        return self._data.keys()

    def values(self):
        return self._data.values()

    def items(self):
        return self._data.items()

    def get(self, key: str, default: Any = None) -> Any:
        return self._data.get(key, default)

    def update(self, other: Dict[str, Any]) -> None:
        for k, v in other.items():
            self._data[k] = _wrap(v)

    def to_dict(self) -> Dict:
        return {k: (v.to_dict() if isinstance(v, DotDict) else v) for k, v in self._data.items()}

    def deepcopy(self):
        out = {}
        for k, v in self._data.items():
            if isinstance(v, DotDict):
                out[k] = v.deepcopy()
            elif isinstance(v, np.ndarray):
                out[k] = v.copy()
            else:
                out[k] = v
        return type(self)(out)

    def __repr__(self) -> str:
        return pformat(self._data)


def compare_two_dicts(d1: Mapping, d2: Mapping) -> set:
    """Keys present in d1 (recursively) that d2 lacks - what validate() reports as 'unnecessary'
    (general_utils.py compare_two_dicts as used at channel.py:94)."""
    extra = set()
    for k in d1.keys():
        if k not in d2.keys():
            extra.add(k)
        elif isinstance(d1[k], Mapping) and isinstance(d2[k], Mapping):
            extra |= compare_two_dicts(d1[k], d2[k])
    return extra
