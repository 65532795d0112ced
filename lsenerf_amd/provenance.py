"""Which kernel sources a committed counter summary was measured on.

``profiles/pmc_traffic.json`` holds numbers that ``bench.py`` cannot measure inside its own run (rocprofv3 ``--pmc`` passes:
HBM-side traffic of the hash kernels, matrix-core busy cycles of the fused MLPs).  Each group of numbers is stamped with a
digest of the source files its kernels are compiled from; ``bench.py`` re-computes the digests from the tree it runs in and
reports the numbers only when they match -- a kernel change that forgot to regenerate the file yields ``null`` fields and
``"traffic_stale": true`` instead of stale counters under a fresh timing.  (Content digests, not git objects: the GPU box gets a
snapshot without ``.git``.)  No torch, no GPU."""
from __future__ import annotations

import hashlib
import os
from typing import Dict, Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
_INCLUDE = os.path.join(os.path.dirname(_HERE), "include")

# counter group -> the sources its kernels are built from (common.h and the C-ABI header are part of every translation unit)
GROUPS = {
    "hash": ("hashgrid.hip", "common.h", "lse_hip.h"),
    "mlp": ("mlp.hip", "mlp_x6.h", "common.h", "lse_hip.h"),
}


def _path(name: str) -> str:
    return os.path.join(_INCLUDE if name == "lse_hip.h" else _CSRC, name)


def source_digest(group: str) -> str:
    """sha256 over (file name, content) of the group's sources, first 16 hex digits."""
    h = hashlib.sha256()
    for name in GROUPS[group]:
        h.update(name.encode() + b"\0")
        with open(_path(name), "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    return h.hexdigest()[:16]


def source_digests() -> Dict[str, str]:
    return {g: source_digest(g) for g in GROUPS}


def counters_current(summary: dict, group: str) -> bool:
    """True iff ``summary`` (the parsed profiles/pmc_traffic.json) says its ``group`` numbers were measured on this tree."""
    stamp: Optional[str] = (summary.get("source_digests") or {}).get(group)
    return stamp is not None and stamp == source_digest(group)
