"""ctypes binding of the C-ABI in include/lse_hip.h (liblse_hip.so, gfx950).

There is NO fallback: if the shared library is missing or a call fails, this raises.  The CPU oracle under
``oracle/`` is test infrastructure and is never imported from here.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int32, c_int64, c_uint32, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LSE_HIP_LIB", os.path.join(_HERE, "liblse_hip.so"))   # override: A/B builds of the same ABI
# the development build (make -C lsenerf_amd/csrc dev): same ABI + the tuning knobs of csrc/dev_knobs.h and the superseded kernel
# variants.  Never loaded by the product path: tools/ and the variant tests switch to it explicitly (dev_library()).
DEV_LIB_PATH = os.environ.get("LSE_HIP_DEV_LIB", os.path.join(_HERE, "liblse_hip_dev.so"))

LSE_MAX_GRID_LEVELS = 32
LSE_MAX_OCC_LEVELS = 8
LSE_IN_ROWMAJOR, LSE_IN_LEVELMAJOR = 0, 1
LSE_ACT_NONE, LSE_ACT_SIGMOID = 0, 1
LSE_ABI_VERSION = 6
LSE_MLP_ARITH_AUTO, LSE_MLP_ARITH_F32_MFMA = 0, 1
LSE_TRAVERSE_FMA_SETUP = 1


class GridDesc(Structure):
    _fields_ = [("n_levels", c_int32), ("n_features", c_int32), ("offsets", c_uint32 * (LSE_MAX_GRID_LEVELS + 1)),
                ("scales", c_float * LSE_MAX_GRID_LEVELS), ("resolutions", c_uint32 * LSE_MAX_GRID_LEVELS)]


class HashBwdOpts(Structure):
    _fields_ = [("impl", c_int32), ("gran", c_int32), ("few_runs", c_int32), ("second_probe", c_int32), ("rounds", c_int32),
                ("dbg", c_int32), ("interleave_from_scale", c_float), ("stage_max", c_int32), ("coarse_levels", c_int32),
                ("replicas", c_int32), ("replica_levels", c_int32), ("workspace", c_void_p), ("workspace_bytes", c_int64),
                ("prefetch", c_int32)]


class EpilogueDesc(Structure):
    _fields_ = [("rgb_mapped", c_int32), ("rgb_mapper", c_int32), ("evs_mapper", c_int32), ("ev_one_dim", c_int32),
                ("deblur_group", c_int32), ("evs_loss_weight", c_float), ("event_loss_kind", c_int32)]


LSE_MAP_IDENTITY, LSE_MAP_GT, LSE_MAP_POWPOW, LSE_MAP_MLP, LSE_MAP_RGB_MLP = 1, 2, 3, 4, 5


class MapperMlp(Structure):
    """lse_mapper_mlp: the four nn.Linear layers of an MLP intensity mapper (device pointers) and where their gradients go."""
    _fields_ = [("w", c_void_p * 4), ("b", c_void_p * 4), ("dw", c_void_p * 4), ("db", c_void_p * 4)]


LSE_ONE_DIM_NONE, LSE_ONE_DIM_LEARNED, LSE_ONE_DIM_GRAY = 0, 1, 2
LSE_EVLOSS_LOG, LSE_EVLOSS_ENERF_NORM = 0, 1


class MlpDesc(Structure):
    _fields_ = [("n_in", c_int32), ("width", c_int32), ("n_hidden_layers", c_int32), ("out_activation", c_int32),
                ("in_layout", c_int32), ("w0_ld", c_int32), ("w0_col", c_int32), ("w0_mask_col0", c_int32), ("arith", c_int32)]


P = c_void_p
I32, I64, F32 = c_int32, c_int64, c_float

# name -> argtypes (all return int except the two misc functions); mirrors include/lse_hip.h one to one
SIGNATURES = {
    "lse_traverse_grids": [P, P, I32, P, P, I32, I32, I32, I32, P, P, F32, F32, I32, P, P, P, P, P, I32, P],
    "lse_traverse_grids_slots": [P, P, I32, P, P, I32, I32, I32, I32, P, P, F32, F32, I64, P, P, P, P, I32, P],
    "lse_compact_ray_slots": [P, P, I64, P, I32, P, P, P, P],
    "lse_pack_info_from_counts": [P, I32, P, P, P],
    "lse_fake_sample_if_empty": [P, I32, P, P, P, P, P, P, P, I64, I32, I32, P],
    "lse_ray_planes": [F32, F32, P, P, P, F32, I32, P, P, P],
    "lse_visibility_mask": [P, P, P, P, I32, F32, F32, P, P, P],
    "lse_visibility_mask_alpha": [P, P, I32, F32, F32, P, P, P],
    "lse_visibility_mask_cap": [P, P, P, P, I32, F32, F32, P, P, P, P],
    "lse_compact_samples": [P, P, P, I32, P, P, P, P, P, P, P],
    "lse_compact_features": [P, P, P, I32, P, P, P, I32, I64, I64, P, P, P, P],
    "lse_positions_fwd": [P, P, P, P, P, I64, P, I32, P, P, P, P],
    "lse_positions_bwd": [P, P, P, P, P, I64, P, I32, P, P, P, P],
    "lse_ray_grad_reduce": [P, P, P, P, I32, P, P, P],
    "lse_hash_fwd": [POINTER(GridDesc), P, P, P, I64, P, P],
    "lse_hash_bwd": [POINTER(GridDesc), P, P, P, P, P, I64, P, P],
    "lse_hash_bwd_levels": [POINTER(GridDesc), P, P, P, P, P, I32, I32, I32, I64, P, P],
    "lse_hash_bwd_ex": [POINTER(GridDesc), P, P, P, P, P, I32, I32, I32, I64, P, POINTER(HashBwdOpts), P],
    "lse_mlp_fwd": [POINTER(MlpDesc), P, P, P, P, P, I32, P, I32, P, P, F32, I64, P, P],
    "lse_mlp_bwd": [POINTER(MlpDesc), P, P, P, I32, P, I32, P, P, P, F32, P, P, P, P, P, P, P, P, I64, P, P],
    "lse_mlp_wgrad": [POINTER(MlpDesc), P, P, P, P, P, I64, P],
    "lse_segment_sum_rows": [P, I32, P, I32, P, P],
    "lse_ray_features_fwd": [P, P, P, I32, I32, P, P],
    "lse_ray_features_bwd": [P, P, P, I32, I32, I32, P, P, P],
    "lse_ray_bias_fwd": [P, P, P, I32, I32, P, I32, I32, P, P, P],
    "lse_ray_bias_bwd": [P, P, I32, I32, I32, P, I32, I32, P, P, P, P, P],
    "lse_linear_fwd": [P, P, I32, I32, I32, P, P],
    "lse_linear_bwd_input": [P, P, I32, I32, I32, P, P],
    "lse_gemm_tn_acc": [P, I32, P, I32, I32, I64, P, I32, P],
    "lse_density_fwd": [P, P, F32, P, I64, P],
    "lse_density_bwd": [P, P, F32, P, P, I64, P],
    "lse_volrend_fwd": [P, P, P, P, I32, P, I32, P, P, P, P, P],
    "lse_volrend_bwd": [P, P, P, P, I32, P, I32, P, P, P, P, P, P, P],
    "lse_volrend_depth_fwd": [P, P, P, P, I32, P, I32, P, P, P, P, P, P, P, P],
    "lse_render_weight_fwd": [P, P, P, P, I32, P, P, P, P],
    "lse_render_weight_bwd": [P, P, P, P, I32, P, P, P, P],
    "lse_loss_epilogue_fwd": [POINTER(EpilogueDesc), P, P, I32, P, P, P, P, I32, P, P, P, POINTER(MapperMlp), POINTER(MapperMlp), P, P],
    "lse_loss_epilogue_bwd": [POINTER(EpilogueDesc), P, P, I32, P, P, P, P, I32, P, P, P, POINTER(MapperMlp), POINTER(MapperMlp),
                              P, P, P, P, P, P, P],
    "lse_occ_update_cells": [P, P, P, I64, F32, P, P],
    "lse_occ_binarize": [P, I64, P, P, P],
    "lse_adam_step": [P, P, P, P, I64, F32, F32, F32, F32, I32, F32, P],
    "lse_adam_step_dev": [P, P, P, P, I64, P, F32, P],
    "lse_adam_schedule_dev": [P, P, P, P],
}
# exported by the development build only (csrc/dev_knobs.h)
DEV_SIGNATURES = {
    "lse_set_option": [c_char_p, I64],
    "lse_get_option": [c_char_p, POINTER(c_int64)],
}

_lib = None
_prod_lib = None       # the library that ships, while dev_library() has swapped the development build in
_dev_lib = None


class LseHipError(RuntimeError):
    pass


def _open(path: str, what: str, extra: dict):
    if not os.path.exists(path):
        raise LseHipError(
            f"{path} not found: {what} has not been built (run `python -c 'import __graft_entry__ as g; g.build()'` or "
            f"`make -C lsenerf_amd/csrc all dev`).  There is no CPU fallback for the product path.")
    lib = ctypes.CDLL(path)
    lib.lse_last_error.restype = c_char_p
    lib.lse_last_error.argtypes = []
    lib.lse_abi_version.restype = c_int32
    lib.lse_abi_version.argtypes = []
    for name, argtypes in {**SIGNATURES, **extra}.items():
        fn = getattr(lib, name)       # AttributeError here == header/library mismatch: fail loudly
        fn.argtypes = argtypes
        fn.restype = c_int32
    lib.lse_hash_bwd_default_opts.restype = None
    lib.lse_hash_bwd_default_opts.argtypes = [POINTER(HashBwdOpts)]
    lib.lse_hash_bwd_workspace_bytes.restype = c_int64
    lib.lse_hash_bwd_workspace_bytes.argtypes = [POINTER(GridDesc), POINTER(HashBwdOpts)]
    v = lib.lse_abi_version()
    if v != LSE_ABI_VERSION:
        raise LseHipError(f"{os.path.basename(path)} ABI version {v} != binding version {LSE_ABI_VERSION}")
    return lib


def load():
    """Load liblse_hip.so (built by ``__graft_entry__.build()`` / ``make -C lsenerf_amd/csrc``).  ``LSE_DEV=1`` in the environment
    makes the development build the process's library from the start (A/B tools; ``LSE_OPT_<NAME>=<int>`` then seeds its knobs)."""
    global _lib
    if _lib is not None:
        return _lib
    if os.environ.get("LSE_DEV", "0") not in ("", "0"):
        _lib = _load_dev()
        for k, val in os.environ.items():
            if k.startswith("LSE_OPT_"):
                if _lib.lse_set_option(k[len("LSE_OPT_"):].lower().encode(), int(val)) != 0:
                    raise LseHipError(f"{k}: {_lib.lse_last_error().decode()}")
        return _lib
    _lib = _open(LIB_PATH, "the HIP extension", {})
    return _lib


def _load_dev():
    global _dev_lib
    if _dev_lib is None:
        _dev_lib = _open(DEV_LIB_PATH, "the development build of the HIP extension", DEV_SIGNATURES)
    return _dev_lib


def dev_available() -> bool:
    return os.path.exists(DEV_LIB_PATH)


class dev_library:
    """``with _lib.dev_library(): ...`` -- every C-ABI call inside the block goes to the DEVELOPMENT build (liblse_hip_dev.so: same
    entry points + the tuning knobs and superseded kernel variants of csrc/dev_knobs.h).  For tools and variant tests only; knobs
    changed inside the block are restored on exit."""

    def __enter__(self):
        global _lib, _prod_lib
        load()
        self._outer = _lib
        dev = _load_dev()
        self._saved = {}
        _prod_lib, _lib = (_prod_lib or self._outer), dev
        return self

    def __exit__(self, *exc):
        global _lib
        for name, v in self._saved.items():
            _load_dev().lse_set_option(name.encode(), v)
        _lib = self._outer
        return False

    def set_option(self, name: str, value: int) -> None:
        if name not in self._saved:
            self._saved[name] = get_option(name)
        set_option(name, value)


# bench.py sets this to {"names": set or None (= every entry point), "events": []}: those entry points are then bracketed by HIP events
# recorded on torch's current stream (the stream every kernel is launched on).
TIMING = None


_TIMING_ALIAS = {"lse_hash_bwd_ex": "lse_hash_bwd", "lse_hash_bwd_levels": "lse_hash_bwd"}   # one operation, three entry points


def call(name: str, *args):
    lib = load()
    t = TIMING
    tname = _TIMING_ALIAS.get(name, name)
    if t is not None and (t["names"] is None or tname in t["names"]):
        import torch
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = getattr(lib, name)(*args)
        e1.record()
        t["events"].append((tname, e0, e1))
    else:
        rc = getattr(lib, name)(*args)
    if rc != 0:
        raise LseHipError(f"{name} failed (rc={rc}): {lib.lse_last_error().decode()}")


def _dev_active():
    lib = load()
    if not hasattr(lib, "lse_set_option") or lib is not _dev_lib:
        raise LseHipError("tuning knobs exist in the development build only (csrc/dev_knobs.h): use `with _lib.dev_library():` "
                          "or LSE_DEV=1; liblse_hip.so has no state to set")
    return lib


def set_option(name: str, value: int) -> None:
    """Tuning knob of the DEVELOPMENT build (csrc/dev_knobs.h); speed only, never results."""
    _dev_active()
    call("lse_set_option", name.encode(), int(value))


def get_option(name: str) -> int:
    _dev_active()
    v = c_int64(0)
    call("lse_get_option", name.encode(), ctypes.byref(v))
    return int(v.value)


def hash_bwd_default_opts() -> HashBwdOpts:
    o = HashBwdOpts()
    load().lse_hash_bwd_default_opts(ctypes.byref(o))
    return o
