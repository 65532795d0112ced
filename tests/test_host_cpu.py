"""CPU tests of the host side: the C-ABI library loads and exports every symbol include/lse_hip.h declares, descriptor
structs match the header, host logic (level tables, flat parameter buffers, config coercions, module wiring, error
behaviour) works without a GPU.  No compute calls are made here."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "lse_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(lse_[a-z0-9_]+)\s*\(", src)))


def _exported(path):
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    return {ln.split()[-1] for ln in out.splitlines() if " T " in ln and ln.split()[-1].startswith("lse_")}


def test_library_exports_every_declared_symbol():
    from lsenerf_amd import _lib
    lib = _lib.load()
    names = header_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"liblse_hip.so does not export {n}"
    # and the binding covers exactly the header (no stale or missing signatures)
    assert set(_lib.SIGNATURES) | {"lse_abi_version", "lse_last_error", "lse_hash_bwd_default_opts", "lse_hash_bwd_workspace_bytes"} == set(names)
    assert lib.lse_abi_version() == _lib.LSE_ABI_VERSION == 6


def test_the_shipped_library_has_no_state_to_set():
    """ABI 5 (include/lse_hip.h, "NO GLOBAL OR THREAD-LOCAL STATE"): the exported symbols of liblse_hip.so are exactly the header's
    functions -- no lse_set_device_count (the device-side count is the `n_dev` argument of the per-sample entry points), no
    lse_set_option / lse_get_option (tuning knobs are constants; the development build liblse_hip_dev.so, csrc/dev_knobs.h, adds
    exactly those two).  The header says so in as many words."""
    from lsenerf_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "lse_hip.h")).read()
    assert "NO GLOBAL OR THREAD-LOCAL STATE" in hdr
    for gone in ("lse_set_device_count", "lse_set_option", "lse_get_option"):
        assert gone + "(" not in re.sub(r"/\*.*?\*/", "", hdr, flags=re.S), gone
    shipped = _exported(_lib.LIB_PATH)
    assert shipped == set(header_functions()), shipped ^ set(header_functions())
    assert _lib.dev_available(), "liblse_hip_dev.so is built by __graft_entry__.build() (make -C lsenerf_amd/csrc dev)"
    assert _exported(_lib.DEV_LIB_PATH) == shipped | {"lse_set_option", "lse_get_option"}
    # the per-sample entry points carry the count themselves
    src = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    for name in ("lse_positions_fwd", "lse_positions_bwd", "lse_hash_fwd", "lse_hash_bwd", "lse_hash_bwd_levels", "lse_hash_bwd_ex",
                 "lse_mlp_fwd", "lse_mlp_bwd"):
        m = re.search(r"\b" + name + r"\s*\((.*?)\)\s*;", src, flags=re.S)
        assert m and "const int64_t *n_dev" in m.group(1), name
    with pytest.raises(_lib.LseHipError, match="development build"):
        _lib.set_option("mlp_fwd_cfg", 44)


def test_dev_knobs_and_hash_bwd_opts_without_gpu():
    """Kernel selections are call arguments (lse_hash_bwd_ex's options, lse_mlp_desc.arith, traversal flags); the development
    build's knobs are writable inside ``_lib.dev_library()`` only and restored when the block ends."""
    from lsenerf_amd import _lib
    o = _lib.hash_bwd_default_opts()
    assert (o.impl, o.gran, o.few_runs, o.second_probe, o.rounds, o.dbg, o.stage_max, o.coarse_levels) == (2, 6, 8, 3, 32, 0, 48, 0)
    assert (o.replicas, o.replica_levels, o.prefetch, o.workspace) == (16, 4, 0, None)
    with _lib.dev_library() as dev:
        assert _lib.get_option("hash_fwd_mapping") == 4 and _lib.get_option("mlp_bwd_cfg") == 28 and _lib.get_option("traverse_vec") == 1
        dev.set_option("mlp_fwd_cfg", 44)
        assert _lib.get_option("mlp_fwd_cfg") == 44
        dev.set_option("hash_bwd_few_runs", 9)
        assert _lib.hash_bwd_default_opts().few_runs == 9           # (the dev build's defaults follow its knobs)
        with pytest.raises(_lib.LseHipError):
            _lib.set_option("no_such_option", 1)
    with _lib.dev_library():
        assert _lib.get_option("mlp_fwd_cfg") == 28 and _lib.get_option("hash_bwd_few_runs") == 8      # restored on exit
    assert _lib.hash_bwd_default_opts().few_runs == 8
    # invalid kernel selections are rejected before anything is launched; development variants are not in the shipped library
    d = _lib.GridDesc()
    d.n_levels, d.n_features = 1, 2
    d.offsets[0], d.offsets[1], d.scales[0], d.resolutions[0] = 0, 8, 1.0, 2
    o.impl = 7
    with pytest.raises(_lib.LseHipError):
        _lib.call("lse_hash_bwd_ex", ctypes.byref(d), None, None, None, None, None, 0, 0, 1, 0, None, ctypes.byref(o), None)
    o.impl = 1
    with pytest.raises(_lib.LseHipError, match="development variant"):
        _lib.call("lse_hash_bwd_ex", ctypes.byref(d), None, None, None, None, None, 0, 0, 1, 0, None, ctypes.byref(o), None)


def test_argument_counts_match_header():
    from lsenerf_amd import _lib
    src = open(os.path.join(ROOT, "include", "lse_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    for name, argtypes in _lib.SIGNATURES.items():
        m = re.search(r"\b" + name + r"\s*\((.*?)\)\s*;", src, flags=re.S)
        assert m, name
        n_args = len([a for a in m.group(1).split(",") if a.strip()])
        assert n_args == len(argtypes), (name, n_args, len(argtypes))


def test_descriptor_struct_layouts():
    from lsenerf_amd import _lib
    assert ctypes.sizeof(_lib.GridDesc) == 4 + 4 + 33 * 4 + 32 * 4 + 32 * 4
    assert ctypes.sizeof(_lib.MlpDesc) == 36


def test_invalid_arguments_fail_loudly_without_gpu():
    """Argument validation happens before any launch, so these error paths are exercised on CPU."""
    from lsenerf_amd import _lib
    lib = _lib.load()
    d = _lib.MlpDesc(24, 64, 1, 0, 0)
    rc = lib.lse_mlp_fwd(ctypes.byref(d), None, None, None, None, None, 16, None, 0, None, None, 0.0, 8, None, None)
    assert rc == -1 and b"n_in" in lib.lse_last_error()
    g = _lib.GridDesc()
    g.n_levels, g.n_features = 16, 4
    rc = lib.lse_hash_fwd(ctypes.byref(g), None, None, None, 8, None, None)
    assert rc == -1 and b"n_features" in lib.lse_last_error()
    with pytest.raises(_lib.LseHipError):
        _lib.call("lse_traverse_grids", None, None, 4, None, None, 1, 8, 8, 8, None, None, 0.1, 0.0, 0, None, None, None,
                  None, None, 0, None)
    # zero-sized work is a no-op, not an error
    assert lib.lse_volrend_fwd(None, None, None, None, 0, None, 0, None, None, None, None, None) == 0


def test_missing_library_is_an_error_not_a_fallback(monkeypatch):
    from lsenerf_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/liblse_hip.so")
    with pytest.raises(_lib.LseHipError, match="no CPU fallback"):
        _lib.load()


def test_cpu_tensors_are_rejected():
    from lsenerf_amd import _lib, ops
    with pytest.raises(_lib.LseHipError, match="no CPU fallback"):
        ops.hash_encode(torch.rand(4, 3), torch.rand(16), ops.make_grid_meta(n_levels=4, log2_hashmap_size=4))


def test_grid_meta_matches_oracle_restatement():
    from lsenerf_amd import ops
    from oracle import hashgrid as hg
    for kw in (dict(), dict(n_levels=4, log2_hashmap_size=12), dict(n_levels=16, log2_hashmap_size=14, max_res=1024)):
        a, b = ops.make_grid_meta(**kw), hg.tcnn_grid_meta(**kw)
        assert list(a.offsets) == b.offsets and list(a.resolutions) == b.resolutions and list(a.scales) == b.scales
    m = ops.make_grid_meta()
    assert m.n_params == 12196240
    d = m.desc()
    assert d.n_levels == 16 and d.offsets[16] == 6098120 and d.resolutions[0] == 16


def test_model_wiring_and_param_groups():
    import lsenerf_amd as la
    cfg = la.LSENeRFModelConfig(evs_mapping_method="None", map_mode="None", ev_one_dim="True", rgb_loss_type="None")
    assert cfg.evs_mapping_method is None and cfg.map_mode == "evs_rgb" and cfg.ev_one_dim == "learned"
    assert cfg.rgb_loss_type == "linspace"
    m = la.LSENeRFModel(la.LSENeRFModelConfig(use_mapping=True, mapping_method="identity", map_mode="co_map",
                                              evs_mapping_method="powpow"), torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), 12)
    assert abs(m.config.render_step_size - 0.0034641016) < 1e-9                      # R:lse_nerf/lsenerf.py:180-182
    assert isinstance(m.renderer_rgb, la.LinearRenderer)                              # :209-213
    assert m.occupancy_grid.binaries.shape == (4, 128, 128, 128) and m.occupancy_grid.aabbs[3].tolist() == [-8.0] * 3 + [8.0] * 3
    names = {n for n, _ in m.field.named_parameters()}
    assert names == {"mlp_base_grid.params", "mlp_base_mlp.params", "mlp_head.params",
                     "embedding_appearance.embedding.weight"}
    assert m.field.mlp_base_mlp.params.numel() == 3072 and m.field.mlp_head.params.numel() == 9216
    groups = m.get_param_groups()
    assert set(groups) == {"fields"}
    n = sum(p.numel() for p in groups["fields"])
    assert n == 12196240 + 3072 + 9216 + 32 + 3 + 1        # grid + MLPs + global embedding + ThreeToOne + Powpow
    assert not any(k.startswith("hash_table") for k in names)  # the reference's dead 67 MB table is not allocated


def test_mlp_intensity_mappers_are_identity_initialised(monkeypatch):
    """R:lse_nerf/intensity_mappers.py:8-63, 93-97: "mlp" (1 -> 16 -> 16 -> 16 -> 1) and "rgb_mlp" (3 -> ... -> 3), ReLU hidden
    layers, sigmoid output, fitted to the identity on [0, 1] at construction; selectable through mapping_method /
    evs_mapping_method like the closed-form mappers, and -- since ABI 6 -- taken by the fused epilogue wherever the reference's own
    nn.Linear accepts the channel count (a one-channel "mlp" on three channels raises there: such a plan stays on the torch route,
    which raises the same error)."""
    import lsenerf_amd as la
    from lsenerf_amd import model as M
    assert set(M.MAPPERS_DICT) == {"mlp", "rgb_mlp", "gt", "identity", "powpow"}
    torch.manual_seed(0)
    monkeypatch.setattr(M.MLP_Mapper, "init_steps", 1500)
    monkeypatch.setattr(M.RGB_MLP_Mapper, "init_steps", 1500)
    for cls, dim in ((M.MLP_Mapper, 1), (M.RGB_MLP_Mapper, 3)):
        mp = cls()
        assert {n for n, _ in mp.named_parameters()} == {f"mlp.layers.{i}.{k}" for i in range(4) for k in ("weight", "bias")}
        assert mp.mlp.layers[0].weight.shape == (16, dim) and mp.mlp.layers[3].weight.shape == (dim, 16)
        x = torch.linspace(0.05, 0.95, 37)[:, None].repeat(1, dim)
        assert float((mp(x) - x).detach().abs().max()) < 0.05                     # identity on the grey axis after the fit
    m = la.LSENeRFModel(la.LSENeRFModelConfig(use_mapping=True, mapping_method="rgb_mlp", map_mode="co_map", evs_mapping_method="mlp",
                                              ev_one_dim="True"), torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), 4)
    assert isinstance(m.rgb_mapper, M.RGB_MLP_Mapper) and isinstance(m.evs_mapper, M.MLP_Mapper)
    n_map = sum(p.numel() for p in m.rgb_mapper.parameters()) + sum(p.numel() for p in m.evs_mapper.parameters())
    assert n_map == (3 * 16 + 16 + 2 * (16 * 16 + 16) + 16 * 3 + 3) + (1 * 16 + 16 + 2 * (16 * 16 + 16) + 16 + 1)
    assert sum(p.numel() for p in m.get_param_groups()["fields"]) >= n_map        # trained with the field (R:lsenerf.py:237-243)
    from lsenerf_amd import _lib
    fields, pow_rgb, pow_evs, w31, mlp_rgb, mlp_evs = m._epilogue_desc()           # -> the fused epilogue's MLP kernels
    assert fields[:4] == (1, _lib.LSE_MAP_RGB_MLP, _lib.LSE_MAP_MLP, _lib.LSE_ONE_DIM_LEARNED) and pow_rgb is None and pow_evs is None
    assert [tuple(p.shape) for p in mlp_rgb] == [(16, 3), (16,), (16, 16), (16,), (16, 16), (16,), (3, 16), (3,)]
    assert [tuple(p.shape) for p in mlp_evs] == [(16, 1), (16,), (16, 16), (16,), (16, 16), (16,), (1, 16), (1,)]
    assert all(a is b for a, b in zip(mlp_rgb, m.rgb_mapper.parameters())) and all(a is b for a, b in zip(mlp_evs, m.evs_mapper.parameters()))
    m.train()
    routed = m.route_outputs({"rgb": torch.rand(5, 3)}, None, ev_out=True)
    assert routed["rgb"].shape == (5, 3) and routed["ev_out"].shape == (5, 1)
    # channel counts the first nn.Linear does not take: the plan keeps the torch route (and fails there like the reference)
    for bad in (dict(mapping_method="mlp", map_mode="co_map", evs_mapping_method="gt", ev_one_dim="learned"),       # mlp on rgb
                dict(mapping_method="identity", map_mode="co_map", evs_mapping_method="mlp", ev_one_dim=False),       # mlp on 3 event channels
                dict(mapping_method="identity", map_mode="co_map", evs_mapping_method="rgb_mlp", ev_one_dim="gt")):  # rgb_mlp on 1
        mb = la.LSENeRFModel(la.LSENeRFModelConfig(use_mapping=True, **bad), torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), 4)
        assert mb._epilogue_desc() is None, bad
    ok = la.LSENeRFModel(la.LSENeRFModelConfig(use_mapping=True, mapping_method="rgb_mlp", map_mode="rgb_evs", ev_one_dim=False),
                         torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), 4)
    f2 = ok._epilogue_desc()
    assert f2[0][:4] == (0, _lib.LSE_MAP_IDENTITY, _lib.LSE_MAP_RGB_MLP, _lib.LSE_ONE_DIM_NONE) and f2[4] == () and len(f2[5]) == 8


def test_flat_params_and_adam_schedule_cpu():
    from lsenerf_amd.optim import FlatAdam, FlatParams
    a, b = torch.nn.Parameter(torch.randn(5, 3)), torch.nn.Parameter(torch.randn(70))
    a0 = a.detach().clone()
    flat = FlatParams([a, b])
    assert torch.equal(a.detach(), a0) and a.data_ptr() == flat.data.data_ptr()
    (a.sum() * 2 + (b * b).sum()).backward()
    assert torch.allclose(flat.grad[:15], torch.full((15,), 2.0)) and flat.grad.data_ptr() == a.grad.data_ptr()
    assert torch.allclose(flat.grad[flat.offsets[1]:flat.offsets[1] + 70], 2 * b.detach())
    flat.zero_grad()
    assert float(flat.grad.abs().sum()) == 0 and a.grad.data_ptr() == flat.grad.data_ptr()
    opt = FlatAdam.__new__(FlatAdam)
    opt.lr_init, opt.lr_final, opt.max_steps, opt.step_count = 1e-2, 1e-4, 200000, 0
    assert abs(opt.current_lr() - 1e-2) < 1e-12
    opt.step_count = 200000
    assert abs(opt.current_lr() - 1e-4) < 1e-12
    opt.step_count = 100000
    assert abs(opt.current_lr() - 1e-3) < 1e-9


def test_embedding_modes_and_errors():
    import lsenerf_amd as la
    from lsenerf_amd.field import EvsFrameEmbedding, GlobalEmbedding
    f = la.LSEField(torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), 10, embd_config=la.LSEEmbeddingConfig("evs_emb"))
    assert isinstance(f.embedding_appearance, EvsFrameEmbedding) and f.embedding_appearance.embedding.weight.shape == (10, 32)
    g = la.LSEField(torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), 10)
    assert isinstance(g.embedding_appearance, GlobalEmbedding) and g.embedding_appearance.embedding.weight.shape == (1, 32)
    rs = la.RaySamples(la.Frustums(torch.zeros(2, 3), torch.zeros(2, 3), torch.zeros(2, 1), torch.zeros(2, 1)))
    with pytest.raises(AssertionError):
        g.get_outputs(rs, None)                                    # R:lse_nerf/lse_field.py:293
    with pytest.raises(AttributeError, match="Camera indices are not provided."):
        g.get_outputs(rs, torch.zeros(2, 15))                      # R:lse_nerf/lse_field.py:295-296
    f.eval()
    tab, idx = f._eval_emb(4, "cpu")
    assert tab is None and idx is None                             # eval_mode "zero" (R:lse_nerf/lse_embeddings.py:51-55)
    est = la.LSEOccGridEstimator([-1, -1, -1, 1, 1, 1], 8, 2)
    est.eval()
    with pytest.raises(RuntimeError):
        est.update_every_n_steps(0, lambda x: x)


def test_outer_boundary_with_the_references_keyword_set():
    """The nerfstudio-facing surface with the reference's own keywords (no nerfstudio needed): the model accepts a config object
    that is NOT lsenerf_amd's dataclass -- what ``ns-train`` hands over -- and converts it field by field
    (R:lse_nerf/lsenerf.py:47-99); ``get_training_callbacks(training_callback_attributes)`` returns TrainingCallback objects bound
    to BEFORE_TRAIN_ITERATION (NGPModel); ``get_metrics_dict`` (R:lse_nerf/lsenerf.py:378-388) and the IS_EVAL parameter group
    (:246-247) exist."""
    import types
    import lsenerf_amd as la
    from lsenerf_amd import model as M, ns_plugin
    # every keyword of the reference's LSENeRFModelConfig with the reference's defaults + the InstantNGP fields it inherits
    ref_kw = dict(evs_loss_weight=1.0, emb_norm_weight=1e-2, event_loss_type="log_loss", use_mapping=False, mapping_method="mlp",
                  evs_mapping_method="None", ev_one_dim="learned", rgb_loss_type="linspace", use_mapper_loss=False,
                  mapper_loss_weight=0.25, scaler_weight=1.0, map_mode="ev_rgb", eval_num_rays_per_chunk=3512,
                  grid_resolution=128, grid_levels=4, max_res=2048, log2_hashmap_size=19, alpha_thre=0.01, cone_angle=0.004,
                  render_step_size=None, near_plane=0.05, far_plane=1e3, background_color="random", disable_scene_contraction=False)
    assert la.LSENeRFModelConfig(**ref_kw) == la.LSENeRFModelConfig()          # same names, same defaults
    foreign_embed = types.SimpleNamespace(embedding_type="evs_emb", metadata="dummy", emb_dim=32, eval_mode="mean", extra=1)
    foreign = types.SimpleNamespace(_target=object, collider_params=None, loss_coefficients={}, embed_config=foreign_embed,
                                    **{**ref_kw, "evs_mapping_method": None, "evs_loss_weight": 0.5, "grid_levels": 2,
                                       "grid_resolution": 16})
    cfg = ns_plugin.convert_model_config(foreign)
    assert isinstance(cfg, la.LSENeRFModelConfig) and cfg.evs_loss_weight == 0.5 and cfg.grid_levels == 2
    assert isinstance(cfg.embed_config, la.LSEEmbeddingConfig) and cfg.embed_config.embedding_type == "evs_emb" \
        and cfg.embed_config.eval_mode == "mean"
    m = la.LSENeRFModel(foreign, la.SceneBox(aabb=torch.tensor([[-1.0, -1, -1], [1, 1, 1]])), num_train_data=12)
    assert isinstance(m.config, la.LSENeRFModelConfig) and m.occupancy_grid.binaries.shape == (2, 16, 16, 16)
    assert type(m.field.embedding_appearance).__name__ == "EvsFrameEmbedding"
    # training callbacks: nerfstudio's signature and object shape
    (cb,) = m.get_training_callbacks(training_callback_attributes=object())
    assert cb.update_every_num_iters == 1 and len(cb.where_to_run) == 1 and "BEFORE_TRAIN_ITERATION" in str(cb.where_to_run[0]).upper()
    assert cb.func == m.update_occupancy_grid and hasattr(cb, "run_callback_at_location")
    (cb0,) = m.get_training_callbacks()
    assert type(cb0) is type(cb)
    # metrics
    out = {"rgb": torch.full((5, 3), 0.5), "num_samples_per_ray": torch.tensor([3, 0, 2, 1, 4])}
    md = m.get_metrics_dict(out, {"image": torch.full((5, 3), 0.6)})
    assert abs(float(md["psnr"]) - 20.0) < 1e-4 and int(md["num_samples_per_batch"]) == 10
    md2 = m.get_metrics_dict({"col_out": out, "prev_out": None, "next_out": None}, {"col_batch": {"image": torch.full((5, 3), 0.6)}})
    assert set(md2) == {"col"} and abs(float(md2["col"]["psnr"]) - 20.0) < 1e-4
    assert m.get_metrics_dict({"col_out": None, "prev_out": out, "next_out": out}, {}) == {}
    # evaluation run: only the appearance embedding is optimised
    n_all = len(m.get_param_groups()["fields"])
    M.gbconfig.IS_EVAL = True
    try:
        ev = m.get_param_groups()["fields"]
        assert n_all > 1 and len(ev) == 1 and ev[0] is m.field.embedding_appearance.embedding.weight
    finally:
        M.gbconfig.IS_EVAL = False
    # the plugin entry point still fails with the explanatory message where nerfstudio is absent
    with pytest.raises(ModuleNotFoundError, match="nerfstudio==0.3.2"):
        ns_plugin.build_method_specification()


def test_bench_request_floor_counts_the_lines_the_oracle_indexing_touches():
    """bench.py's second roofline of the hash backward (roofline.atomic) rests on hash_bwd_request_floor: distinct 64-byte lines
    of the gradient table per 64-sample window.  Checked here against a brute-force count over the oracle's corner indices."""
    import numpy as np
    import torch
    import bench
    from lsenerf_amd import ops
    from oracle import hashgrid as oh
    g = torch.Generator().manual_seed(5)
    n = 64 * 7 + 13                                              # a ragged tail: the last window is padded with its last sample
    t = torch.linspace(0.0, 1.0, n)[:, None]
    x01 = (0.3 + 0.4 * t * torch.tensor([[0.6, -0.3, 0.74]]) + 0.2 * torch.rand(1, 3, generator=g)).clamp(0.0, 1.0)   # a straight ray
    meta, ometa = ops.make_grid_meta(), oh.tcnn_grid_meta()
    want = 0
    for l in range(ometa.n_levels):
        assert ometa.offsets[l] % 8 == 0                         # levels start on 64-byte lines
        lines = (oh.tcnn_corner_indices(x01, ometa, l).numpy() - ometa.offsets[l]) >> 3
        for s in range(0, n, 64):
            want += np.unique(lines[s:s + 64]).size
    assert bench.hash_bwd_request_floor(x01, meta) == want


def test_unknown_map_mode_falls_through_like_the_reference():
    """The reference's DEFAULT map_mode is the spelling "ev_rgb" (R:lse_nerf/lsenerf.py:80), which names none of the three routes:
    its get_outputs falls through every branch (:335-363) -- no mapper runs, no "ev_out" is produced -- so with use_mapping an
    rgb-only batch works and an event batch fails in get_loss_dict with KeyError('ev_out') (:432-433).  Same here, never a
    KeyError('ev_rgb') from the routing plan, and such a configuration never takes the fused epilogue."""
    import lsenerf_amd as la
    m = la.LSENeRFModel(la.LSENeRFModelConfig(use_mapping=True, mapping_method="identity", num_levels=4, hidden_dim=32,
                                              hidden_dim_color=32, grid_levels=1, grid_resolution=8), torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), 4).train()
    assert m.config.map_mode == "ev_rgb"
    plan = m._plan()
    assert plan["mode"] is None and plan["unrouted"] and plan["ev_mapper"] is None and plan["ev_key"] == "ev_out"
    assert m._epilogue_desc() is None
    raw = {"rgb": torch.rand(6, 3), "accumulation": torch.rand(6, 1), "depth": torch.rand(6, 1)}
    routed = m.route_outputs(dict(raw), None, ev_out=True)
    assert "ev_out" not in routed and torch.equal(routed["rgb"], raw["rgb"].clamp_min(1e-5))
    loss = m.get_loss_dict({"col_out": routed, "prev_out": None, "next_out": None},
                           {"col_batch": {"image": torch.rand(6, 3)}, "evs_batch": None})
    assert set(loss) == {"rgb_loss"}
    with pytest.raises(KeyError, match="ev_out"):
        m.get_loss_dict({"col_out": None, "prev_out": routed, "next_out": routed},
                        {"col_batch": None, "evs_batch": {"image": torch.rand(6, 1)}})
    # a route that exists is unaffected
    m2 = la.LSENeRFModel(la.LSENeRFModelConfig(use_mapping=True, mapping_method="identity", map_mode="evs_rgb", num_levels=4, hidden_dim=32,
                                               hidden_dim_color=32, grid_levels=1, grid_resolution=8), torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), 4)
    assert m2._plan()["mode"] == "evs_rgb" and not m2._plan()["unrouted"]


def test_counter_summary_is_reported_only_for_the_sources_it_was_measured_on(tmp_path, monkeypatch):
    """profiles/pmc_traffic.json carries per-group digests of the kernel sources (lsenerf_amd/provenance.py); bench.py nulls the
    counter fields and says `traffic_stale` when the tree has moved on."""
    from lsenerf_amd import provenance as pv
    d = pv.source_digests()
    assert set(d) == {"hash", "mlp"} and all(len(v) == 16 for v in d.values())
    assert pv.counters_current({"source_digests": dict(d)}, "hash") and pv.counters_current({"source_digests": dict(d)}, "mlp")
    assert not pv.counters_current({}, "hash") and not pv.counters_current({"source_digests": {"hash": "0" * 16}}, "hash")
    # a one-byte change of a kernel source changes its group's digest and only that one
    src = tmp_path / "csrc"
    inc = tmp_path / "include"
    src.mkdir(); inc.mkdir()
    for name in {n for g in pv.GROUPS.values() for n in g}:
        with open(pv._path(name), "rb") as f:
            (inc if name == "lse_hip.h" else src).joinpath(name).write_bytes(f.read())
    monkeypatch.setattr(pv, "_CSRC", str(src)); monkeypatch.setattr(pv, "_INCLUDE", str(inc))
    assert pv.source_digests() == d
    with open(src / "mlp_x6.h", "ab") as f:
        f.write(b"\n")
    d2 = pv.source_digests()
    assert d2["hash"] == d["hash"] and d2["mlp"] != d["mlp"]
    assert pv.counters_current({"source_digests": dict(d)}, "hash") and not pv.counters_current({"source_digests": dict(d)}, "mlp")


def test_loading_a_grid_state_invalidates_what_was_derived_from_the_old_grid():
    from lsenerf_amd.grid_estimator import LSEOccGridEstimator
    a, b = LSEOccGridEstimator([-1.0, -1, -1, 1, 1, 1], 8, 2), LSEOccGridEstimator([-1.0, -1, -1, 1, 1, 1], 8, 2)
    a.occs.fill_(0.25); a.binaries.fill_(True)
    b._occ_mean_host = 0.0
    v = b.grid_version
    b.load_state_dict(a.state_dict())
    assert b.grid_version == v + 1 and b._occ_mean_host is None and float(b.occs.mean()) == 0.25
    v = b.grid_version
    b._occ_mean_host = 0.25
    b.mark_all_occupied(0.5)
    assert b.grid_version == v + 1 and b._occ_mean_host is None
    b.check_deferred_overflow()        # no count-free call was made: nothing to read, no device needed


def test_hash_bwd_replica_workspace_is_bounded_by_a_byte_budget():
    """16 replicas x the 4 coarsest levels is 16 MB for the default grid (levels of 16^3 .. 43^3 cells) -- and would be 268 MB for
    a grid whose coarse levels are already hashed (base resolution 64 at T = 2^19), where replicas buy nothing: the replicated
    levels stop at the first level that pushes one replica past 2 MB."""
    import ctypes
    from lsenerf_amd import _lib, ops
    lib = _lib.load()
    d = ops.make_grid_meta().desc()
    assert lib.lse_hash_bwd_workspace_bytes(ctypes.byref(d), None) == 16 * 2 * 125568 * 4
    big = ops.make_grid_meta(base_resolution=64, max_res=4096).desc()
    per_replica = [2 * 4 * int(big.offsets[l]) for l in range(1, 5)]
    assert per_replica[3] > (8 << 20)                                            # what the unbounded rule would replicate
    got = lib.lse_hash_bwd_workspace_bytes(ctypes.byref(big), None)
    keep = max([0] + [b for b in per_replica if b <= (2 << 20)])
    assert got == 16 * keep and got <= 16 * (2 << 20)


def test_raybundle_cat_of_row_blocks_of_one_buffer_is_that_buffer():
    """lsenerf_amd.graph keeps the three bundles of a captured step as row blocks of one static buffer per field, so that joining them
    (RayBundle.cat(alias_blocks=True), inside LSENeRFModel.train_step_bundles) costs no kernel and the ray gradients arrive in ONE leaf.
    The alias is the OPT-IN of callers that own the blocks: the plain join keeps torch.cat's semantics (a fresh tensor; in-place work
    on the joined bundle never reaches a data manager's source rays)."""
    from lsenerf_amd import RayBundle
    base_o = torch.randn(10, 3, requires_grad=True)
    base_d = torch.randn(10, 3)
    ids = torch.arange(10)
    parts = [RayBundle(origins=base_o[a:b], directions=base_d[a:b], metadata={"appearance_id": ids[a:b]}) for a, b in ((0, 4), (4, 7), (7, 10))]
    plain = RayBundle.cat(parts)
    assert plain.origins is not base_o and plain.origins.data_ptr() != base_o.data_ptr() and torch.equal(plain.origins, base_o)
    v0 = base_d._version
    plain.directions.mul_(2.0)                                 # (what a collider / normalisation may do to the joined bundle)
    assert base_d._version == v0 and not torch.equal(plain.directions, base_d)
    rb = RayBundle.cat(parts, alias_blocks=True)
    assert rb.origins is base_o and rb.directions is base_d and rb.metadata["appearance_id"] is ids
    (rb.origins * 2).sum().backward()
    assert torch.equal(base_o.grad, torch.full((10, 3), 2.0))
    # anything else is a real concatenation: a gap, another order, parts of different tensors, a partial cover
    for sel in (((0, 4), (5, 10)), ((4, 7), (0, 4), (7, 10)), ((0, 4), (4, 7))):
        p2 = [RayBundle(origins=base_o[a:b], directions=base_d[a:b]) for a, b in sel]
        r2 = RayBundle.cat(p2, alias_blocks=True)
        assert r2.origins is not base_o and torch.equal(r2.origins, torch.cat([base_o[a:b] for a, b in sel]))
    mixed = RayBundle.cat([RayBundle(origins=base_o[0:4], directions=base_d[0:4]), RayBundle(origins=torch.zeros(2, 3), directions=torch.zeros(2, 3))])
    assert mixed.origins.shape == (6, 3)


def test_hash_bwd_hooks_belong_to_their_table_and_die_with_their_owner():
    import gc
    from lsenerf_amd import ops
    t1, t2 = torch.zeros(64), torch.zeros(64)
    ops.set_hash_bwd_hook(t1, "before", lambda: None)
    assert ops.get_hash_bwd_hook(t1, "before") is not None and ops.get_hash_bwd_hook(t2, "before") is None
    assert ops.get_hash_bwd_hook(t1.view(8, 8), "before") is not None           # looked up by storage address (autograd hands out other objects)
    ops.set_hash_bwd_hook(t1, "before", None)
    assert not ops._HASH_BWD_HOOKS
    owner = t2.view(-1)                     # installed through an object that goes away while the storage lives on
    ops.set_hash_bwd_hook(owner, "split", (3, lambda: None))
    assert ops.get_hash_bwd_hook(t2, "split")[0] == 3
    del owner
    gc.collect()
    assert ops.get_hash_bwd_hook(t2, "split") is None and not ops._HASH_BWD_HOOKS


def test_event_thresholds_of_the_fused_enerf_loss():
    """evs_batch["e_thresh"] reaches lse_loss_epilogue_* as one float per event ray whatever form the data manager hands it in
    (R:lse_nerf/lse_dataset.py:63 a 1-element tensor per image, R:lse_nerf/lse_pixel_sampler.py:36-37 indexed per ray); a wrong
    length is an error, None stays None (= 1), and enerf_norm_loss configurations take the fused epilogue."""
    import lsenerf_amd as la
    from lsenerf_amd import _lib, ops
    like = torch.zeros(5, 1)
    assert ops._event_thresholds(None, 5, like) is None and ops._event_thresholds(0.2, 5, None) is None
    assert torch.equal(ops._event_thresholds(0.25, 5, like), torch.full((5,), 0.25))
    assert torch.equal(ops._event_thresholds(torch.tensor([0.5]), 5, like), torch.full((5,), 0.5))
    per_ray = torch.arange(5.0).reshape(5, 1) + 1
    out = ops._event_thresholds(per_ray, 5, like)
    assert out.shape == (5,) and out.is_contiguous() and torch.equal(out, per_ray.reshape(-1))
    with pytest.raises(ValueError, match="e_thresh"):
        ops._event_thresholds(torch.ones(4), 5, like)
    m = la.LSENeRFModel(la.LSENeRFModelConfig(use_mapping=True, mapping_method="identity", map_mode="co_map", evs_mapping_method="powpow",
                                              event_loss_type="enerf_norm_loss"), torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), 4)
    fields = m._epilogue_desc()[0]
    assert len(fields) == 7 and fields[6] == _lib.LSE_EVLOSS_ENERF_NORM
    assert m._e_thresh({"evs_batch": {"e_thresh": 0.2}}, fields) == 0.2
    with pytest.raises(KeyError):
        m._e_thresh({"evs_batch": {}}, fields)              # the reference's evs_batch["e_thresh"] (R:lse_nerf/lsenerf.py:416)
    log_fields = fields[:6] + (_lib.LSE_EVLOSS_LOG,)
    assert m._e_thresh({"evs_batch": {}}, log_fields) is None
