"""nerfstudio .ckpt import / export: key mapping, latest-step rule, round trip, reference-style files."""
import os

import pytest
import torch


def _model(levels=2, res=16):
    from lsenerf_amd import LSENeRFModel, LSENeRFModelConfig
    torch.manual_seed(3)
    cfg = LSENeRFModelConfig(grid_levels=levels, grid_resolution=res, num_levels=4, log2_hashmap_size=12)
    return LSENeRFModel(cfg, torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), num_train_data=5)


def test_key_mapping_both_ways():
    from lsenerf_amd import checkpoint as ck
    ref = {"_model.field.mlp_base_grid.tcnn_encoding.params": "field.mlp_base_grid.params",
           "_model.field.mlp_base_mlp.tcnn_encoding.params": "field.mlp_base_mlp.params",
           "_model.field.mlp_head.tcnn_encoding.params": "field.mlp_head.params",
           "_model.field.embedding_appearance.embedding.weight": "field.embedding_appearance.embedding.weight",
           "_model.occupancy_grid.occs": "occupancy_grid.occs",
           "module._model.occupancy_grid.binaries": "occupancy_grid.binaries",          # DDP prefix
           "_model.field.mlp_base_grid.hash_table": None,                                 # dead torch-layout table
           "datamanager.train_camera_optimizer.pose_adjustment": None}
    for k, v in ref.items():
        assert ck.reference_to_local_key(k) == v, k
    for k, v in ref.items():
        if v is not None and not k.startswith("module."):
            assert ck.local_to_reference_key(v) == k


def test_round_trip_and_latest_step(tmp_path):
    from lsenerf_amd import checkpoint as ck
    a, b = _model(), _model()
    with torch.no_grad():
        for p in a.parameters():
            p.add_(torch.randn_like(p) * 0.1)
        a.occupancy_grid.occs.uniform_(0, 1)
        a.occupancy_grid.binaries.copy_(a.occupancy_grid.occs.view_as(a.occupancy_grid.binaries) > 0.5)
    d = str(tmp_path / "nerfstudio_models")
    ck.save_nerfstudio_checkpoint(d, b, step=10)
    p = ck.save_nerfstudio_checkpoint(d, a, step=2000)
    assert os.path.basename(p) == "step-000002000.ckpt" and ck.latest_step(d) == 2000
    saved = torch.load(p, weights_only=True)
    assert set(saved) == {"step", "pipeline", "optimizers", "scalers"} and saved["step"] == 2000
    assert "_model.field.mlp_base_grid.tcnn_encoding.params" in saved["pipeline"]
    assert "_model.field.mlp_head.tcnn_encoding.params" in saved["pipeline"]
    info = ck.load_nerfstudio_checkpoint(d, b)                     # directory -> latest step
    assert info["step"] == 2000 and info["missing"] == [] and info["unexpected"] == []
    for (k, va), (_, vb) in zip(a.state_dict().items(), b.state_dict().items()):
        assert torch.equal(va, vb), k
    info = ck.load_nerfstudio_checkpoint(d, b, load_step=10)
    assert info["step"] == 10
    ck.save_nerfstudio_checkpoint(d, a, step=3000, keep_only_latest=True)
    assert sorted(os.listdir(d)) == ["step-000003000.ckpt"]
    with pytest.raises(FileNotFoundError):
        ck.load_nerfstudio_checkpoint(d, b, load_step=7)


def test_reference_style_file_with_dead_table_half_params_and_ddp_prefix(tmp_path):
    """What a checkpoint written by the reference looks like: DDP `module.` prefix, the dead 67 MB-style torch table, a
    half-precision tcnn build, data-manager and camera-optimiser entries."""
    from lsenerf_amd import checkpoint as ck
    m = _model()
    sd = m.state_dict()
    pipe = {"module." + ck.local_to_reference_key(k): v.clone() for k, v in sd.items()}
    gk = "module._model.field.mlp_base_grid.tcnn_encoding.params"
    pipe[gk] = torch.linspace(-1, 1, sd["field.mlp_base_grid.params"].numel()).half()
    pipe["module._model.field.mlp_base_grid.hash_table"] = torch.zeros(64, 2)
    pipe["module._model.camera_optimizer.pose_adjustment"] = torch.zeros(5, 6)
    pipe["module.datamanager.train_camera_optimizer.pose_adjustment"] = torch.zeros(5, 6)
    f = str(tmp_path / "step-000000123.ckpt")
    torch.save({"step": 123, "pipeline": pipe, "optimizers": {}, "scalers": {}}, f)
    info = ck.load_nerfstudio_checkpoint(f, m, drop_camera_optimizer=True)
    assert info["step"] == 123
    assert any("hash_table" in k for k in info["dropped"]) and any("datamanager" in k for k in info["dropped"])
    assert any("camera_optimizer" in k for k in info["dropped"]) and info["unexpected"] == []
    got = m.field.mlp_base_grid.params
    assert got.dtype == torch.float32 and torch.allclose(got, pipe[gk].float())
    # shape mismatches are loud
    pipe[gk] = torch.zeros(7)
    torch.save({"step": 1, "pipeline": pipe, "optimizers": {}, "scalers": {}}, f)
    with pytest.raises(ValueError, match="shape mismatch"):
        ck.load_nerfstudio_checkpoint(f, m)
    torch.save({"foo": 1}, f)
    with pytest.raises(ValueError, match="not a nerfstudio checkpoint"):
        ck.load_nerfstudio_checkpoint(f, m)


def test_flat_adam_state_survives_a_checkpoint_round_trip(tmp_path):
    """Training resume: FlatAdam's moments / step count are stored in torch.optim.Adam's layout under the reference's
    "fields" param-group name and come back through the weights_only loader (parity unpinned: the reference ships no
    checkpoint, the layout follows nerfstudio 0.3.2's Trainer.save_checkpoint)."""
    from lsenerf_amd.checkpoint import load_nerfstudio_checkpoint, save_nerfstudio_checkpoint
    from lsenerf_amd.optim import FlatAdam, FlatParams
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Linear(7, 2))
    flat = FlatParams(model.parameters())
    opt = FlatAdam(flat, lr=1e-2, lr_final=1e-4, max_steps=100)
    opt.exp_avg.copy_(torch.randn_like(opt.exp_avg))
    opt.exp_avg_sq.copy_(torch.rand_like(opt.exp_avg_sq))
    opt.step_count = 37
    sd = opt.state_dict()
    assert set(sd) == {"state", "param_groups", "lsenerf_amd_layout"} and len(sd["state"]) == 4
    assert sd["state"][0]["exp_avg"].shape == model[0].weight.shape
    ref = torch.optim.Adam(model.parameters(), lr=1e-2)
    ref.load_state_dict(sd)                                    # the layout is torch.optim.Adam's own
    save_nerfstudio_checkpoint(str(tmp_path), model, 37, optimizers={"fields": opt})
    with pytest.raises(TypeError):
        save_nerfstudio_checkpoint(str(tmp_path), model, 38, optimizers={"fields": object()})
    model2 = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Linear(7, 2))
    flat2 = FlatParams(model2.parameters())
    opt2 = FlatAdam(flat2, lr=1e-2, lr_final=1e-4, max_steps=100)
    res = load_nerfstudio_checkpoint(str(tmp_path), model2, optimizers={"fields": opt2})
    assert res["step"] == 37 and opt2.step_count == 37
    sd2 = opt2.state_dict()
    for i in sd["state"]:                                      # (the 64-float alignment pads between parameters carry no state)
        assert torch.equal(sd2["state"][i]["exp_avg"], sd["state"][i]["exp_avg"])
        assert torch.equal(sd2["state"][i]["exp_avg_sq"], sd["state"][i]["exp_avg_sq"])
    assert abs(opt2.current_lr() - opt.current_lr()) < 1e-12


def test_reference_written_optimizer_state_is_matched_by_name_not_by_position(tmp_path):
    """A "fields" optimizer state written by the REFERENCE indexes the reference's parameter list: embedding first, then the
    dead torch-layout hash table (never receives a gradient -> no state entry, but it consumes an index), zero-sized tcnn
    parameters of the SH encoding, the three tcnn parameter vectors, then the mapper modules (R:lse_nerf/lsenerf.py:231-249,
    module order of R:lse_nerf/lse_field.py:155-262).  The loader translates indices through the pipeline key names; moments
    land on the tensors of the same name, unknown non-empty parameters and shape mismatches raise (parity unpinned: no
    reference checkpoint exists, the layout follows torch.optim.Adam + nerfstudio 0.3.2)."""
    import lsenerf_amd as la
    from lsenerf_amd import checkpoint as ck
    from lsenerf_amd.optim import FlatAdam, FlatParams
    m = la.LSENeRFModel(la.LSENeRFModelConfig(grid_resolution=8, grid_levels=1, log2_hashmap_size=8, num_levels=4,
                                              use_mapping=True, mapping_method="powpow", map_mode="co_map",
                                              evs_mapping_method="powpow"),
                        torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), 6)
    sd = m.state_dict()
    f = lambda k: sd[k].clone()
    # the reference's registration order (buffers interleaved, Sequential alias at the end of the field)
    pipe = {}
    pipe["_model.field.aabb"] = f("field.aabb")
    pipe["_model.field.max_res"] = f("field.max_res")
    pipe["_model.field.num_levels"] = f("field.num_levels")
    pipe["_model.field.log2_hashmap_size"] = f("field.log2_hashmap_size")
    pipe["_model.field.embedding_appearance.embedding.weight"] = f("field.embedding_appearance.embedding.weight")   # index 0
    pipe["_model.field.direction_encoding.tcnn_encoding.params"] = torch.zeros(0)                                   # index 1
    pipe["_model.field.mlp_base_grid.hash_table"] = torch.zeros(4 * 256, 2)                                         # index 2 (dead)
    pipe["_model.field.mlp_base_grid.tcnn_encoding.params"] = f("field.mlp_base_grid.params")                       # index 3
    pipe["_model.field.mlp_base_mlp.tcnn_encoding.params"] = f("field.mlp_base_mlp.params")                         # index 4
    pipe["_model.field.mlp_base.0.hash_table"] = pipe["_model.field.mlp_base_grid.hash_table"]                      # aliases
    pipe["_model.field.mlp_base.0.tcnn_encoding.params"] = pipe["_model.field.mlp_base_grid.tcnn_encoding.params"]
    pipe["_model.field.mlp_base.1.tcnn_encoding.params"] = pipe["_model.field.mlp_base_mlp.tcnn_encoding.params"]
    pipe["_model.field.mlp_head.tcnn_encoding.params"] = f("field.mlp_head.params")                                 # index 5
    pipe["_model.occupancy_grid.occs"] = f("occupancy_grid.occs")
    pipe["_model.rgb_mapper.pow_coeff"] = f("rgb_mapper.pow_coeff")                                                 # index 6
    pipe["_model.evs_mapper.pow_coeff"] = f("evs_mapper.pow_coeff")                                                 # index 8 (after rgb_to_one)
    pipe["_model.rgb_to_one.weights"] = f("rgb_to_one.weights")                                                     # index 7
    order = ck.reference_param_order(pipe)
    assert order == ["field.embedding_appearance.embedding.weight", "field.direction_encoding.tcnn_encoding.params",
                     "field.mlp_base_grid.hash_table", "field.mlp_base_grid.tcnn_encoding.params",
                     "field.mlp_base_mlp.tcnn_encoding.params", "field.mlp_head.tcnn_encoding.params", "rgb_mapper.pow_coeff",
                     "rgb_to_one.weights", "evs_mapper.pow_coeff"]
    g = torch.Generator().manual_seed(2)
    state = {}
    for i, k in enumerate(order):
        if i in (1, 2):                   # no gradient ever -> torch.optim.Adam holds no state for them
            continue
        t = pipe["_model." + k]
        state[i] = {"step": torch.tensor(41.0), "exp_avg": torch.randn(t.shape, generator=g), "exp_avg_sq": torch.rand(t.shape, generator=g)}
    opt_sd = {"state": state, "param_groups": [{"lr": 1e-2, "betas": (0.9, 0.999), "eps": 1e-15, "params": list(range(len(order)))}]}
    path = str(tmp_path / "step-000000041.ckpt")
    torch.save({"step": 41, "pipeline": pipe, "optimizers": {"fields": opt_sd}, "scalers": {}}, path)

    flat = FlatParams(m.get_param_groups()["fields"])
    opt = FlatAdam(flat, lr=1e-2, eps=1e-15)
    ck.load_nerfstudio_checkpoint(path, m, optimizers={"fields": opt})
    assert opt.step_count == 41
    name_of = {id(p): n for n, p in m.named_parameters()}
    for p, o in zip(flat.params, flat.offsets):
        ref_key = ck.local_to_reference_key(name_of[id(p)])[len("_model."):]
        i = order.index(ref_key)
        assert torch.equal(opt.exp_avg[o:o + p.numel()].view(p.shape), state[i]["exp_avg"]), ref_key
        assert torch.equal(opt.exp_avg_sq[o:o + p.numel()].view(p.shape), state[i]["exp_avg_sq"]), ref_key
    # the positional reading of the same file is refused (9 reference entries vs 7 local parameters)
    with pytest.raises(ValueError, match="not a state dict of this parameter list"):
        FlatAdam(flat, lr=1e-2).load_state_dict(opt_sd)
    # a non-empty reference parameter without a local counterpart is named in the error
    pipe2 = dict(pipe)
    pipe2["_model.field.mlp_transient.tcnn_encoding.params"] = torch.zeros(5)
    with pytest.raises(ValueError, match="mlp_transient"):
        ck.reference_optimizer_index_map(pipe2, m, flat.params)
