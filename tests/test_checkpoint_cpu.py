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
    assert set(sd) == {"state", "param_groups"} and len(sd["state"]) == 4
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
