import sys, torch
sys.path.insert(0, '.')
from tests.util import make_model_pair, random_rays, nmax_err
from lsenerf_amd import ops
from oracle import field as ofield
hip, orc = make_model_pair(grid_levels=4, grid_resolution=64, occupied_frac=0.25, param_scale=300.0, alpha_thre=0.0)
R = 96
o, d = random_rays(R, seed=21)
b = hip.occupancy_grid
near = torch.full((R,), 0.05).cuda(); far = torch.full((R,), 1e3).cuda()
ri, ts, te, packed = ops.traverse_grids(o.cuda(), d.cuda(), b._binaries_u8(), b.aabbs, near, far, hip.config.render_step_size, 0.004)
print("N", ri.shape)
fld = hip.field
x01, sel = ops.positions(o.cuda(), d.cuda(), ri, ts, te, packed, True, None)
li = ri.long().cpu()
pos = ofield.frustum_positions(o[li], d[li], ts.cpu()[:, None], te.cpu()[:, None])
p_ref, sel_ref = orc.field.normalize(pos)
print("x01", nmax_err(x01, p_ref), "sel eq", bool((sel.cpu().bool() == sel_ref).all()), float(sel_ref.float().mean()))
y = fld.mlp_base_grid.forward_levelmajor(x01)
y_ref = orc.field.encode(p_ref)
print("hash", nmax_err(y.permute(1, 0, 2).reshape(y.shape[1], -1), y_ref), float(y_ref.abs().max()))
h = ops.fused_mlp(fld.mlp_base_mlp.params, y, fld.mlp_base_mlp.meta(), y.shape[1])
h_ref = orc.field.mlp_base(y_ref)
print("base mlp", nmax_err(h, h_ref), float(h_ref.abs().max()))
sigma = ops.density_from_mlp_out(h, sel, 1.0)
dens_ref, geo_ref = orc.field.get_density(pos)
print("sigma", nmax_err(sigma, dens_ref[:, 0]))
rgb16 = fld.rgb_packed(h, d.cuda(), torch.zeros(R, dtype=torch.int32).cuda(), ri, packed, fld._train_emb_table())
rgb_ref = orc.field.get_outputs(d[li], geo_ref, torch.zeros(li.shape[0], dtype=torch.long))
print("rgb", nmax_err(rgb16[:, :3], rgb_ref))
# pieces of the head
feat = ops.ray_features(d.cuda(), fld._train_emb_table(), torch.zeros(R, dtype=torch.int32).cuda())
sh = ofield.sh4_tcnn((d + 1) / 2)
print("sh", nmax_err(feat[:, :16], sh), "emb", nmax_err(feat[:, 31:63], orc.field.params["embedding"][0].expand(R, 32)))
