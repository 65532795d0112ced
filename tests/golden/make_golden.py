#!/usr/bin/env python3
"""Generates the committed golden fixtures under tests/golden/ (seed 96 = the reference's script seed,
R:scripts/train_lse_data.sh:8).

PARITY UNPINNED: the reference holds no golden vectors for this path and its third-party dependencies
(nerfstudio/nerfacc/tinycudann) cannot be imported in the build container, so these fixtures are outputs of the
CPU oracle (oracle/), i.e. of the restated algorithms -- they freeze the oracle against drift and give the HIP path
a machine-independent target.  If a machine with the real stack becomes available, feed it the *inputs* stored
here and diff.

    python tests/golden/make_golden.py        # rewrites tests/golden/*.npz
Fixture inventory = SURVEY.md section 8c (i)-(vii), sized to stay small (big tables are re-generated from the
stored seed, and only outputs / checksums are stored).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import hashgrid as hg          # noqa: E402
from oracle import volrend as vr           # noqa: E402
from oracle import sampling as osamp       # noqa: E402
from oracle.field import FieldOracle, TcnnMLP, sh4_tcnn, sh4_nerfstudio   # noqa: E402
from oracle.model import ModelOracle       # noqa: E402

SEED = 96


def gen(seed=SEED):
    return torch.Generator().manual_seed(seed)


def bits_checksum(a: np.ndarray) -> np.int64:
    return np.int64(a.view(np.uint32).astype(np.uint64).sum() % (1 << 62))


def save(name, **arrs):
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **{k: np.asarray(v) for k, v in arrs.items()})


def hash_inputs(L, T, n, seed):
    g = gen(seed)
    meta = hg.tcnn_grid_meta(n_levels=L, log2_hashmap_size=T)
    table = (torch.rand(meta.n_params, generator=g) * 2 - 1) * 0.1
    x = torch.rand(n, 3, generator=g)
    w = torch.rand(n, L * 2, generator=g)
    return meta, table, x, w


def make_hash():
    for L, T, n in ((4, 10, 1024), (16, 12, 1024)):
        meta, table, x, w = hash_inputs(L, T, n, SEED + L)
        tc, xc = table.clone().requires_grad_(True), x.clone().requires_grad_(True)
        y = hg.hash_encode_tcnn(xc, tc, meta)
        (y * w).sum().backward()
        per_level_sum = np.array([float(tc.grad[2 * meta.offsets[l]:2 * meta.offsets[l + 1]].double().sum())
                                  for l in range(L)])
        save(f"hash_tcnn_L{L}_T{T}", L=L, T=T, n=n, seed=SEED + L, x=x.numpy(), y=y.detach().numpy(),
             dx=xc.grad.numpy(), dtable_head=tc.grad[:4096].numpy(), dtable_level_sums=per_level_sum,
             idx_level_last=hg.tcnn_corner_indices(x, meta, L - 1).numpy().astype(np.int64),
             offsets=np.array(meta.offsets), resolutions=np.array(meta.resolutions),
             scales=np.array(meta.scales, dtype=np.float32))
    # nerfstudio torch layout (CPU baseline field)
    g = gen(SEED)
    meta = hg.torch_grid_meta(4, 16, 128, 10, 2)
    table = hg.init_torch_table(meta, generator=g)
    x = torch.rand(512, 3, generator=g)
    save("hash_torch_L4_T10", x=x.numpy(), table=table.numpy(), y=hg.hash_encode_torch(x, table, meta).numpy(),
         scalings=meta.scalings.numpy())


BIG_HASH = ((4, 10), (16, 12), (16, 19))        # (levels, log2 table size): SURVEY 8c (i) at 4096 points; T = 19 is the production size
BIG_N, BIG_STRIDE = 4096, 8


def level_sums(t, meta, L, absolute=False):
    """float64 sum (of absolute values) of every level's slice of a flat table-shaped tensor."""
    f = (lambda v: v.abs()) if absolute else (lambda v: v)
    return np.array([float(f(t[2 * meta.offsets[l]:2 * meta.offsets[l + 1]].double()).sum()) for l in range(L)])


def make_hash_big():
    """4096-point fixtures of the tcnn layout, stored COMPACTLY: the table (48.8 MB at T = 19) and the upstream gradient ``w`` are
    regenerated from the stored seed by ``hash_inputs`` (torch's CPU generator is platform independent); of the outputs every 8th
    row of ``y`` is stored in full, plus per-level float64 sums / absolute sums of ALL rows and of the whole table gradient, ``dx`` in
    full, and the corner indices of the first hashed and the last level (the bit-exact integer side)."""
    for L, T in BIG_HASH:
        meta, table, x, w = hash_inputs(L, T, BIG_N, SEED + 100 * T + L)
        tc, xc = table.clone().requires_grad_(True), x.clone().requires_grad_(True)
        y = hg.hash_encode_tcnn(xc, tc, meta)
        (y * w).sum().backward()
        yd = y.detach().double().reshape(BIG_N, L, 2)
        hashed = [l for l in range(L) if not meta.is_dense(l)]
        save(f"hash_tcnn_L{L}_T{T}_n{BIG_N}", L=L, T=T, n=BIG_N, seed=SEED + 100 * T + L, stride=BIG_STRIDE, x=x.numpy(),
             y_rows=y.detach()[::BIG_STRIDE].numpy(), y_level_sums=yd.sum(dim=(0, 2)).numpy(), y_level_abs_sums=yd.abs().sum(dim=(0, 2)).numpy(),
             dx=xc.grad.numpy(), dtable_level_sums=level_sums(tc.grad, meta, L), dtable_level_abs_sums=level_sums(tc.grad, meta, L, True),
             dtable_head=tc.grad[:4096].numpy(), table_checksum=bits_checksum(table.numpy()), w_checksum=bits_checksum(w.numpy()),
             first_hashed_level=(hashed[0] if hashed else -1),
             idx_first_hashed=(hg.tcnn_corner_indices(x, meta, hashed[0]).numpy().astype(np.int64) if hashed else np.zeros((0, 8), np.int64)),
             idx_level_last=hg.tcnn_corner_indices(x, meta, L - 1).numpy().astype(np.int64),
             offsets=np.array(meta.offsets), resolutions=np.array(meta.resolutions), scales=np.array(meta.scales, dtype=np.float32))
    # nerfstudio torch layout at the reference's size (R:lse_nerf/lse_field.py:43-65: L = 16, T = 2^19, 16 -> 2048): the 67 MB table is
    # regenerated from the seed (init_torch_table), the 4096 points are stored
    g = gen(SEED + 1619)
    meta = hg.torch_grid_meta(16, 16, 2048, 19, 2)
    table = hg.init_torch_table(meta, generator=g)
    x = torch.rand(BIG_N, 3, generator=g)
    y = hg.hash_encode_torch(x, table, meta).double().reshape(BIG_N, 16, 2)
    save(f"hash_torch_L16_T19_n{BIG_N}", seed=SEED + 1619, n=BIG_N, stride=BIG_STRIDE, x=x.numpy(), table_checksum=bits_checksum(table.numpy()),
         y_rows=y.float().reshape(BIG_N, 32)[::BIG_STRIDE].numpy(), y_level_sums=y.sum(dim=(0, 2)).numpy(),
         y_level_abs_sums=y.abs().sum(dim=(0, 2)).numpy(), scalings=meta.scalings.numpy())


def make_mlp():
    for name, (i, layers, w, o, act) in {"base": (32, 2, 64, 16, None), "head": (63, 3, 64, 3, "Sigmoid")}.items():
        g = gen(SEED + len(name))
        m = TcnnMLP(i, layers, w, o, act)
        params = m.init_params(g)
        x = torch.randn(256, i, generator=g)
        wgt = torch.randn(256, o, generator=g)
        pc, xc = params.clone().requires_grad_(True), x.clone().requires_grad_(True)
        out = m.forward(xc, pc)
        (out * wgt).sum().backward()
        save(f"mlp_{name}", params=params.numpy(), x=x.numpy(), w=wgt.numpy(), out=out.detach().numpy(),
             dparams=pc.grad.numpy(), dx=xc.grad.numpy())


def make_sh():
    g = gen(SEED)
    d = torch.randn(1024, 3, generator=g)
    d = d / d.norm(dim=-1, keepdim=True)
    save("sh4", d=d.numpy(), tcnn=sh4_tcnn((d + 1) / 2).numpy(), nerfstudio=sh4_nerfstudio((d + 1) / 2).numpy())


def traverse_inputs(levels, res, seed):
    g = gen(seed)
    R = 256
    o = torch.randn(R, 3, generator=g)
    o = 1.5 * o / o.norm(dim=-1, keepdim=True)
    d = (torch.rand(R, 3, generator=g) - 0.5) - o
    d = d / d.norm(dim=-1, keepdim=True)
    b = torch.rand((levels, res, res, res), generator=g) < 0.3
    aabbs = torch.stack([osamp.enlarge_aabb(torch.tensor([-1.0, -1, -1, 1, 1, 1]), 2 ** i) for i in range(levels)])
    near = 0.05 + torch.rand(R, generator=g) * 0.0034641
    far = torch.full((R,), 1e3)
    return o, d, b, aabbs, near, far


def make_traverse():
    for levels, res, cone in ((1, 32, 0.0), (4, 32, 0.004), (4, 128, 0.004), (1, 128, 0.0)):
        seed = SEED + levels * 1000 + res
        o, d, b, aabbs, near, far = traverse_inputs(levels, res, seed)
        ri, ts, te, packed = osamp.traverse_grids(o, d, b, aabbs, near, far, 0.0034641, cone)
        n64 = int(packed[:64, 1].sum())
        save(f"traverse_l{levels}_r{res}_c{int(cone * 1000)}", levels=levels, res=res, cone=cone, seed=seed,
             step=np.float32(0.0034641), rays_o=o.numpy(), rays_d=d.numpy(), near=near.numpy(),
             counts=packed[:, 1].numpy().astype(np.int32), n_total=ri.numel(),
             ts_first64rays=ts[:n64].numpy(), te_first64rays=te[:n64].numpy(),
             ts_checksum=bits_checksum(ts.numpy()), te_checksum=bits_checksum(te.numpy()))


def make_volrend():
    g = gen(SEED)
    lengths = [0, 1, 63, 64, 65, 1024, 0, 7]
    cnt = torch.tensor(lengths)
    R = len(lengths)
    packed = torch.stack([torch.cumsum(cnt, 0) - cnt, cnt], -1)
    N = int(cnt.sum())
    ri = torch.repeat_interleave(torch.arange(R), cnt)
    ts = torch.rand(N, generator=g) * 2
    te = ts + 0.0034641
    sig = torch.rand(N, generator=g) * 30 * (torch.rand(N, generator=g) < 0.3)
    rgb = torch.rand(N, 3, generator=g)
    a, bb, c = torch.randn(R, 3, generator=g), torch.randn(R, generator=g), torch.randn(R, generator=g)
    sc, cc = sig.clone().requires_grad_(True), rgb.clone().requires_grad_(True)
    w = vr.render_weight_from_density(ts, te, sc, packed)[0]
    out_rgb = vr.accumulate_along_rays(w, cc, ri, R)
    acc = vr.accumulate_along_rays(w, None, ri, R)[:, 0]
    dep = vr.accumulate_along_rays(w, ((ts + te) / 2)[:, None], ri, R)[:, 0]
    ((out_rgb * a).sum() + (acc * bb).sum() + (dep * c).sum()).backward()
    vis = vr.render_visibility_from_density(ts, te, sig, packed, 1e-4, 0.05)
    save("volrend_ragged", lengths=np.array(lengths), ts=ts.numpy(), te=te.numpy(), sigma=sig.numpy(), rgb=rgb.numpy(),
         g_rgb=a.numpy(), g_acc=bb.numpy(), g_dep=c.numpy(), weights=w.detach().numpy(), out_rgb=out_rgb.detach().numpy(),
         acc=acc.detach().numpy(), depth_num=dep.detach().numpy(), d_sigma=sc.grad.numpy(), d_rgb=cc.grad.numpy(),
         visibility=vis.numpy())


def make_grid_update():
    og = osamp.OccGridOracle(torch.tensor([-1.0, -1, -1, 1, 1, 1]), 16, 2)
    fn = lambda x: torch.exp(-4 * (x ** 2).sum(-1, keepdim=True)) * 0.05   # noqa: E731
    og.update(0, fn, gen=gen(SEED))
    save("occ_update_step0", occs=og.occs.numpy(), binaries=og.binaries.numpy())


def make_config1():
    """BASELINE config 1: 32x32 RGB-only synthetic scene, L=4 hash grid / 2x32 MLP, torch-native field on CPU."""
    f = FieldOracle("torch", num_levels=4, hidden_dim=32, hidden_dim_color=32, log2_hashmap_size=12, max_res=128,
                    num_embeddings=1, contraction=False, seed=SEED)
    m = ModelOracle(f, grid_resolution=32, grid_levels=1, alpha_thre=0.0, cone_angle=0.0)
    m.grid.binaries[:] = True
    m.training = False
    H = W = 32
    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    dirs = torch.stack([(xs - W / 2) / W, -(ys - H / 2) / H, -torch.ones_like(xs, dtype=torch.float32)], -1).reshape(-1, 3)
    dirs = dirs / dirs.norm(dim=-1, keepdim=True)
    o = torch.tensor([0.0, 0.0, 2.5]).expand(H * W, 3).contiguous()
    with torch.no_grad():
        out = m.exec_get_outputs(o, dirs.contiguous())
    save("config1_render", rays_o=o.numpy(), rays_d=dirs.numpy(), rgb=out["rgb"].numpy(), acc=out["accumulation"].numpy(),
         depth=out["depth"].numpy(), counts=out["num_samples_per_ray"].numpy().astype(np.int32))


def epilogue_mlp_inputs():
    """Inputs of the loss-epilogue fixture: co_map routing, "rgb_mlp" on the colour side, ThreeToOne + "mlp" on the event side
    (R:lse_nerf/lsenerf.py:350-363, R:lse_nerf/intensity_mappers.py:28-62).  The mappers' parameters are nn.Linear's default draw
    (uniform +-1/sqrt(fan_in)), NOT the identity fit of :8-26 -- that fit is host-side set-up, and an unfitted mapper exercises both
    sides of every ReLU."""
    g = gen(SEED + 7)
    n_col, n_ev = 301, 130
    rad = {k: torch.rand(n, 3, generator=g) * 1.2 - 0.02 for k, n in (("col", n_col), ("prev", n_ev), ("next", n_ev))}
    rad["prev"][:3] = 0.0                                      # below the 1e-5 clamp
    col_gt, evs_gt = torch.rand(n_col, 3, generator=g), (torch.rand(n_ev, 1, generator=g) - 0.5) * 0.4
    w31 = torch.tensor([[0.2, 0.5, 0.3]])

    def mlp(in_dim):
        dims = [in_dim, 16, 16, 16, in_dim]
        out = []
        for l in range(4):
            k = 1.0 / dims[l] ** 0.5
            out += [(torch.rand(dims[l + 1], dims[l], generator=g) * 2 - 1) * k, (torch.rand(dims[l + 1], generator=g) * 2 - 1) * k]
        return out
    return rad, col_gt, evs_gt, w31, mlp(3), mlp(1)


def make_epilogue():
    from oracle.losses import loss_dict, mlp_mapper, route_outputs
    rad, col_gt, evs_gt, w31, p_rgb, p_evs = epilogue_mlp_inputs()
    leaf = lambda t: t.clone().requires_grad_(True)
    rad_l = {k: leaf(v) for k, v in rad.items()}
    w31_l, prgb_l, pevs_l = leaf(w31), [leaf(p) for p in p_rgb], [leaf(p) for p in p_evs]
    kw = dict(training=True, use_mapping=True, map_mode="co_map", rgb_loss_type="linspace", rgb_mapper=mlp_mapper(prgb_l),
              evs_mapper=mlp_mapper(pevs_l), three_to_one_w=w31_l)
    routed = [route_outputs(rad_l[k], ev_out=(k != "col"), **kw) for k in ("col", "prev", "next")]
    ld = loss_dict(routed[0], routed[1], routed[2], col_gt, evs_gt, use_mapping=True, evs_loss_weight=1.0)
    (ld["rgb_loss"] * 1.3 + ld["event_loss"] * 0.6).backward()
    arrs = dict(col=rad["col"].numpy(), prev=rad["prev"].numpy(), next=rad["next"].numpy(), col_gt=col_gt.numpy(), evs_gt=evs_gt.numpy(),
                w31=w31.numpy(), loss_weights=np.float32([1.3, 0.6]), rgb_loss=np.float32(float(ld["rgb_loss"].detach())),
                event_loss=np.float32(float(ld["event_loss"].detach())), d_col=rad_l["col"].grad.numpy(), d_prev=rad_l["prev"].grad.numpy(),
                d_next=rad_l["next"].grad.numpy(), d_w31=w31_l.grad.numpy())
    for side, ps, pl in (("rgb", p_rgb, prgb_l), ("evs", p_evs, pevs_l)):
        for i, (p, l) in enumerate(zip(ps, pl)):
            arrs[f"{side}_mlp_p{i}"] = p.numpy()
            arrs[f"{side}_mlp_d{i}"] = l.grad.numpy()
    # the same inputs under enerf_norm_loss (R:lse_nerf/lsenerf.py:406-419) with a per-ray event threshold
    e_thresh = 0.15 + 0.1 * torch.rand(rad["prev"].shape[0], 1, generator=gen(SEED + 8))
    rad_e = {k: leaf(v) for k, v in rad.items()}
    w31_e, prgb_e, pevs_e = leaf(w31), [leaf(p) for p in p_rgb], [leaf(p) for p in p_evs]
    kw.update(rgb_mapper=mlp_mapper(prgb_e), evs_mapper=mlp_mapper(pevs_e), three_to_one_w=w31_e)
    routed = [route_outputs(rad_e[k], ev_out=(k != "col"), **kw) for k in ("col", "prev", "next")]
    le = loss_dict(routed[0], routed[1], routed[2], col_gt, evs_gt, use_mapping=True, evs_loss_weight=1.0, event_loss="enerf_norm_loss",
                   e_thresh=e_thresh)
    le["event_loss"].backward()
    arrs.update(e_thresh=e_thresh.numpy(), enerf_event_loss=np.float32(float(le["event_loss"].detach())),
                enerf_d_prev=rad_e["prev"].grad.numpy(), enerf_d_next=rad_e["next"].grad.numpy(), enerf_d_w31=w31_e.grad.numpy())
    for i, l in enumerate(pevs_e):
        arrs[f"enerf_evs_mlp_d{i}"] = l.grad.numpy()
    save("loss_epilogue_co_map_mlp", **arrs)


if __name__ == "__main__":
    torch.set_num_threads(4)
    if "--only-big-hash" in sys.argv:          # (round 5: the 4096-point fixtures were added without rewriting the older files)
        make_hash_big()
    elif "--only-epilogue" in sys.argv:        # (round 5, ABI 6: likewise)
        make_epilogue()
    else:
        make_hash(); make_hash_big(); make_mlp(); make_sh(); make_traverse(); make_volrend(); make_grid_update(); make_config1()
        make_epilogue()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f"{f:40s} {os.path.getsize(os.path.join(HERE, f)) / 1024:8.1f} KB")
