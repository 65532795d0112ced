"""GPU parity tests proper: every HIP kernel, called through the C-ABI, against the CPU oracle on the same seeded
inputs (sizes the oracle finishes in seconds).  Bit-exact for integer/index outputs and sampler edges; stated fp32
tolerances (tests/util.py) elsewhere.  The oracle itself is "parity unpinned" (oracle/__init__.py)."""
import numpy as np
import pytest
import torch

from tests.util import TOL_GRAD_BLOCK, blockwise_nmax_err, hash_level_bounds, mlp_row_bounds, rel_l2  # noqa: E402
from tests.util import TOL_FWD, TOL_GRAD, nmax_err, random_binaries, random_rays

pytestmark = pytest.mark.gpu


def _ops():
    from lsenerf_amd import ops
    return ops


# ------------------------------------------------------------------------------------------------ sampler
@pytest.mark.parametrize("levels,res,cone,step", [(1, 32, 0.0, 0.01), (4, 32, 0.004, 0.005), (4, 128, 0.004, 0.0034641),
                                                  (2, 16, 0.0, 0.02), (1, 128, 0.0, 0.0)])
def test_traverse_bit_exact(levels, res, cone, step):
    from oracle import sampling as osamp
    ops = _ops()
    R = 300 if step > 0 else 64
    o, d = random_rays(R, seed=levels * 10 + res)
    if levels == 2:   # some rays starting inside the scene
        o2, d2 = random_rays(R, seed=5, inside=True)
        o[::2], d[::2] = o2[::2], d2[::2]
    b = random_binaries(levels, res, 0.3, seed=res)
    aabbs = torch.stack([osamp.enlarge_aabb(torch.tensor([-1.0, -1, -1, 1, 1, 1]), 2 ** i) for i in range(levels)])
    near = torch.full((R,), 0.05) + torch.rand(R, generator=torch.Generator().manual_seed(1)) * max(step, 1e-3)
    far = torch.full((R,), 1e3)
    ri, ts, te, packed = osamp.traverse_grids(o, d, b, aabbs, near, far, step, cone)
    hri, hts, hte, hpacked = ops.traverse_grids(o.cuda(), d.cuda(), b.cuda().view(torch.uint8), aabbs.cuda(),
                                                near.cuda(), far.cuda(), step, cone)
    assert ri.numel() > 100
    assert torch.equal(hpacked.cpu(), packed)
    assert torch.equal(hri.cpu().long(), ri)
    assert torch.equal(hts.cpu(), ts), "t_starts differ bitwise"
    assert torch.equal(hte.cpu(), te), "t_ends differ bitwise"


def test_traverse_fma_setup_is_bit_exact_against_the_fma_oracle():
    """Call flag LSE_TRAVERSE_FMA_SETUP (``fma_setup=True``): the a*b+c sites of the traversal set-up as fused multiply-adds, i.e. what nvcc's default
    contraction makes of nerfacc's grid.cu.  1024 sphere rays through a carved 4-level 128^3 grid with cone-angle steps (the
    workload on which the two arithmetic conventions differ in about one interval per million,
    tests/test_oracle_cpu.py::test_fma_contraction_changes_few_sample_intervals): the kernel with the option on equals the FMA
    build of the C oracle bit for bit, with the option off the strict build -- each setting is exact against its own oracle."""
    from oracle import sampling as osamp
    from lsenerf_amd import _lib
    ops = _ops()
    g = torch.Generator().manual_seed(96)
    R = 1024
    o = torch.randn(R, 3, generator=g)
    o = 1.5 * o / o.norm(dim=-1, keepdim=True)
    d = (torch.rand(R, 3, generator=g) - 0.5) - o
    d = d / d.norm(dim=-1, keepdim=True)
    aabbs = torch.stack([osamp.enlarge_aabb(torch.tensor([-1.0, -1, -1, 1, 1, 1]), 2 ** i) for i in range(4)])
    step = float(np.float32(2 * np.sqrt(3.0) / 1000))
    near, far = torch.full((R,), 0.05), torch.full((R,), 1e3)
    b = torch.rand(4, 128, 128, 128, generator=g) < 0.43
    for fma in (1, 0):        # a call argument since ABI 5 (LSE_TRAVERSE_FMA_SETUP): nothing to set, nothing to restore
        ri, ts, te, packed = osamp.traverse_grids(o, d, b, aabbs, near, far, step, 0.004, fma=bool(fma))
        hri, hts, hte, hpacked = ops.traverse_grids(o.cuda(), d.cuda(), b.cuda().view(torch.uint8), aabbs.cuda(), near.cuda(),
                                                    far.cuda(), step, 0.004, fma_setup=bool(fma))
        assert ri.numel() > 300_000
        assert torch.equal(hpacked.cpu(), packed), fma
        assert torch.equal(hri.cpu().long(), ri) and torch.equal(hts.cpu(), ts) and torch.equal(hte.cpu(), te), fma


def test_estimator_carries_the_traversal_convention_to_both_sampler_paths():
    """``LSEOccGridEstimator.traverse_fma`` (the attribute that replaced the process-wide option of ABI 4) reaches the synchronising
    marcher and the count-free (deferred) one: with it, both equal the FMA build of the C oracle bit for bit; without it, the
    strict build.  A carved 2-level 32^3 grid with cone-angle steps, where the two conventions differ in a handful of intervals."""
    from oracle import sampling as osamp
    from lsenerf_amd.grid_estimator import LSEOccGridEstimator
    g = torch.Generator().manual_seed(11)
    R = 2048
    o = torch.randn(R, 3, generator=g)
    o = 1.5 * o / o.norm(dim=-1, keepdim=True)
    d = (torch.rand(R, 3, generator=g) - 0.5) - o
    d = d / d.norm(dim=-1, keepdim=True)
    b = torch.rand(2, 32, 32, 32, generator=g) < 0.4
    est = LSEOccGridEstimator([-1.0, -1, -1, 1, 1, 1], resolution=32, levels=2).cuda()
    est.binaries.copy_(b.cuda())
    step = 0.005
    near, far = torch.full((R,), 0.05), torch.full((R,), 1e3)
    for fma in (False, True):
        ri, ts, te, packed = osamp.traverse_grids(o, d, b, est.aabbs.cpu(), near, far, step, 0.004, fma=fma)
        est.traverse_fma = fma
        h_ri, h_ts, h_te = est.sampling(o.cuda(), d.cuda(), near_plane=0.05, far_plane=1e3, render_step_size=step, cone_angle=0.004,
                                        stratified=False)
        assert torch.equal(h_ri.cpu().long(), ri) and torch.equal(h_ts.cpu(), ts) and torch.equal(h_te.cpu(), te), fma
        res = est.sampling(o.cuda(), d.cuda(), near_plane=0.05, far_plane=1e3, render_step_size=step, cone_angle=0.004,
                           stratified=False, deferred=True)
        n = int(res[4])
        assert n == ri.numel() and torch.equal(res[1][:n].cpu(), ts) and torch.equal(res[2][:n].cpu(), te), fma
        est.check_deferred_overflow()


def test_c_abi_calls_from_concurrent_host_threads_on_their_own_streams():
    """ABI 5: "no global or thread-local state ... any number of host threads may call concurrently on their own streams".  Four
    Python threads (ctypes releases the GIL inside every call), each with its own stream, tensors, sizes and device-side count,
    run hash forward / backward and both fused MLPs for a while; every thread's results equal what the same calls give alone
    (forward and d(x) bit for bit -- no atomics there --, table gradients to float-atomic noise).  One of the threads also provokes
    argument errors throughout: the message it reads back is its own (lse_last_error is per thread) and nobody else's calls fail."""
    import threading
    from lsenerf_amd import _lib
    ops = _ops()
    meta = ops.make_grid_meta(n_levels=8, log2_hashmap_size=15)
    g = torch.Generator().manual_seed(3)
    table = ((torch.rand(meta.n_params, generator=g) * 2 - 1) * 0.1).cuda()
    mm = ops.MlpMeta(32, 64, 1, _lib.LSE_ACT_NONE, _lib.LSE_IN_LEVELMAJOR)
    params = (torch.randn(mm.n_params, generator=g) * 0.2).cuda()
    sizes = (7000, 12345, 20011, 4099)
    xs = [torch.rand(n, 3, generator=g).cuda() for n in sizes]
    ws = [torch.randn(meta.n_levels, n, 2, generator=g).cuda() for n in sizes]
    ins = [torch.randn(16, n, 2, generator=g).cuda() for n in sizes]
    counts = [torch.tensor([n - 17 * (i + 1)], dtype=torch.int64, device="cuda") for i, n in enumerate(sizes)]

    def work(i, n_dev):
        x = xs[i].clone().requires_grad_(True)
        t = table.clone().requires_grad_(True)
        y = ops.hash_encode(x, t, meta, n_dev=n_dev)
        (y * ws[i]).sum().backward()
        out = ops.fused_mlp(params, ins[i], mm, sizes[i])
        return y.detach(), x.grad, t.grad, out.detach()

    alone = [work(i, counts[i]) for i in range(4)]
    torch.cuda.synchronize()
    results, errors = [None] * 4, []

    def run(i):
        try:
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                for it in range(12):
                    r = work(i, counts[i])
                    if i == 3:      # this thread's own mistakes must stay its own
                        d = _lib.GridDesc()
                        d.n_levels, d.n_features = 4, 3
                        rc = _lib.load().lse_hash_fwd(ctypes.byref(d), None, None, None, 8, None, None)
                        assert rc == -1 and b"n_features" in _lib.load().lse_last_error()
                s.synchronize()
            results[i] = r
        except Exception as e:      # noqa: BLE001
            errors.append((i, repr(e)))
    import ctypes
    threads = [threading.Thread(target=run, args=(i,)) for i in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for i in range(4):
        m = int(counts[i])
        y, dx, dt, out = results[i]
        y0, dx0, dt0, out0 = alone[i]
        assert torch.equal(y[:, :m], y0[:, :m]) and torch.equal(dx[:m], dx0[:m]) and torch.equal(out, out0), i
        assert nmax_err(dt, dt0, 1e-20) < 1e-5, i


@pytest.mark.parametrize("levels,res,cone,step", [(1, 32, 0.0, 0.01), (4, 32, 0.004, 0.005), (4, 128, 0.0, 0.0034641)])
def test_traverse_single_pass_equals_two_pass(levels, res, cone, step):
    """The single-pass marcher (fixed-capacity ray slots + compaction) returns the two-pass result bit for bit; a violated
    capacity bound is detected and falls back."""
    from oracle import sampling as osamp
    ops = _ops()
    R = 257
    o, d = random_rays(R, seed=levels + res)
    b = random_binaries(levels, res, 0.4, seed=res + 1)
    aabbs = torch.stack([osamp.enlarge_aabb(torch.tensor([-1.0, -1, -1, 1, 1, 1]), 2 ** i) for i in range(levels)]).cuda()
    near = (torch.full((R,), 0.05) + torch.rand(R, generator=torch.Generator().manual_seed(2)) * step).cuda()
    far = torch.full((R,), 1e3).cuda()
    args = (o.cuda(), d.cuda(), b.cuda().view(torch.uint8), aabbs, near, far, step, cone)
    two = ops.traverse_grids(*args)
    diag = float((aabbs[-1, 3:] - aabbs[-1, :3]).norm())
    one = ops.traverse_grids(*args, max_span=diag)
    tiny = ops.traverse_grids(*args, max_span=10 * step)          # bound far too small -> overflow -> fallback
    assert two[0].numel() > 100
    for got in (one, tiny):
        for a_, b_ in zip(got, two):
            assert torch.equal(a_, b_)


@pytest.mark.parametrize("step", [0.0034641, 2.0 ** -8, 2.0 ** -8 + 2.0 ** -21, 2.0 ** -7 + 2.0 ** -24, 0.01, 3e-4])
def test_traverse_vector_march_is_bit_exact_across_binades_and_ties(step):
    """Constant step: the 64-steps-at-once marcher (bit-pattern arithmetic progression per binade) must reproduce the
    serial float recurrence exactly -- long rays crossing many binades (t from 0.03 to ~28), step sizes whose sum with t is
    an exact rounding tie in some binades (2^-8 + 2^-21 at t in [8,16)), sparse and full grids, and the serial kernel
    (LSE_TRAVERSE_VEC=0 semantics are covered by the C oracle: both must agree with it)."""
    from oracle import sampling as osamp
    ops = _ops()
    R, levels, res = 96, 4, 32
    g = torch.Generator().manual_seed(int(step * 1e7) % 1000)
    o = (torch.rand(R, 3, generator=g) - 0.5) * 0.2
    d = torch.randn(R, 3, generator=g)
    d = d / d.norm(dim=-1, keepdim=True)
    d[:8] = torch.tensor([1.0, 0.0, 0.0])                        # axis-parallel rays: many samples per cell
    aabbs = torch.stack([osamp.enlarge_aabb(torch.tensor([-2.0, -2, -2, 2, 2, 2]), 2 ** i) for i in range(levels)])
    near = torch.full((R,), 0.03) + torch.rand(R, generator=g) * step
    far = torch.full((R,), 1e3)
    for frac in (1.0, 0.35):
        b = random_binaries(levels, res, frac, seed=7) if frac < 1 else torch.ones(levels, res, res, res, dtype=torch.bool)
        ri, ts, te, packed = osamp.traverse_grids(o, d, b, aabbs, near, far, float(step), 0.0)
        args = (o.cuda(), d.cuda(), b.cuda().view(torch.uint8), aabbs.cuda(), near.cuda(), far.cuda(), float(step), 0.0)
        for kw in ({}, {"max_span": float((aabbs[-1, 3:] - aabbs[-1, :3]).norm())}):
            if kw and ri.numel() / R * 1.0 > 3e5:
                continue
            hri, hts, hte, hpacked = ops.traverse_grids(*args, **kw)
            assert torch.equal(hpacked.cpu(), packed)
            assert torch.equal(hts.cpu(), ts) and torch.equal(hte.cpu(), te) and torch.equal(hri.cpu().long(), ri)
        assert float(ts.max()) > 16.0 and ri.numel() > 1000


def test_traverse_full_size_properties():
    """Metric size (4096 rays x 1024 samples, 4-level 128^3 grid, constant step): size-independent properties of the packed
    output -- exact per-ray counts, contiguous intervals (t_end[i] == t_start[i+1] bit for bit inside a ray), every
    interval exactly one float-add of the step long, ray-sorted indices -- and single pass == two-pass, vector == serial."""
    import os
    from oracle import sampling as osamp
    ops = _ops()
    R, S = 4096, 1024
    g = torch.Generator().manual_seed(11)
    o = (torch.rand(R, 3, generator=g) - 0.5).cuda()
    d = torch.randn(R, 3, generator=g)
    d = (d / d.norm(dim=-1, keepdim=True)).cuda()
    step = 2 * 3 ** 0.5 / 1000
    aabbs = torch.stack([osamp.enlarge_aabb(torch.tensor([-1.0, -1, -1, 1, 1, 1]), 2 ** i) for i in range(4)]).cuda()
    b = torch.ones(4, 128, 128, 128, dtype=torch.uint8).cuda()
    near = torch.full((R,), 0.05).cuda()
    far = torch.full((R,), 0.05 + S * step - 0.25 * step).cuda()
    diag = float((aabbs[-1, 3:] - aabbs[-1, :3]).norm())
    one = ops.traverse_grids(o, d, b, aabbs, near, far, step, 0.0, max_span=min(diag, float(far[0] - near[0])))
    two = ops.traverse_grids(o, d, b, aabbs, near, far, step, 0.0)
    for a_, b_ in zip(one, two):
        assert torch.equal(a_, b_)
    ri, ts, te, packed = one
    assert ri.numel() == R * S and bool((packed[:, 1] == S).all())
    assert torch.equal(ri.view(R, S), torch.arange(R, device="cuda", dtype=torch.int32)[:, None].expand(R, S))
    ts, te = ts.view(R, S), te.view(R, S)
    assert torch.equal(te[:, :-1], ts[:, 1:])                                 # contiguous inside a ray
    assert torch.equal(te, ts + torch.tensor(step, dtype=torch.float32, device="cuda"))   # one float add per step
    assert torch.equal(ts[:, 0], near)


def test_traverse_edge_cases():
    from oracle import sampling as osamp
    ops = _ops()
    aabbs = torch.tensor([[-1.0, -1, -1, 1, 1, 1]])
    # rays that miss, rays parallel to an axis, a ray starting exactly on a face, empty grid, full grid
    o = torch.tensor([[5.0, 5, 5], [-2.0, 0.1, 0.2], [0.0, 0.0, -3.0], [-1.0, 0.3, 0.3], [0.2, 0.2, 0.2]])
    d = torch.tensor([[1.0, 0, 0], [1.0, 0, 0], [0.0, 0, 1.0], [0.6, 0.8, 0.0], [-0.57735026, 0.57735026, 0.57735026]])
    near, far = torch.zeros(5), torch.full((5,), 1e10)
    for b in (torch.zeros(1, 8, 8, 8, dtype=torch.bool), torch.ones(1, 8, 8, 8, dtype=torch.bool)):
        ref = osamp.traverse_grids(o, d, b, aabbs, near, far, 0.05, 0.0)
        got = ops.traverse_grids(o.cuda(), d.cuda(), b.cuda().view(torch.uint8), aabbs.cuda(), near.cuda(), far.cuda(),
                                 0.05, 0.0)
        assert torch.equal(got[3].cpu(), ref[3])
        assert torch.equal(got[0].cpu().long(), ref[0]) and torch.equal(got[1].cpu(), ref[1]) and torch.equal(got[2].cpu(), ref[2])
    # zero rays
    e = torch.zeros(0, 3).cuda()
    got = ops.traverse_grids(e, e, b.cuda().view(torch.uint8), aabbs.cuda(), torch.zeros(0).cuda(), torch.zeros(0).cuda(), 0.05, 0.0)
    assert got[0].numel() == 0 and got[3].shape == (0, 2)


def test_visibility_and_compaction():
    from oracle import volrend as ovr
    ops = _ops()
    g = torch.Generator().manual_seed(2)
    lengths = [0, 1, 63, 64, 65, 200, 0, 1024, 3]
    cnt = torch.tensor(lengths)
    packed = torch.stack([torch.cumsum(cnt, 0) - cnt, cnt], -1)
    N = int(cnt.sum())
    ts = torch.rand(N, generator=g)
    te = ts + 0.01
    sig = torch.rand(N, generator=g) * 40
    ri = torch.repeat_interleave(torch.arange(len(lengths)), cnt).int()
    eps, thre = 1e-4, 0.05
    trans, alphas = ovr.render_transmittance_from_density(ts, te, sig, packed)
    ref = ovr.render_visibility_from_density(ts, te, sig, packed, eps, thre)
    o_ri, o_ts, o_te, new_packed, mask = ops.visibility_compact(ri.cuda(), ts.cuda(), te.cuda(), sig.cuda(), packed.cuda(), eps, thre)
    mask = mask.cpu().bool()
    # threshold comparisons on transcendental results: allow flips only within 1e-5 relative of the thresholds
    near_thr = ((trans - eps).abs() < 1e-5 * eps) | ((alphas - thre).abs() < 1e-5 * thre)
    assert bool(((mask == ref) | near_thr).all())
    assert torch.equal(o_ri.cpu(), ri[mask]) and torch.equal(o_ts.cpu(), ts[mask]) and torch.equal(o_te.cpu(), te[mask])
    assert torch.equal(new_packed.cpu()[:, 1], torch.zeros(len(lengths), dtype=torch.long).index_add_(0, ri[mask].long(), torch.ones(int(mask.sum()), dtype=torch.long)))


# ------------------------------------------------------------------------------------------------ positions
@pytest.mark.parametrize("contraction", [True, False])
def test_positions_fwd_bwd(contraction):
    from oracle import field as ofield
    ops = _ops()
    R, S = 64, 50
    o, d = random_rays(R, seed=3, inside=True)
    o = o * 3.0   # some samples beyond |x|>1 so the contraction branch is hit
    cnt = torch.full((R,), S)
    packed = torch.stack([torch.cumsum(cnt, 0) - cnt, cnt], -1)
    ri = torch.repeat_interleave(torch.arange(R), cnt)
    ts = torch.rand(R * S, generator=torch.Generator().manual_seed(4)) * 3
    te = ts + 0.01
    aabb = torch.tensor([[-1.5, -1.0, -2.0], [1.0, 2.0, 1.5]])
    fo = ofield.FieldOracle("tcnn", contraction=contraction, aabb=aabb, log2_hashmap_size=10, num_levels=4)
    oc, dc = o.clone().requires_grad_(True), d.clone().requires_grad_(True)
    pos = ofield.frustum_positions(oc[ri], dc[ri], ts[:, None], te[:, None])
    p_ref, sel_ref = fo.normalize(pos)
    og, dg = o.clone().cuda().requires_grad_(True), d.clone().cuda().requires_grad_(True)
    x01, sel = ops.positions(og, dg, ri.int().cuda(), ts.cuda(), te.cuda(), packed.cuda(), contraction,
                             None if contraction else aabb.flatten().tolist())
    if not contraction:   # (the contraction maps everything into the open unit cube)
        assert sel_ref.float().mean() > 0.1 and sel_ref.float().mean() < 0.999
    edge = (p_ref.detach() - 0.5).abs().max(-1).values > 0.49999   # in/out decisions within fp noise of the faces
    assert bool(((sel.cpu().bool() == sel_ref) | edge).all())
    keep = (sel.cpu().bool() == sel_ref)
    assert nmax_err(x01.cpu()[keep], p_ref[keep]) < TOL_FWD
    w = torch.rand(R * S, 3, generator=torch.Generator().manual_seed(5))
    w[~keep] = 0
    (p_ref * w).sum().backward()
    (x01 * w.cuda()).sum().backward()
    assert nmax_err(og.grad, oc.grad) < TOL_GRAD
    assert nmax_err(dg.grad, dc.grad) < TOL_GRAD


# ------------------------------------------------------------------------------------------------ hash grid
@pytest.mark.parametrize("L,T", [(16, 19), (4, 12), (16, 14)])
def test_hash_fwd_bwd(L, T):
    from oracle import hashgrid as ohg
    ops = _ops()
    N = 5000
    g = torch.Generator().manual_seed(L)
    x = torch.rand(N, 3, generator=g)
    x[:8] = torch.tensor([[0.0, 0, 0], [1.0, 1, 1], [0.5, 0.5, 0.5], [1.0, 0, 0.25], [0, 1.0, 0], [0.999999, 0.5, 0.1],
                          [1e-7, 1e-7, 1e-7], [0.25, 0.75, 1.0]])
    meta_o = ohg.tcnn_grid_meta(n_levels=L, log2_hashmap_size=T)
    meta = ops.make_grid_meta(n_levels=L, log2_hashmap_size=T)
    assert list(meta.offsets) == meta_o.offsets and list(meta.resolutions) == meta_o.resolutions
    table = (torch.rand(meta.n_params, generator=g) * 2 - 1) * 0.1
    tc = table.clone().requires_grad_(True)
    xc = x.clone().requires_grad_(True)
    y_ref = ohg.hash_encode_tcnn(xc, tc, meta_o)                       # [N, L*2]
    tg = table.clone().cuda().requires_grad_(True)
    xg = x.clone().cuda().requires_grad_(True)
    y = ops.hash_encode(xg, tg, meta)                                   # [L, N, 2]
    y_nl = y.permute(1, 0, 2).reshape(N, -1)
    assert nmax_err(y_nl, y_ref) < TOL_FWD
    w = torch.rand(N, L * 2, generator=g)
    (y_ref * w).sum().backward()
    (y_nl * w.cuda()).sum().backward()
    assert nmax_err(tg.grad, tc.grad) < TOL_GRAD
    # ... and level by level, each scaled by its own maximum, plus relative L2
    assert blockwise_nmax_err(tg.grad, tc.grad, hash_level_bounds(meta)) < TOL_GRAD_BLOCK
    assert rel_l2(tg.grad, tc.grad) < TOL_GRAD
    # d/dx: exclude points sitting on a cell boundary at some level (floor() flips are fp-order dependent)
    assert nmax_err(xg.grad[8:], xc.grad[8:]) < TOL_GRAD


def _ray_coherent_points(n_rays, per_ray, seed, step=2 * 3 ** 0.5 / 1000 / 4):
    """Unit-cube positions as the metric workload produces them: `per_ray` CONSECUTIVE samples of each ray at the auto step
    size (divided by 4: the field maps [-2,2] onto [0,1]), rays through the cube in random directions."""
    g = torch.Generator().manual_seed(seed)
    d = torch.nn.functional.normalize(torch.randn(n_rays, 3, generator=g), dim=-1)
    o = 0.5 - d * (step * per_ray / 2) + (torch.rand(n_rays, 3, generator=g) - 0.5) * 0.2
    t = step * (torch.arange(per_ray, dtype=torch.float32) + 0.5)
    x = (o[:, None, :] + d[:, None, :] * t[None, :, None]).reshape(-1, 3)
    return x.clamp(1e-6, 1 - 1e-6).contiguous()


def _hash_bwd_ex(ops, meta, x, dy, table, with_dx=True, **opt_kw):
    import ctypes
    from lsenerf_amd import _lib
    o = _lib.hash_bwd_default_opts()
    for k, v in opt_kw.items():
        assert hasattr(o, k), k
        setattr(o, k, v)
    dt = torch.zeros_like(table)
    dx = torch.empty_like(x) if with_dx else None
    desc = meta.desc()
    # the replica workspace this selection needs (zero on entry; the call must hand it back zeroed)
    nbytes = int(_lib.load().lse_hash_bwd_workspace_bytes(ctypes.byref(desc), ctypes.byref(o)))
    ws = torch.zeros(max(nbytes // 4, 1), dtype=torch.float32, device=x.device)
    if nbytes:
        o.workspace, o.workspace_bytes = ws.data_ptr(), nbytes
    P = lambda t_: ctypes.c_void_p(t_.data_ptr()) if t_ is not None else None
    _lib.call("lse_hash_bwd_ex", ctypes.byref(desc), P(x), P(dy), P(table), P(dt), P(dx), 0, 0, meta.n_levels, x.shape[0],
              None, ctypes.byref(o), ops._stream())
    assert not bool(ws.any()), "the replica workspace must read zero again after the call"
    return dt, dx


def test_hash_bwd_metric_regime_every_kernel_variant_vs_oracle():
    """The regime bench.py times: 64 rays x 1024 CONSECUTIVE samples each (16 waves per ray, runs of many lanes per cell on
    the coarse levels, ~2 cells per step on the finest), L = 16, T = 2^19.  The default kernel (run scan, cross-row carry,
    LDS sector cache, few-runs path, pair-lane collision atomics), its 64-B-line and second-probe variants and the
    16-lanes-per-sample kernel are selected through lse_hash_bwd_ex's CALL ARGUMENTS in one process and all compared with
    oracle/hashgrid.py, level by level."""
    from oracle import hashgrid as ohg
    ops = _ops()
    meta_o = ohg.tcnn_grid_meta(n_levels=16, log2_hashmap_size=19)
    meta = ops.make_grid_meta(n_levels=16, log2_hashmap_size=19)
    x = _ray_coherent_points(64, 1024, seed=3)
    N = x.shape[0]
    g = torch.Generator().manual_seed(4)
    table = (torch.rand(meta.n_params, generator=g) * 2 - 1) * 0.1
    w = torch.randn(N, 32, generator=g)
    tc, xc = table.clone().requires_grad_(True), x.clone().requires_grad_(True)
    (ohg.hash_encode_tcnn(xc, tc, meta_o) * w).sum().backward()
    dy = w.reshape(N, 16, 2).permute(1, 0, 2).contiguous().cuda()          # level-major, as the fused MLP hands it over
    xg, tg = x.cuda(), table.cuda()
    bounds = hash_level_bounds(meta)
    variants = {"default": {}, "unpaired_sectors": {"gran": 2}, "paired_probe2": {"gran": 4, "second_probe": 2}, "stage_all": {"stage_max": 64}, "stage_none": {"stage_max": 0}, "batched_probe0": {"second_probe": 0}, "batched_probe1": {"second_probe": 1}, "batched_probe5": {"second_probe": 5},
                "batched_no_few_runs": {"few_runs": 0}, "per_level_pass": {"impl": 1}, "line_cache": {"impl": 1, "gran": 3},
                "second_probe": {"impl": 1, "second_probe": 1}, "no_few_runs": {"impl": 1, "few_runs": 0},
                "lanes16": {"impl": 0}, "lanes16_r64": {"impl": 0, "rounds": 64},
                # round 3: the cache-free high-occupancy kernel takes the levels below `coarse_levels` (a split of the level
                # range between two launches; all of them = that kernel alone, incl. its multi-trip staging on the fine levels)
                "coarse6": {"coarse_levels": 6}, "coarse9": {"coarse_levels": 9}, "coarse_all": {"coarse_levels": 16},
                "coarse8_impl1": {"coarse_levels": 8, "impl": 1}, "coarse5_lanes16": {"coarse_levels": 5, "impl": 0},
                # second-generation flush of the paired sector cache (transposed list, read-and-zero exchange, bulk key reset)
                "flush2": {"gran": 6}, "flush2_coarse7": {"gran": 6, "coarse_levels": 7}, "flush2_stage_all": {"gran": 6, "stage_max": 64},
                "flush2_no_few_runs": {"gran": 6, "few_runs": 0, "second_probe": 0},
                # coarse-level replicas (default: 8 replicas of levels 0-3): off, more of them, every dense level, with the coarse kernel
                "no_replicas": {"replicas": 0}, "replicas8": {"replicas": 8}, "replicas32_levels6": {"replicas": 32, "replica_levels": 6},
                "replicas_coarse_kernel": {"coarse_levels": 7, "replicas": 16, "replica_levels": 5},
                "replicas_few_runs16": {"few_runs": 16, "replica_levels": 16, "replicas": 4},
                # next level's dy / table operands fetched before the current level's cache pass
                "prefetch": {"prefetch": 1}, "prefetch_no_few_runs": {"prefetch": 1, "few_runs": 0}, "prefetch_stage_all": {"prefetch": 1, "stage_max": 64},
                "prefetch_coarse5": {"prefetch": 1, "coarse_levels": 5},
                # pair-aligned flush lists
                "few_runs8": {"few_runs": 8}, "dense_steps_thresholds": {"few_runs": 3, "stage_max": 56}, "round5_first_thresholds": {"few_runs": 6, "stage_max": 32}, "few_runs2_stage_all": {"few_runs": 2, "stage_max": 64}, "round4_thresholds": {"few_runs": 6, "stage_max": 16},
                # 320 slots (52 KB of LDS per workgroup: three workgroups = 12 waves per CU; non-power-of-two slot arithmetic)
                # the compute half of two levels issued together (development build)
                "slots320": {"gran": 8}, "slots320_dense_steps": {"gran": 8, "few_runs": 6, "stage_max": 32}, "slots320_probe0": {"gran": 8, "second_probe": 0},
                "slots320_stage_all_no_few_runs": {"gran": 8, "stage_max": 64, "few_runs": 0}, "slots320_no_replicas": {"gran": 8, "replicas": 0},
                # the workgroup size sets the portions LDS is handed out in (production: 512 slots, ONE wave per workgroup):
                # 384 slots x 2 waves (10 waves per CU), 448 x 1 (9), 384 x 1, 512 x 4 (the shape until round 5), 512 x 2
                "slots384_wg2": {"gran": 9}, "slots384_wg2_dense_steps": {"gran": 9, "few_runs": 6, "stage_max": 32}, "slots384_wg2_stage_all_no_few_runs": {"gran": 9, "stage_max": 64, "few_runs": 0},
                "slots448_wg1": {"gran": 10}, "slots448_wg1_probe0_stage_all": {"gran": 10, "second_probe": 0, "stage_max": 64}, "slots384_wg1": {"gran": 11}, "slots512_wg4": {"gran": 12}, "slots512_wg4_dense_steps": {"gran": 12, "few_runs": 6, "stage_max": 32}, "slots512_wg2": {"gran": 13},
                "aligned_pairs": {"gran": 7}, "aligned_pairs_stage_all": {"gran": 7, "stage_max": 64}, "aligned_pairs_probe0": {"gran": 7, "second_probe": 0}}
    # floor() decisions of samples that sit within rounding of a cell face may differ between the two position formulas
    on_face = torch.zeros(N, dtype=torch.bool)
    for sc in meta.scales:
        p = x.double() * sc + 0.5
        on_face |= ((p - p.round()).abs() < 1e-4).any(-1)
    from lsenerf_amd import _lib

    def check(name, kw):
        dt, dx = _hash_bwd_ex(ops, meta, xg, dy, tg, **kw)
        assert nmax_err(dt, tc.grad) < TOL_GRAD, name
        blk = blockwise_nmax_err(dt, tc.grad, bounds)
        assert blk < TOL_GRAD_BLOCK, (name, blk)
        assert rel_l2(dt, tc.grad) < TOL_GRAD, name
        assert nmax_err(dx.cpu()[~on_face], xc.grad[~on_face]) < TOL_GRAD, name

    # The library that ships holds the production kernels only (ABI 5): the batched sector-cache kernel with the second-generation
    # flush (impl 2, gran 6) under any staging / probing / few-runs / replica setting, and the generic 16-lanes-per-sample kernel.
    # Every other selection is a development variant: rejected there, held against the oracle in liblse_hip_dev.so.
    shipped = lambda kw: (kw.get("impl", 2) == 0 or (kw.get("impl", 2) == 2 and kw.get("gran", 6) == 6 and not kw.get("prefetch"))) \
        and not kw.get("coarse_levels") and kw.get("rounds", 32) == 32
    n_shipped = 0
    for name, kw in variants.items():
        if shipped(kw):
            check(name, kw)
            n_shipped += 1
        else:
            with pytest.raises(_lib.LseHipError, match="development variant"):
                _hash_bwd_ex(ops, meta, xg, dy, tg, **kw)
    assert n_shipped >= 16
    assert _lib.dev_available(), "liblse_hip_dev.so is built by __graft_entry__.build() (make -C lsenerf_amd/csrc dev)"
    with _lib.dev_library():
        for name, kw in variants.items():
            check("dev:" + name, kw)


def test_hash_bwd_chunk_mapping_covers_every_sample_count():
    """The default hash backward hands workgroup b the chunk (b % 8) * ceil(B / 8) + b / 8 of the B chunks the samples fill (every
    XCD walks one contiguous eighth) and launches a multiple of 8 workgroups over the CAPACITY: sample counts around the chunk
    (64) and 8-chunk borders (and around 256 / 2048, the borders of the measured four-chunks-per-wave variant), and device-side
    counts anywhere between 0 and the capacity, must each be covered exactly once -- checked against the oracle (host counts) and
    against the same call on the truncated arrays (device counts)."""
    import ctypes
    from oracle import hashgrid as ohg
    from lsenerf_amd import _lib
    ops = _ops()
    meta_o = ohg.tcnn_grid_meta(n_levels=16, log2_hashmap_size=15)
    meta = ops.make_grid_meta(n_levels=16, log2_hashmap_size=15)
    g = torch.Generator().manual_seed(12)
    table = (torch.rand(meta.n_params, generator=g) * 2 - 1) * 0.1
    cap = 256 * 8 * 2 + 200
    x_all = _ray_coherent_points(4, (cap + 3) // 4, seed=5)[:cap].contiguous()
    w_all = torch.randn(cap, 32, generator=g)
    tg = table.cuda()
    for n in (1, 63, 64, 65, 255, 256, 257, 64 * 7 + 1, 64 * 8, 64 * 8 + 1, 64 * 9 - 1, 64 * 16 + 17, 256 * 8 - 1, 256 * 8, 256 * 8 + 1,
              256 * 9 + 65, cap):
        x, w = x_all[:n].contiguous(), w_all[:n].contiguous()
        tc = table.clone().requires_grad_(True)
        (ohg.hash_encode_tcnn(x, tc, meta_o) * w).sum().backward()
        dy = w.reshape(n, 16, 2).permute(1, 0, 2).contiguous().cuda()
        dt, _ = _hash_bwd_ex(ops, meta, x.cuda(), dy, tg)
        assert nmax_err(dt, tc.grad) < TOL_GRAD, n
        assert rel_l2(dt, tc.grad) < TOL_GRAD, n
    # device-side counts: arrays of capacity extent, dy with the capacity as its level stride
    xg = x_all.cuda()
    dy_cap = w_all.reshape(cap, 16, 2).permute(1, 0, 2).contiguous().cuda()
    desc = meta.desc()
    P = lambda t_: ctypes.c_void_p(t_.data_ptr()) if t_ is not None else None
    for n_dev in (0, 1, 64, 65, 255, 256, 257, 64 * 8 - 1, 64 * 8, 64 * 8 + 1, 1000, 256 * 8 - 1, 256 * 8, 256 * 8 + 1, 256 * 11 + 3,
                  cap - 1, cap, cap + 77):
        cnt = torch.tensor([n_dev], dtype=torch.int64, device="cuda")
        dt = torch.zeros_like(tg); dx = torch.full_like(xg, 7.0)
        _lib.call("lse_hash_bwd", ctypes.byref(desc), P(xg), P(dy_cap), P(tg), P(dt), P(dx), cap, P(cnt), ops._stream())
        m = min(n_dev, cap)
        if m == 0:
            assert not bool(dt.any())
            continue
        dt_ref, dx_ref = _hash_bwd_ex(ops, meta, xg[:m].contiguous(), dy_cap[:, :m].contiguous(), tg)
        assert nmax_err(dt, dt_ref) < 1e-5, n_dev          # same kernel, other launch extent: summation order only
        assert torch.allclose(dx[:m], dx_ref, rtol=1e-5, atol=1e-7), n_dev
        assert bool((dx[m:] == 7.0).all()), n_dev          # nothing beyond the device-side count is written


def test_hash_full_size_forward_subset_and_backward_linearity():
    """BASELINE size: N = 2^22 ray-coherent samples, T = 2^19.  The oracle evaluates a 2^16-sample random subset of the
    forward and of d(x); the table gradient is checked through linearity of the encoding in the table:
    <dtable, dt> == <dy, encode(x; dt)> for random dt (fp64 inner products), level by level."""
    from oracle import hashgrid as ohg
    ops = _ops()
    meta_o = ohg.tcnn_grid_meta(n_levels=16, log2_hashmap_size=19)
    meta = ops.make_grid_meta(n_levels=16, log2_hashmap_size=19)
    x = _ray_coherent_points(4096, 1024, seed=7).cuda()
    N = x.shape[0]
    assert N == 1 << 22
    g = torch.Generator(device="cuda").manual_seed(8)
    table = (torch.rand(meta.n_params, generator=g, device="cuda") * 2 - 1) * 0.1
    y = ops.hash_encode(x, table, meta)                                   # [16, N, 2]
    sub = torch.randperm(N, generator=torch.Generator().manual_seed(9))[: 1 << 16]
    xs = x[sub.cuda()].cpu().requires_grad_(True)
    tcpu = table.cpu()
    y_ref = ohg.hash_encode_tcnn(xs, tcpu, meta_o)
    y_sub = y[:, sub.cuda(), :].permute(1, 0, 2).reshape(len(sub), -1)
    assert nmax_err(y_sub, y_ref) < TOL_FWD
    for l in range(16):                                                   # every level separately: coarse ones are 100x larger
        assert nmax_err(y_sub[:, 2 * l:2 * l + 2], y_ref[:, 2 * l:2 * l + 2]) < TOL_FWD, l
    dy = torch.randn(16, N, 2, generator=g, device="cuda")
    dt, dx = _hash_bwd_ex(ops, meta, x, dy, table)
    # d(x) on the subset against the oracle (a sample's d(x) depends on that sample only)
    w_sub = dy[:, sub.cuda(), :].permute(1, 0, 2).reshape(len(sub), -1).cpu()
    (y_ref * w_sub).sum().backward()
    on_face = torch.zeros(len(sub), dtype=torch.bool)
    for sc in meta.scales:
        p = xs.detach().double() * sc + 0.5
        on_face |= ((p - p.round()).abs() < 1e-4).any(-1)
    assert nmax_err(dx[sub.cuda()].cpu()[~on_face], xs.grad[~on_face]) < TOL_GRAD
    # table gradient: adjoint identity, per level (J is block diagonal over levels)
    bounds = hash_level_bounds(meta)
    for trial in range(2):
        dtab = torch.randn(meta.n_params, generator=g, device="cuda")
        jd = ops.hash_encode(x, dtab, meta)                               # J dt, level-major
        for l in range(16):
            lhs = float((dt[bounds[l]:bounds[l + 1]].double() * dtab[bounds[l]:bounds[l + 1]].double()).sum())
            rhs = float((dy[l].double() * jd[l].double()).sum())
            scale = float(dt[bounds[l]:bounds[l + 1]].double().norm() * dtab[bounds[l]:bounds[l + 1]].double().norm())
            assert abs(lhs - rhs) <= 1e-5 * scale, (trial, l, lhs, rhs, scale)


def test_hash_bwd_level_ranges_add_up():
    """lse_hash_bwd_levels: two launches over disjoint level ranges (the second accumulating d(x)) == one full launch."""
    import ctypes
    from lsenerf_amd import _lib
    ops = _ops()
    meta = ops.make_grid_meta()
    g = torch.Generator().manual_seed(5)
    N = 3000
    t = torch.linspace(0, 1, N)[:, None]
    x = (0.1 + 0.8 * t * torch.tensor([[0.9, 0.5, 0.3]]) + 0.01 * torch.rand(N, 3, generator=g)).clamp(0, 1).cuda()   # ray-like
    table = ((torch.rand(meta.n_params, generator=g) * 2 - 1) * 0.1).cuda()
    dy = torch.randn(meta.n_levels, N, 2, generator=g).cuda()
    desc = meta.desc()
    P = lambda t_: ctypes.c_void_p(t_.data_ptr())
    full_t, full_x = torch.zeros_like(table), torch.empty_like(x)
    _lib.call("lse_hash_bwd", ctypes.byref(desc), P(x), P(dy), P(table), P(full_t), P(full_x), N, None, ops._stream())
    part_t, part_x = torch.zeros_like(table), torch.full_like(x, 7.0)     # first launch overwrites d(x)
    _lib.call("lse_hash_bwd_levels", ctypes.byref(desc), P(x), P(dy), P(table), P(part_t), P(part_x), 0, 6, 16, N, None, ops._stream())
    lo = 2 * meta.offsets[6]
    assert float(part_t[:lo].abs().max()) == 0.0 and float(part_t[lo:].abs().max()) > 0
    _lib.call("lse_hash_bwd_levels", ctypes.byref(desc), P(x), P(dy), P(table), P(part_t), P(part_x), 1, 0, 6, N, None, ops._stream())
    assert nmax_err(part_t, full_t) < TOL_GRAD and nmax_err(part_x, full_x) < TOL_GRAD
    with pytest.raises(_lib.LseHipError):
        _lib.call("lse_hash_bwd_levels", ctypes.byref(desc), P(x), P(dy), P(table), P(part_t), P(part_x), 0, 9, 3, N, None, ops._stream())


def test_hash_partition_of_unity_and_linearity_full_size():
    """Size-independent properties at the metric size (N = 4096 x 1024 would take 0.5 GB of features; 2^20 here and the
    bench covers the full N): a constant table encodes to that constant; the encoding is linear in the table."""
    ops = _ops()
    N = 1 << 20
    meta = ops.make_grid_meta()
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.rand(N, 3, device="cuda", generator=g)
    ones = torch.ones(meta.n_params, device="cuda")
    y = ops.hash_encode(x, ones, meta)
    assert float((y - 1).abs().max()) < 1e-5
    t1 = torch.randn(meta.n_params, device="cuda", generator=g)
    t2 = torch.randn(meta.n_params, device="cuda", generator=g)
    y12 = ops.hash_encode(x, 2.0 * t1 - 0.5 * t2, meta)
    lin = 2.0 * ops.hash_encode(x, t1, meta) - 0.5 * ops.hash_encode(x, t2, meta)
    assert float((y12 - lin).abs().max()) < 2e-5


# ------------------------------------------------------------------------------------------------ MLP
def _mlp_case(in_dim, width, num_layers, out_dim, act, level_major, N, use_bias, seed=0, arith=0):
    from oracle.field import TcnnMLP
    from lsenerf_amd import _lib
    ops = _ops()
    g = torch.Generator().manual_seed(seed)
    om = TcnnMLP(in_dim, num_layers, width, out_dim, act)
    om.in_pad = in_dim            # kernel-level test: the kernel's n_in is already the padded width (or the 8-wide
    om.shapes[0] = (width, in_dim)  # level-major case whose tcnn ones-padding the host folds into a bias)
    om.n_params = sum(a * b for a, b in om.shapes)
    params = om.init_params(g)
    x = torch.randn(N, in_dim, generator=g)
    R = 37
    ridx = torch.sort(torch.randint(0, R, (N,), generator=g)).values
    cnt = torch.bincount(ridx, minlength=R)
    packed = torch.stack([torch.cumsum(cnt, 0) - cnt, cnt], -1)
    bias = torch.randn(R, width, generator=g) * 0.3 if use_bias else None
    pc, xc = params.clone().requires_grad_(True), x.clone().requires_grad_(True)
    bc = bias.clone().requires_grad_(True) if use_bias else None
    # oracle with the bias injected before the first ReLU
    Ws = om.matrices(pc)
    h = xc @ Ws[0].t()
    if use_bias:
        h = h + bc[ridx]
    h = torch.relu(h)
    for W in Ws[1:-1]:
        h = torch.relu(h @ W.t())
    out_ref = h @ Ws[-1].t()
    if act == "Sigmoid":
        out_ref = torch.sigmoid(out_ref)
    meta = ops.MlpMeta(in_dim, width, num_layers - 1, _lib.LSE_ACT_SIGMOID if act == "Sigmoid" else _lib.LSE_ACT_NONE,
                       _lib.LSE_IN_LEVELMAJOR if level_major else _lib.LSE_IN_ROWMAJOR, arith=arith)
    pg = params.clone().cuda().requires_grad_(True)
    if level_major:
        xin = x.view(N, in_dim // 2, 2).permute(1, 0, 2).contiguous().cuda().requires_grad_(True)
    else:
        xin = x.clone().cuda().requires_grad_(True)
    bg = bias.clone().cuda().requires_grad_(True) if use_bias else None
    out = ops.fused_mlp(pg, xin, meta, N, bg, ridx.int().cuda() if use_bias else None, packed.cuda() if use_bias else None)
    assert out.shape == (N, 16)
    assert nmax_err(out, out_ref) < TOL_FWD
    w = torch.randn(N, 16, generator=g)
    (out_ref * w).sum().backward()
    (out * w.cuda()).sum().backward()
    assert nmax_err(pg.grad, pc.grad) < TOL_GRAD
    gx = xin.grad.permute(1, 0, 2).reshape(N, in_dim) if level_major else xin.grad
    assert nmax_err(gx, xc.grad) < TOL_GRAD
    if use_bias:
        assert nmax_err(bg.grad, bc.grad) < TOL_GRAD


@pytest.mark.parametrize("N", [1, 63, 64, 1000, 4133])
def test_mlp_base_shape(N):
    _mlp_case(32, 64, 2, 16, None, True, N, False)


@pytest.mark.parametrize("N", [5, 4133])
def test_mlp_head_shape_with_ray_bias(N):
    _mlp_case(16, 64, 3, 16, "Sigmoid", False, N, True)


@pytest.mark.parametrize("fused_wgrad,tiled,fused_bias", [(True, False, True), (True, True, False), (False, False, False)])
def test_mlp_backward_variants_agree(monkeypatch, fused_wgrad, tiled, fused_bias):
    """The in-library cross-checks: row-major saved activations, d_act0 + lse_segment_sum_rows, unfused lse_mlp_wgrad."""
    ops = _ops()
    monkeypatch.setattr(ops, "FUSED_WGRAD", fused_wgrad)
    monkeypatch.setattr(ops, "ACT_TILED", tiled)
    monkeypatch.setattr(ops, "FUSED_BIAS_GRAD", fused_bias)
    _mlp_case(16, 64, 3, 16, "Sigmoid", False, 2051, True, seed=3)
    _mlp_case(32, 64, 2, 16, None, True, 2051, False, seed=4)


@pytest.mark.parametrize("recompute", [True, False])
def test_mlp_first_hidden_layer_recomputed_or_saved(monkeypatch, recompute):
    """Two hidden layers on a row-major input (the head): by default the forward does not save the first hidden layer and
    the backward recomputes it (act_tiled = 2); both routes against the oracle, with and without a per-row bias, ragged N."""
    ops = _ops()
    monkeypatch.setattr(ops, "RECOMPUTE_FIRST_LAYER", recompute)
    for n in (31, 2051):
        _mlp_case(16, 64, 3, 16, "Sigmoid", False, n, True, seed=5)
        _mlp_case(16, 64, 3, 16, None, False, n, False, seed=6)
    _mlp_case(64, 64, 3, 16, "Sigmoid", False, 1000, True, seed=7)      # nerfstudio-style 64-wide head input
    _mlp_case(16, 32, 3, 16, None, False, 777, True, seed=8)


@pytest.mark.parametrize("recompute_all,f32_mfma_fwd", [(True, False), (True, True), (False, False)])
def test_mlp_third_generation_bf16_pieces(monkeypatch, recompute_all, f32_mfma_fwd):
    """lsenerf_amd/csrc/mlp_x6.h: f32 operands cut into three bf16 pieces, six piece products per multiply on the bf16 matrix
    cores (f32-equivalent error bound -> the SAME tolerances as the f32-MFMA kernels), forward (lse_mlp_desc.arith = AUTO; the
    f32-MFMA forward through arith = LSE_MLP_ARITH_F32_MFMA, a call argument since ABI 5) and the backward that recomputes every
    hidden layer (act_tiled = 3), head and base shapes, ragged sizes, rows of every length."""
    from lsenerf_amd import _lib
    ops = _ops()
    monkeypatch.setattr(ops, "RECOMPUTE_ALL", recompute_all)
    arith = _lib.LSE_MLP_ARITH_F32_MFMA if f32_mfma_fwd else _lib.LSE_MLP_ARITH_AUTO
    for n in (1, 17, 31, 32, 33, 2051, 9000):
        _mlp_case(16, 64, 3, 16, "Sigmoid", False, n, True, seed=20 + n, arith=arith)
        _mlp_case(16, 64, 3, 16, None, False, n, False, seed=21 + n, arith=arith)
        _mlp_case(32, 64, 2, 16, None, True, n, False, seed=22 + n, arith=arith)
    _mlp_case(32, 64, 2, 16, None, False, 1500, False, seed=23, arith=arith)      # row-major 32-wide input


@pytest.mark.parametrize("head", [True, False])
def test_bf16_piece_arithmetic_has_the_error_level_of_f32(head):
    """The claim behind dtype "f32" in bench.py: the third-generation forward (three bf16 pieces per operand, six piece products per
    multiply, f32 accumulate) is NOT a reduced-precision path.  Both arithmetic routes against a float64 evaluation of the same
    network on the same inputs: the bf16-piece route must be within f32 rounding (a few 1e-7 of the output scale) and no worse than
    twice the f32-MFMA route + 1e-7."""
    import ctypes
    from lsenerf_amd import _lib
    ops = _ops()
    g = torch.Generator().manual_seed(31)
    N, R = 20000, 40
    if head:
        meta = ops.MlpMeta(16, 64, 2, _lib.LSE_ACT_SIGMOID, _lib.LSE_IN_ROWMAJOR)
        params = torch.randn(16 * 64 + 64 * 64 + 16 * 64, generator=g) * 0.2
        x = torch.randn(N, 16, generator=g)
        bias = torch.randn(R, 64, generator=g) * 0.3
        ridx = (torch.arange(N) * R // N).int()
        h = torch.relu(x.double() @ params[:1024].view(64, 16).double().T + bias.double()[ridx.long()])
        h = torch.relu(h @ params[1024:5120].view(64, 64).double().T)
        ref = torch.sigmoid(h @ params[5120:].view(16, 64).double().T)
        xin = x.cuda()
    else:
        meta = ops.MlpMeta(32, 64, 1, _lib.LSE_ACT_NONE, _lib.LSE_IN_LEVELMAJOR)
        params = torch.randn(32 * 64 + 16 * 64, generator=g) * 0.2
        x = torch.randn(N, 32, generator=g)
        bias = ridx = None
        h = torch.relu(x.double() @ params[:2048].view(64, 32).double().T)
        ref = h @ params[2048:].view(16, 64).double().T
        xin = x.view(N, 16, 2).permute(1, 0, 2).contiguous().cuda()
    import dataclasses
    P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    pc, bc, rc = params.cuda(), (bias.cuda() if head else None), (ridx.cuda() if head else None)
    errs = {}
    for impl, arith in ((1, _lib.LSE_MLP_ARITH_F32_MFMA), (2, _lib.LSE_MLP_ARITH_AUTO)):     # the route is part of the descriptor
        desc = dataclasses.replace(meta, arith=arith).desc()
        out = torch.empty(N, 16, device="cuda")
        _lib.call("lse_mlp_fwd", ctypes.byref(desc), P(pc), P(xin), P(bc), P(rc), P(out), 16, None, 1, None, None, 0.0, N, None, ops._stream())
        errs[impl] = float((out.double().cpu() - ref).abs().max() / ref.abs().max())
    assert errs[1] < 1e-6 and errs[2] < 1e-6, errs
    assert errs[2] <= 2 * errs[1] + 1e-7, errs


def test_bf16_piece_arithmetic_on_extreme_and_non_finite_inputs():
    """The bf16-piece forward at the ends of the exponent range and on non-finite inputs (base shape: 32 inputs, ReLU, linear
    output -- positively homogeneous, so f(s x) / s must equal f(x)).
      * inputs scaled by 2^60 and by 2^-60: bf16 has f32's exponent range, the three pieces of a value span 2^-16 of its
        magnitude, so neither end loses a piece: the relative error against float64 stays at the f32 level (< 1e-6), as unscaled;
      * one input row holding +inf, one holding NaN: the remainder of a piece is formed on the matrix core as D = C - I * hi, so an
        infinite input becomes inf - inf = NaN in every hidden pre-activation of its row, and the kernels' ReLU is a maximum with 0,
        which returns the non-NaN operand: the row's hidden layer is all zeros and (bias-free network) so is its output.  MEASURED
        AND STATED, not wanted: tcnn's ReLU (`x * (x > 0)`) would hand NaN / inf on to the output.  Non-finite inputs therefore come
        back as a ZERO row -- silently -- while EVERY other row is bit-identical to the run without them (a matrix-core column
        never mixes samples).  The path's MLP inputs are hash features and clamped base outputs, which are finite by construction."""
    import ctypes
    from lsenerf_amd import _lib
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    N = 4096 + 37
    meta = ops.MlpMeta(32, 64, 1, _lib.LSE_ACT_NONE, _lib.LSE_IN_LEVELMAJOR)
    params = (torch.randn(32 * 64 + 16 * 64, generator=g) * 0.2).cuda()
    x = torch.randn(N, 32, generator=g)
    desc = meta.desc()
    P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None

    def run(xx):
        xin = xx.view(N, 16, 2).permute(1, 0, 2).contiguous().cuda()
        out = torch.empty(N, 16, device="cuda")
        _lib.call("lse_mlp_fwd", ctypes.byref(desc), P(params), P(xin), None, None, P(out), 16, None, 1, None, None, 0.0, N, None, ops._stream())
        return out.cpu()
    pd = params.double().cpu()
    ref = torch.relu(x.double() @ pd[:2048].view(64, 32).T) @ pd[2048:].view(16, 64).T
    base = run(x)
    assert float((base.double() - ref).abs().max() / ref.abs().max()) < 1e-6
    for e in (60, -60):
        sc = float(2.0 ** e)
        out = run(x * sc)
        assert torch.isfinite(out).all(), e
        err = float((out.double() / sc - ref).abs().max() / ref.abs().max())
        assert err < 1e-6, (e, err)
    bad = x.clone()
    bad[100, 3] = float("inf")
    bad[2000, 17] = float("nan")
    out = run(bad)
    assert float(out[100].abs().max()) == 0.0 and float(out[2000].abs().max()) == 0.0
    keep = torch.ones(N, dtype=torch.bool)
    keep[100] = keep[2000] = False
    assert torch.equal(out[keep], base[keep])


@pytest.mark.parametrize("pattern", ["short", "mixed", "tile_aligned", "long"])
def test_mlp_head_view_bias_column_row_patterns(pattern):
    """The head as the field calls it (first-layer view: leading dimension 64, column offset 15, column 0 masked, compact 4-column
    output): the per-row bias gradient rides in the free column of the dW0 product of the recomputing backward.  One product
    covers two 16-sample column tiles, so only one row per 32-sample tile may use the column and every other row of the tile
    takes the segmented scan -- exercised here with rows of 0..3, 0..100, exactly 16 / 32 / 48 and ~1000 samples.  Third- and
    second-generation backward must agree to summation noise; both against a float64 network (relative L2: single ReLU-gate
    flips are allowed to exist, see tests/util.py)."""
    from lsenerf_amd import _lib
    ops = _ops()
    g = torch.Generator().manual_seed({"short": 1, "mixed": 2, "tile_aligned": 3, "long": 4}[pattern])
    R = 400
    if pattern == "short":
        cnt = torch.randint(0, 4, (R,), generator=g)
    elif pattern == "mixed":
        cnt = torch.randint(0, 101, (R,), generator=g)
    elif pattern == "tile_aligned":
        cnt = torch.tensor([16, 32, 48, 0, 16, 16, 32])[torch.randint(0, 7, (R,), generator=g)]
    else:
        cnt = torch.randint(900, 1100, (40,), generator=g)
        R = 40
    N = int(cnt.sum())
    ridx = torch.repeat_interleave(torch.arange(R), cnt)
    packed = torch.stack([torch.cumsum(cnt, 0) - cnt, cnt], -1)
    meta = ops.MlpMeta(16, 64, 2, _lib.LSE_ACT_SIGMOID, _lib.LSE_IN_ROWMAJOR, 64, 15, 1)
    params = torch.randn(64 * 64 + 64 * 64 + 16 * 64, generator=g) * 0.15
    x = torch.randn(N, 16, generator=g)
    bias = torch.randn(R, 64, generator=g) * 0.3
    w = torch.randn(N, 4, generator=g)
    got = {}
    for gen3 in (True, False):
        ops.RECOMPUTE_ALL = gen3
        try:
            p = params.clone().cuda().requires_grad_(True)
            xc = x.clone().cuda().requires_grad_(True)
            b = bias.clone().cuda().requires_grad_(True)
            out = ops.fused_mlp(p, xc, meta, N, b, ridx.int().cuda(), packed.cuda(), out_cols=4)
            (out * w.cuda()).sum().backward()
        finally:
            ops.RECOMPUTE_ALL = True
        got[gen3] = (out.detach().cpu(), p.grad.cpu(), xc.grad.cpu(), b.grad.cpu())
    for a, c in zip(got[True], got[False]):
        assert nmax_err(a, c) < 1e-5
    assert torch.all(got[True][3][cnt == 0] == 0)
    # float64 network
    p64 = params.double().requires_grad_(True)
    x64, b64 = x.double().requires_grad_(True), bias.double().requires_grad_(True)
    msk = torch.ones(16, dtype=torch.float64)
    msk[0] = 0
    h = torch.relu(x64 @ (p64[:4096].view(64, 64)[:, 15:31] * msk).T + b64[ridx])
    h = torch.relu(h @ p64[4096:8192].view(64, 64).T)
    o = torch.sigmoid(h @ p64[8192:].view(16, 64).T)[:, :4]
    (o * w.double()).sum().backward()
    from tests.util import rel_l2
    assert nmax_err(got[True][0], o.detach()) < TOL_FWD
    for a, c in zip(got[True][1:], (p64.grad, x64.grad, b64.grad)):
        assert rel_l2(a, c) < 10 * TOL_GRAD       # (a few thousand samples: one flipped ReLU gate is 1e-3 of the L2 norm)


def test_mlp_row_bias_grad_many_short_rows():
    """Rows of 0..3 samples: several row boundaries inside every 16-sample tile of the fused bias-gradient scan."""
    ops = _ops()
    from lsenerf_amd import _lib
    g = torch.Generator().manual_seed(11)
    N, R, W = 3001, 2000, 64
    ridx = torch.sort(torch.randint(0, R, (N,), generator=g)).values
    cnt = torch.bincount(ridx, minlength=R)
    packed = torch.stack([torch.cumsum(cnt, 0) - cnt, cnt], -1)
    meta = ops.MlpMeta(16, W, 2, _lib.LSE_ACT_NONE, _lib.LSE_IN_ROWMAJOR)
    params = (torch.randn(16 * W + W * W + W * 16, generator=g) * 0.2)
    x = torch.randn(N, 16, generator=g)
    bias = torch.randn(R, W, generator=g)
    grads = []
    for fused in (True, False):
        ops.FUSED_BIAS_GRAD = fused
        try:
            b = bias.clone().cuda().requires_grad_(True)
            out = ops.fused_mlp(params.cuda(), x.cuda(), meta, N, b, ridx.int().cuda(), packed.cuda())
            (out * out).sum().backward()
        finally:
            ops.FUSED_BIAS_GRAD = True
        grads.append(b.grad.cpu())
    # reference: torch autograd on the same maths
    Wm = [params[:16 * W].view(W, 16), params[16 * W:16 * W + W * W].view(W, W), params[16 * W + W * W:].view(16, W)]
    bc = bias.clone().requires_grad_(True)
    h = torch.relu(x @ Wm[0].t() + bc[ridx]); h = torch.relu(h @ Wm[1].t()); o = h @ Wm[2].t()
    (o * o).sum().backward()
    assert nmax_err(grads[0], bc.grad) < TOL_GRAD and nmax_err(grads[1], bc.grad) < TOL_GRAD
    assert torch.all(grads[0][cnt == 0] == 0)


def test_mlp_config1_shapes():
    _mlp_case(8, 32, 2, 16, None, True, 777, False)          # L=4 hash grid -> 2x32 base MLP (BASELINE config 1)
    _mlp_case(16, 32, 3, 16, "Sigmoid", False, 777, True)
    _mlp_case(32, 64, 2, 16, None, False, 500, False)         # row-major 32-wide input
    _mlp_case(64, 64, 3, 16, "Sigmoid", False, 500, False)    # the un-split 64-wide head input


# ------------------------------------------------------------------------------------------------ SH / per-ray features
@pytest.mark.parametrize("emb_dim", [8, 16, 32, 48, 97])
def test_ray_bias_takes_other_embedding_widths(emb_dim):
    """lse_ray_bias_fwd / _bwd with appearance embeddings other than the reference's 32 columns (padded head input 48 .. 128):
    row_bias = [SH16 | 0 x 15 | emb[idx] | ones-padding] W_in^T and its gradients w.r.t. directions, embedding rows and W_in against
    float64 autograd."""
    from oracle.field import sh4_tcnn
    ops = _ops()
    R, width = 300, 64
    in_pad = (31 + emb_dim + 15) // 16 * 16
    _, d = random_rays(R, seed=emb_dim)
    g = torch.Generator().manual_seed(emb_dim)
    emb = torch.randn(5, emb_dim, generator=g)
    idx = torch.randint(0, 5, (R,), generator=g)
    head = torch.randn(width * in_pad + 2 * width * width, generator=g) * 0.3      # W_in first, the rest of the head's parameters behind it
    gout = torch.randn(R, width, generator=g)
    dc, ec, wc = (t.double().clone().requires_grad_(True) for t in (d, emb, head))
    feat = torch.cat([sh4_tcnn((dc + 1) / 2), torch.zeros(R, 15, dtype=torch.float64), ec[idx],
                      torch.ones(R, in_pad - 31 - emb_dim, dtype=torch.float64)], -1)
    ref = feat @ wc[: width * in_pad].view(width, in_pad).t()
    (ref * gout.double()).sum().backward()
    dg, eg, wg = (t.clone().cuda().requires_grad_(True) for t in (d, emb, head))
    out = ops.ray_bias(dg, eg, idx.int().cuda(), wg, width)
    assert out.shape == (R, width) and nmax_err(out, ref.float()) < TOL_FWD
    (out * gout.cuda()).sum().backward()
    assert nmax_err(dg.grad, dc.grad.float()) < TOL_GRAD
    assert nmax_err(eg.grad, ec.grad.float()) < TOL_GRAD
    assert nmax_err(wg.grad, wc.grad.float()) < TOL_GRAD
    assert float(wg.grad[width * in_pad:].abs().max()) == 0.0


def test_ray_features_and_linear():
    from oracle.field import sh4_tcnn
    ops = _ops()
    R = 500
    _, d = random_rays(R, seed=9)
    g = torch.Generator().manual_seed(9)
    emb = torch.randn(7, 32, generator=g)
    idx = torch.randint(0, 7, (R,), generator=g)
    dc, ec = d.clone().requires_grad_(True), emb.clone().requires_grad_(True)
    ref = torch.cat([sh4_tcnn((dc + 1) / 2), torch.zeros(R, 15), ec[idx], torch.ones(R, 1)], -1)
    dg, eg = d.clone().cuda().requires_grad_(True), emb.clone().cuda().requires_grad_(True)
    feat = ops.ray_features(dg, eg, idx.int().cuda())
    assert nmax_err(feat, ref) < TOL_FWD
    W = torch.randn(64, 64, generator=g) * 0.2
    Wc, Wg = W.clone().requires_grad_(True), W.clone().cuda().requires_grad_(True)
    y_ref = ref @ Wc.t()
    y = ops.linear(feat, Wg)
    assert nmax_err(y, y_ref) < TOL_FWD
    w = torch.randn(R, 64, generator=g)
    (y_ref * w).sum().backward()
    (y * w.cuda()).sum().backward()
    assert nmax_err(Wg.grad, Wc.grad) < TOL_GRAD
    assert nmax_err(dg.grad, dc.grad) < TOL_GRAD
    assert nmax_err(eg.grad, ec.grad) < TOL_GRAD


# ------------------------------------------------------------------------------------------------ volume rendering
def test_volrend_fwd_bwd_ragged():
    from oracle import volrend as ovr
    ops = _ops()
    g = torch.Generator().manual_seed(11)
    lengths = [0, 1, 63, 64, 65, 1024, 0, 7, 129, 2000]       # SURVEY.md 8c (v) + a >16-chunk ray
    cnt = torch.tensor(lengths)
    R = len(lengths)
    packed = torch.stack([torch.cumsum(cnt, 0) - cnt, cnt], -1)
    N = int(cnt.sum())
    ri = torch.repeat_interleave(torch.arange(R), cnt)
    ts = torch.rand(N, generator=g) * 2
    te = ts + 0.0034641
    sig = torch.rand(N, generator=g) * 30 * (torch.rand(N, generator=g) < 0.3)
    rgb16 = torch.rand(N, 16, generator=g)
    sc, cc = sig.clone().requires_grad_(True), rgb16.clone().requires_grad_(True)
    w_ref = ovr.render_weight_from_density(ts, te, sc, packed)[0]
    rgb_ref = ovr.accumulate_along_rays(w_ref, cc[:, :3], ri, R)
    acc_ref = ovr.accumulate_along_rays(w_ref, None, ri, R)[:, 0]
    dep_ref = ovr.accumulate_along_rays(w_ref, ((ts + te) / 2)[:, None], ri, R)[:, 0]
    sg, cg = sig.clone().cuda().requires_grad_(True), rgb16.clone().cuda().requires_grad_(True)
    rgb, acc, dep, w = ops.volume_render(ts.cuda(), te.cuda(), sg, cg, packed.cuda())
    assert nmax_err(w, w_ref) < TOL_FWD and nmax_err(rgb, rgb_ref) < TOL_FWD
    assert nmax_err(acc, acc_ref) < TOL_FWD and nmax_err(dep, dep_ref) < TOL_FWD
    a, b, c = torch.randn(R, 3, generator=g), torch.randn(R, generator=g), torch.randn(R, generator=g)
    ((rgb_ref * a).sum() + (acc_ref * b).sum() + (dep_ref * c).sum()).backward()
    ((rgb * a.cuda()).sum() + (acc * b.cuda()).sum() + (dep * c.cuda()).sum()).backward()
    assert nmax_err(sg.grad, sc.grad) < TOL_GRAD
    assert nmax_err(cg.grad[:, :3], cc.grad[:, :3]) < TOL_GRAD
    assert float(cg.grad[:, 3:].abs().max()) == 0.0


def test_volrend_properties_full_size():
    """Metric size (4096 rays x 1024 samples): weights in [0,1], per-ray sum <= 1, acc + T_end == 1, linear in rgb."""
    ops = _ops()
    R, S = 4096, 1024
    N = R * S
    g = torch.Generator(device="cuda").manual_seed(1)
    cnt = torch.full((R,), S, dtype=torch.long, device="cuda")
    packed = torch.stack([torch.cumsum(cnt, 0) - cnt, cnt], -1).contiguous()
    ts = (0.05 + 0.0034641 * torch.arange(S, device="cuda", dtype=torch.float32)).repeat(R)
    te = ts + 0.0034641
    sig = torch.rand(N, device="cuda", generator=g) * 3
    c1 = torch.rand(N, 3, device="cuda", generator=g)
    c2 = torch.rand(N, 3, device="cuda", generator=g)
    r1, acc, _, w = ops.volume_render(ts, te, sig, c1, packed)
    r2 = ops.volume_render(ts, te, sig, c2, packed)[0]
    r12 = ops.volume_render(ts, te, sig, 0.25 * c1 + 3 * c2, packed)[0]
    assert float(w.min()) >= 0 and float(w.max()) <= 1
    assert float(acc.max()) <= 1 + 1e-5
    t_end = torch.exp(-(sig * (te - ts)).view(R, S).double().sum(-1))
    assert float((acc.double() + t_end - 1).abs().max()) < 1e-5
    assert float((r12 - (0.25 * r1 + 3 * r2)).abs().max()) < 1e-5


# ------------------------------------------------------------------------------------------------ optimiser / grid update
def test_adam_matches_torch():
    ops = _ops()
    n = 100003
    g = torch.Generator().manual_seed(5)
    p0 = torch.randn(n, generator=g)
    ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=1e-2, eps=1e-15)
    p = p0.clone().cuda()
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    for step in range(1, 6):
        grad = torch.randn(n, generator=g) * (10.0 ** torch.randint(-6, 2, (n,), generator=g).float())
        ref.grad = grad.clone()
        opt.step()
        ops.adam_step(p, grad.cuda(), m, v, 1e-2, 0.9, 0.999, 1e-15, step)
    assert nmax_err(p, ref) < 1e-6
    # grad_scale == averaging over ranks
    p2, m2, v2 = p0.clone().cuda(), torch.zeros(n).cuda(), torch.zeros(n).cuda()
    p3, m3, v3 = p0.clone().cuda(), torch.zeros(n).cuda(), torch.zeros(n).cuda()
    ops.adam_step(p2, (grad * 4).cuda(), m2, v2, 1e-2, 0.9, 0.999, 1e-15, 1, 0.25)
    ops.adam_step(p3, grad.cuda(), m3, v3, 1e-2, 0.9, 0.999, 1e-15, 1, 1.0)
    assert torch.allclose(p2, p3, rtol=1e-6, atol=1e-8)


def test_occ_grid_update_matches_oracle():
    from oracle.sampling import OccGridOracle
    from lsenerf_amd import LSEOccGridEstimator
    aabb = torch.tensor([-1.0, -1, -1, 1, 1, 1])
    og = OccGridOracle(aabb, 16, 2)
    hg = LSEOccGridEstimator(aabb, 16, 2).cuda()
    hg.train()

    def occ_fn_cpu(x):
        return (torch.exp(-4 * (x ** 2).sum(-1, keepdim=True)) * 0.05)

    for step in (0, 16):   # warm-up branch: all cells, same rand draws through a shared CPU generator
        gen = torch.Generator().manual_seed(step)
        lv = og.all_cells()
        for lvl, idx in enumerate(lv):
            u = torch.rand(len(idx), 3, generator=gen)
            x = og.cell_points(lvl, idx, u)
            occ = occ_fn_cpu(x).squeeze(-1)
            og.apply_update(lvl, idx, occ, 0.95)
            cell_ids = (lvl * og.cells_per_lvl + idx).cuda()
            from lsenerf_amd import ops
            ops.occ_update_cells(hg.occs, cell_ids, occ.cuda(), 0.95)
        og.finish_update(0.01)
        thre = torch.clamp(hg.occs[hg.occs >= 0].mean(), max=0.01).reshape(1)
        from lsenerf_amd import ops
        ops.occ_binarize(hg.occs, thre, hg._binaries_u8().view(-1))
        assert nmax_err(hg.occs, og.occs) < 1e-6
        assert torch.equal(hg.binaries.cpu(), og.binaries)
    # duplicates resolve to the maximum (documented deviation: upstream keeps an arbitrary duplicate)
    occs = torch.full((8,), 0.5).cuda()
    ids = torch.tensor([1, 1, 1, 3], dtype=torch.long).cuda()
    new = torch.tensor([0.1, 0.9, 0.2, 0.0]).cuda()
    from lsenerf_amd import ops
    ops.occ_update_cells(occs, ids, new, 0.5)
    assert torch.allclose(occs.cpu(), torch.tensor([0.5, 0.9, 0.5, 0.25, 0.5, 0.5, 0.5, 0.5]))
    # full _update path runs (sampling branch) and keeps invariants
    hg._update(300, lambda x: occ_fn_cpu(x.cpu()).cuda())
    assert hg.binaries.dtype == torch.bool and float(hg.occs.min()) >= 0


# ------------------------------------------------------------------------------------------------ end to end
@pytest.mark.parametrize("emb_type", ["global_emb", "evs_emb"])
def test_model_end_to_end_default_config(emb_type):
    from tests.util import compare_model_outputs, make_model_pair
    hip, orc = make_model_pair(grid_levels=4, grid_resolution=64, occupied_frac=0.25, emb_type=emb_type,
                               param_scale=300.0, alpha_thre=0.0)
    o, d = random_rays(96, seed=21)
    aid = torch.randint(0, 8, (96,), generator=torch.Generator().manual_seed(2)) if emb_type == "evs_emb" else None
    res = compare_model_outputs(hip, orc, o, d, aid, check_grads=True)
    assert res["n_samples"] > 2000


@pytest.mark.parametrize("emb_dim", [16, 48])
def test_model_end_to_end_other_embedding_width(emb_dim):
    """LSEEmbeddingConfig.emb_dim (R:lse_nerf/lse_embeddings.py:94-107) other than the default 32: the head's padded input is 48 / 80
    columns wide; renders and every gradient (hash table, both MLPs, embedding rows, rays) against the oracle."""
    from tests.util import compare_model_outputs, make_model_pair
    hip, orc = make_model_pair(grid_levels=2, grid_resolution=32, occupied_frac=0.4, emb_type="evs_emb", param_scale=300.0,
                               alpha_thre=0.0, emb_dim=emb_dim)
    assert hip.field.mlp_head.in_pad == (31 + emb_dim + 15) // 16 * 16
    o, d = random_rays(96, seed=22)
    aid = torch.randint(0, 8, (96,), generator=torch.Generator().manual_seed(3))
    res = compare_model_outputs(hip, orc, o, d, aid, check_grads=True)
    assert res["n_samples"] > 1000


def test_model_config1_small_field():
    """BASELINE config 1 shapes: L=4 hash grid, 2x32 MLPs (the CPU run of the same config is in test_oracle_cpu.py)."""
    from tests.util import compare_model_outputs, make_model_pair
    hip, orc = make_model_pair(grid_levels=1, grid_resolution=32, occupied_frac=0.6, num_levels=4, hidden=32,
                               param_scale=300.0, alpha_thre=0.0, cone_angle=0.0, contraction=False)
    o, d = random_rays(64, seed=4)
    compare_model_outputs(hip, orc, o, d, None, check_grads=True)


def test_sampler_with_visibility_prepass_matches_oracle():
    """Full LSEOccGridEstimator.sampling incl. the sigma_fn pre-pass (R:lse_nerf/lse_grid_estimator.py:109-143):
    the candidate set is bit-exact; surviving sets may differ only for samples within fp noise of a threshold."""
    from tests.util import make_model_pair
    from lsenerf_amd import RayBundle
    hip, orc = make_model_pair(grid_levels=2, grid_resolution=32, occupied_frac=0.5, param_scale=3000.0, alpha_thre=0.01)
    with torch.no_grad():   # make densities large enough that both culling criteria bite
        hip.field.mlp_base_mlp.params.mul_(3.0)
    from tests.util import sync_params_to_oracle
    sync_params_to_oracle(hip, orc.field)
    R = 128
    o, d = random_rays(R, seed=8)
    jit = torch.rand(R, generator=torch.Generator().manual_seed(1))
    hip.train(); orc.training = True
    rb = RayBundle(origins=o.cuda(), directions=d.cuda(), camera_indices=torch.zeros(R, 1, dtype=torch.long).cuda())
    rs, li = hip.sampler(ray_bundle=rb, near_plane=0.05, far_plane=1e3, render_step_size=hip.config.render_step_size,
                         alpha_thre=0.01, cone_angle=0.004, jitter=jit.cuda())
    with torch.no_grad():
        ri_ref, ts_ref, te_ref = orc.sample(o, d, jitter=jit)
    n_h, n_r = rs.ray_indices.numel(), ri_ref.numel()
    assert n_r > 500 and abs(n_h - n_r) <= max(2, int(2e-3 * n_r)), (n_h, n_r)
    a = set(zip(li.cpu().tolist(), rs.frustums.starts[:, 0].cpu().tolist()))
    b = set(zip(ri_ref.tolist(), ts_ref.tolist()))
    assert len(a ^ b) <= max(2, int(4e-3 * n_r))


def test_errors_are_loud():
    from lsenerf_amd import _lib, ops
    with pytest.raises(_lib.LseHipError):
        ops.hash_encode(torch.rand(4, 3), torch.rand(10), ops.make_grid_meta())          # CPU tensors: no fallback
    meta = ops.MlpMeta(24, 64, 1)
    with pytest.raises(_lib.LseHipError):
        ops.fused_mlp(torch.rand(24 * 64 + 16 * 64).cuda(), torch.rand(8, 24).cuda(), meta, 8)


def test_mlp_fused_density_head_and_compact_output():
    """Base MLP with the trunc_exp density head fused (sigma + its gradient folded into output 0) and the head MLP with
    the compact [N,4] output, against the oracle's separate ops."""
    from oracle.field import TcnnMLP, trunc_exp
    from lsenerf_amd import _lib
    ops = _ops()
    g = torch.Generator().manual_seed(3)
    N = 3001
    om = TcnnMLP(32, 2, 64, 16, None)
    params = om.init_params(g)
    x = torch.randn(N, 32, generator=g)
    sel = torch.rand(N, generator=g) < 0.8
    pc, xc = params.clone().requires_grad_(True), x.clone().requires_grad_(True)
    h_ref = om.forward(xc, pc)
    h_ref_big = h_ref * torch.tensor([12.0] + [1.0] * 15)          # push some logits beyond the +-15 clamp of trunc_exp
    sig_ref = 0.7 * trunc_exp(h_ref_big[:, 0]) * sel
    meta = ops.MlpMeta(32, 64, 1, _lib.LSE_ACT_NONE, _lib.LSE_IN_LEVELMAJOR)
    scale_w = torch.cat([torch.ones(64 * 32), torch.cat([torch.full((64,), 12.0), torch.ones(15 * 64)])])
    pg = (params * scale_w).clone().cuda().requires_grad_(True)      # same effect: row 0 of W_out scaled by 12
    xin = x.view(N, 16, 2).permute(1, 0, 2).contiguous().cuda().requires_grad_(True)
    h, sig = ops.fused_mlp(pg, xin, meta, N, density=(sel.cuda().to(torch.uint8), 0.7))
    assert nmax_err(h, h_ref_big) < TOL_FWD and nmax_err(sig, sig_ref, 1e-3) < 5 * TOL_FWD
    w1, w2 = torch.randn(N, 16, generator=g), torch.randn(N, generator=g) * 1e-3
    ((h_ref_big * w1).sum() + (sig_ref * w2).sum()).backward()
    ((h * w1.cuda()).sum() + (sig * w2.cuda()).sum()).backward()
    assert nmax_err(pg.grad * scale_w.cuda(), pc.grad) < TOL_GRAD      # pg = scale_w * pc  =>  dL/dpc = scale_w * dL/dpg
    assert nmax_err(xin.grad.permute(1, 0, 2).reshape(N, 32), xc.grad) < TOL_GRAD
    # compact head output
    oh = TcnnMLP(16, 3, 64, 3, "Sigmoid")
    ph = oh.init_params(g)
    xh = torch.randn(N, 16, generator=g)
    phc, xhc = ph.clone().requires_grad_(True), xh.clone().requires_grad_(True)
    rgb_ref = oh.forward(xhc, phc)
    mh = ops.MlpMeta(16, 64, 2, _lib.LSE_ACT_SIGMOID, _lib.LSE_IN_ROWMAJOR)
    phg, xhg = ph.clone().cuda().requires_grad_(True), xh.clone().cuda().requires_grad_(True)
    out4 = ops.fused_mlp(phg, xhg, mh, N, out_cols=4)
    assert out4.shape == (N, 4) and nmax_err(out4[:, :3], rgb_ref) < TOL_FWD
    w3 = torch.randn(N, 3, generator=g)
    (rgb_ref * w3).sum().backward()
    (out4[:, :3] * w3.cuda()).sum().backward()
    assert nmax_err(phg.grad[:64 * 16 + 64 * 64], phc.grad[:64 * 16 + 64 * 64]) < TOL_GRAD
    assert nmax_err(phg.grad[64 * 16 + 64 * 64:][:3 * 64], phc.grad[64 * 16 + 64 * 64:][:3 * 64]) < TOL_GRAD
    assert nmax_err(xhg.grad, xhc.grad) < TOL_GRAD


def test_hash_fwd_lds_resident_variant_is_bit_identical():
    """Development knob hash_fwd_lds_levels (the "LDS-staged trilinear interpolation" BASELINE.json's north_star names, kept as the A/B partner
    of the L2-resident level-major schedule, profiles/r04_hash_fwd_lds_ab.txt): the k coarsest levels are gathered from an LDS copy of
    their table.  Same multiply-adds in the same order -> bit-identical features, for ragged counts, a device-side count below the
    capacity, and a grid whose level 1 does not fit the LDS (the variant then stops at level 0)."""
    from lsenerf_amd import _lib, ops
    g = torch.Generator().manual_seed(5)
    with _lib.dev_library() as dev:        # a knob of the development build (csrc/dev_knobs.h); restored when the block ends
        for meta, n in ((ops.make_grid_meta(), 200003), (ops.make_grid_meta(n_levels=4, log2_hashmap_size=12), 777),
                        (ops.make_grid_meta(base_resolution=24, max_res=1024), 65537)):
            table = ((torch.rand(meta.n_params, generator=g) * 2 - 1) * 1e-2).cuda()
            x = torch.rand(n, 3, generator=g).cuda()
            x[: n // 2] = (x[:1] + 1e-3 * torch.arange(n // 2, device="cuda")[:, None]).clamp(0, 1)      # ray-like: neighbours share cells
            dev.set_option("hash_fwd_lds_levels", 0)
            ref = ops.hash_encode(x, table, meta)
            n_dev = torch.tensor([n - 1234 if n > 2000 else n - 7], dtype=torch.int64, device="cuda")
            ref_dev = ops.hash_encode(x, table, meta, n_dev=n_dev)
            for k in (1, 2, 5):
                dev.set_option("hash_fwd_lds_levels", k)
                assert torch.equal(ops.hash_encode(x, table, meta), ref), (meta.n_levels, n, k)
                got = ops.hash_encode(x, table, meta, n_dev=n_dev)
                m = int(n_dev)
                assert torch.equal(got[:, :m], ref_dev[:, :m]) and torch.equal(got[:, :m], ref[:, :m]), (meta.n_levels, n, k)
            dev.set_option("hash_fwd_lds_levels", 0)
    # ... and the library that ships computes the same bits as the development build's default schedule
    assert torch.equal(ops.hash_encode(x, table, meta), ref)


@pytest.mark.parametrize("shape", ["head", "base"])
def test_mlp_full_size_properties(shape):
    """The fused MLPs at BASELINE.json's full size (N = 2^22 + 5 samples: a ragged last tile; the persistent grid then walks 131 073
    32-sample tiles, 64 per wave), through properties that do not need an N-sized oracle:
      * rows of the forward, and rows of the input gradient, on a random subset == a float64 torch evaluation of those rows alone
        (a row depends on its own input row -- and its ray's bias row -- only);
      * additivity over samples: the weight (and per-row bias) gradient of the whole batch == the sum of the gradients of 7 unequal
        chunks launched separately (different tile -> wave assignments, different atomic orders);
      * linearity in the upstream gradient: bwd(g1 - 2 g2) == bwd(g1) - 2 bwd(g2) for the parameter gradient."""
    from lsenerf_amd import _lib
    from oracle.field import TcnnMLP
    ops = _ops()
    N = (1 << 22) + 5
    g = torch.Generator(device="cuda").manual_seed(11)
    head = shape == "head"
    in_dim, nl, act = (16, 3, "Sigmoid") if head else (32, 2, None)
    om = TcnnMLP(in_dim, nl, 64, 16, act)
    om.in_pad = in_dim
    om.shapes[0] = (64, in_dim)
    om.n_params = sum(a * b for a, b in om.shapes)
    params = om.init_params(torch.Generator().manual_seed(3)).cuda()
    meta = ops.MlpMeta(in_dim, 64, nl - 1, _lib.LSE_ACT_SIGMOID if head else _lib.LSE_ACT_NONE,
                       _lib.LSE_IN_ROWMAJOR if head else _lib.LSE_IN_LEVELMAJOR)
    if head:
        x = torch.randn(N, in_dim, device="cuda", generator=g)
        R = 4099                                                     # rays of very different lengths, ray-sorted samples
        cuts = torch.sort(torch.randint(1, N, (R - 1,), device="cuda", generator=g)).values
        bounds = torch.cat([torch.zeros(1, dtype=torch.long, device="cuda"), cuts, torch.tensor([N], device="cuda")])
        cnt = bounds[1:] - bounds[:-1]
        packed = torch.stack([bounds[:-1], cnt], -1).contiguous()
        ridx = torch.repeat_interleave(torch.arange(R, device="cuda", dtype=torch.int32), cnt)
        bias = torch.randn(R, 64, device="cuda", generator=g) * 0.3
    else:
        x = torch.randn(in_dim // 2, N, 2, device="cuda", generator=g)          # level-major
        bias = ridx = packed = None

    def run(lo, hi, gout, want_in=False):
        """forward + backward of samples [lo, hi) (chunk boundaries on ray boundaries for the head): (out, d_params, d_bias, d_in)"""
        p = params.clone().requires_grad_(True)
        if head:
            r0, r1 = int(ridx[lo]), int(ridx[hi - 1]) + 1
            xs = x[lo:hi].clone().requires_grad_(want_in)
            b = bias[r0:r1].clone().requires_grad_(True)
            pk = packed[r0:r1].clone()
            pk[:, 0] -= lo
            out = ops.fused_mlp(p, xs, meta, hi - lo, b, (ridx[lo:hi] - r0).contiguous(), pk)
        else:
            xs = x[:, lo:hi].contiguous().requires_grad_(want_in)
            b = None
            out = ops.fused_mlp(p, xs, meta, hi - lo)
        out.backward(gout[lo:hi])
        db = None
        if head:
            db = torch.zeros_like(bias)
            db[r0:r1] = b.grad
        return out.detach(), p.grad, db, (xs.grad if want_in else None)

    g1 = torch.randn(N, 16, device="cuda", generator=g)
    out, dp, db, din = run(0, N, g1, want_in=True)
    # ---- rows against float64
    sub = torch.randint(0, N, (4096,), device="cuda", generator=g)
    sub[:3] = torch.tensor([0, N - 1, N - 5], device="cuda")                         # first row, last row, first row of the ragged tile
    Ws = [W.double() for W in om.matrices(params.cpu())]
    xr = (x[sub] if head else x[:, sub].permute(1, 0, 2).reshape(-1, in_dim)).double().cpu().requires_grad_(True)
    h = xr @ Ws[0].t()
    if head:
        h = h + bias[ridx[sub].long()].double().cpu()
    h = torch.relu(h)
    for W in Ws[1:-1]:
        h = torch.relu(h @ W.t())
    ref = h @ Ws[-1].t()
    if head:
        ref = torch.sigmoid(ref)
    assert nmax_err(out[sub], ref) < TOL_FWD
    ref.backward(g1[sub].double().cpu())
    din_rows = din[sub] if head else din[:, sub].permute(1, 0, 2).reshape(-1, in_dim)
    assert nmax_err(din_rows, xr.grad) < TOL_GRAD
    # ---- additivity over samples
    if head:
        ray_cuts = [0, 7, 500, 1203, 2048, 3000, 4000, R]
        cuts_s = [int(bounds[r]) for r in ray_cuts]
    else:
        cuts_s = [0, 33, 100001, 1 << 20, (1 << 21) + 17, 3000000, 4000000, N]
    dp_sum, db_sum = torch.zeros_like(dp), (torch.zeros_like(db) if head else None)
    for lo, hi in zip(cuts_s[:-1], cuts_s[1:]):
        if hi > lo:
            _, dpc, dbc, _ = run(lo, hi, g1)
            dp_sum += dpc
            if head:
                db_sum += dbc
    assert nmax_err(dp_sum, dp, 1e-12) < TOL_GRAD and blockwise_nmax_err(dp_sum, dp, mlp_row_bounds(om)) < TOL_GRAD_BLOCK
    if head:
        assert nmax_err(db_sum, db, 1e-12) < TOL_GRAD
    # ---- linearity in the upstream gradient
    g2 = torch.randn(N, 16, device="cuda", generator=g)
    _, dp2, _, _ = run(0, N, g2)
    _, dp12, _, _ = run(0, N, g1 - 2.0 * g2)
    assert nmax_err(dp12, dp - 2.0 * dp2, 1e-12) < TOL_GRAD


def test_compact_features_equals_boolean_indexing_for_every_level_grouping():
    """lse_compact_features (survivors of the visibility pre-pass keep their positions, selector and level-major hash features) against
    torch boolean indexing, with the levels split over 1 .. 16 groups of waves per ray (option compact_features_groups; default 4),
    level counts that do not divide evenly, rays without candidates and rays without survivors."""
    from lsenerf_amd import _lib
    ops = _ops()
    g = torch.Generator(device="cuda").manual_seed(2)
    with _lib.dev_library() as dev:
        for L, R in ((16, 257), (5, 33), (1, 7)):
            cnt = torch.randint(0, 200, (R,), device="cuda", generator=g)
            cnt[::7] = 0                                                      # rays that missed every occupied cell
            packed = torch.stack([torch.cumsum(cnt, 0) - cnt, cnt], -1).contiguous()
            N = int(cnt.sum())
            keep = torch.rand(N, device="cuda", generator=g) < 0.6
            seg = torch.repeat_interleave(torch.arange(R, device="cuda"), cnt)
            keep[seg % 5 == 1] = False                                        # rays whose candidates are all culled
            new_cnt = torch.zeros(R, dtype=torch.long, device="cuda").index_add_(0, seg, keep.long())
            new_packed = torch.stack([torch.cumsum(new_cnt, 0) - new_cnt, new_cnt], -1).contiguous()
            n_new = int(new_cnt.sum())
            x01 = torch.rand(N, 3, device="cuda", generator=g)
            sel = (torch.rand(N, device="cuda", generator=g) < 0.9).to(torch.uint8)
            y = torch.randn(L, N, 2, device="cuda", generator=g)
            for groups in (1, 2, 3, 4, 16):
                dev.set_option("compact_features_groups", groups)
                ox, os_, oy = ops.compact_features(keep.to(torch.uint8).contiguous(), packed, new_packed, n_new, x01, sel, y)
                assert torch.equal(ox, x01[keep]) and torch.equal(os_, sel[keep]) and torch.equal(oy, y[:, keep]), (L, R, groups)
                if groups == 4:      # the shipped library's constant: same result from liblse_hip.so
                    with _ShippedLibrary():
                        ox, os_, oy = ops.compact_features(keep.to(torch.uint8).contiguous(), packed, new_packed, n_new, x01, sel, y)
                        assert torch.equal(ox, x01[keep]) and torch.equal(os_, sel[keep]) and torch.equal(oy, y[:, keep]), (L, R)


class _ShippedLibrary:
    """Inside a dev_library() block: back to liblse_hip.so for a few calls."""

    def __enter__(self):
        from lsenerf_amd import _lib
        self._inner, _lib._lib = _lib._lib, _lib._prod_lib
        return self

    def __exit__(self, *exc):
        from lsenerf_amd import _lib
        _lib._lib = self._inner
        return False
