#!/usr/bin/env python3
"""CU-count sensitivity of the hash backward (and of the MLP backward next to it) on CU-masked streams.

The hash backward waits on the memory-side atomic units (17 of 21 G requests/s in every regime), so it may
need far fewer than 256 CUs.  If it does, half a batch's MLP backward could run on the complementary CUs while the other half
scatters.  Measured here at the metric size (bench.py's headline positions):
  1. lse_hash_bwd, whole batch, on streams restricted to 256 / 224 / 192 / 160 / 128 / 96 / 64 CUs;
  2. the base-MLP backward, whole batch, on the same masks (expected: ~ 1 / CUs);
  3. pairs: hash backward of half A on X CUs || head + base MLP backward of half B on the other 256 - X CUs, against the same
     two calls one after the other on the whole chip (what the step does today).
hipExtStreamCreateWithCUMask: bit i of the mask enables CU i; on a multi-XCD part the bits go round-robin over the XCDs, so a
prefix of k bits is k / 8 CUs on every XCD.   usage: python tools/cu_mask_probe.py [out.txt]"""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from lsenerf_amd import ops, _lib

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
hip = ctypes.CDLL("libamdhip64.so")
hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
hip.hipExtStreamCreateWithCUMask.restype = ctypes.c_int
N_CU = torch.cuda.get_device_properties(0).multi_processor_count
out_lines = []


def say(*a):
    s = " ".join(str(x) for x in a)
    print(s, flush=True)
    out_lines.append(s)


def masked_stream(bits):
    """bits: iterable of enabled CU indices."""
    words = [0] * ((N_CU + 31) // 32)
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    arr = (ctypes.c_uint32 * len(words))(*words)
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), len(words), arr)
    assert rc == 0, f"hipExtStreamCreateWithCUMask rc={rc}"
    return torch.cuda.ExternalStream(st.value, device=dev)


def timed(fn, stream=None, iters=7):
    stream = stream or torch.cuda.current_stream()
    ts = []
    for i in range(iters + 2):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(stream):
            e0.record()
            fn()
            e1.record()
        torch.cuda.synchronize()
        if i >= 2:
            ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]


# ---- the headline workload's tensors ---------------------------------------------------------------------------------------
model, _sets, _ = bench.build_workload(dev, 1000)      # (one ray set: the round-1..4 fixed draw)
rb, target, jitter = _sets[0]
cfg, fld = model.config, model.field
with torch.no_grad():
    ri, ts_, te_, packed = model.occupancy_grid.sampling(
        rb.origins.detach(), rb.directions.detach(), near_plane=cfg.near_plane, far_plane=cfg.far_plane,
        render_step_size=cfg.render_step_size, stratified=True, jitter=jitter, return_packed=True)[:4]
    x01, sel = ops.positions(rb.origins.detach(), rb.directions.detach(), ri, ts_, te_, packed, True, None)
n = x01.shape[0]
R = rb.origins.shape[0]
meta = fld.mlp_base_grid.meta
desc = meta.desc()
table = fld.mlp_base_grid.params.detach()
P = lambda t: ctypes.c_void_p(t.data_ptr())
say(f"device CUs {N_CU}; samples {n}; rays {R}")


def make_hash_bwd(lo, hi):
    """hash backward of samples [lo, hi): (callable, keep-alive)."""
    m = hi - lo
    xs = x01[lo:hi].contiguous()
    dy = torch.randn(16, m, 2, device=dev)
    dt = torch.zeros_like(table)
    dx = torch.empty_like(xs)
    o = _lib.hash_bwd_default_opts()
    nb = int(_lib.load().lse_hash_bwd_workspace_bytes(ctypes.byref(desc), ctypes.byref(o)))
    ws = torch.zeros(max(nb // 4, 1), dtype=torch.float32, device=dev)
    if nb:
        o.workspace, o.workspace_bytes = ws.data_ptr(), nb
    keep = (xs, dy, dt, dx, ws, o)

    def run():
        _lib.call("lse_hash_bwd_ex", ctypes.byref(desc), P(xs), P(dy), P(table), P(dt), P(dx), 0, 0, 16, m, None, ctypes.byref(o), ops._stream())
    return run, keep


def make_mlp_bwd(ray_lo, ray_hi, which=("head", "base"), stream=None):
    """MLP backward (head and / or base) of the samples of rays [ray_lo, ray_hi): the forward runs once here, the returned callable
    runs only the backward kernels (torch.autograd.grad with preallocated output gradients, graph retained).  The autograd engine
    runs a node's backward on the stream its FORWARD ran on, so the forward runs under ``stream`` (the masked stream to be timed)."""
    with torch.cuda.stream(stream or torch.cuda.current_stream()):
        res = _make_mlp_bwd(ray_lo, ray_hi, which)
    torch.cuda.synchronize()
    return res


def _make_mlp_bwd(ray_lo, ray_hi, which):
    s_lo, s_hi = int(packed[ray_lo, 0]), int(packed[ray_hi - 1, 0] + packed[ray_hi - 1, 1])
    m = s_hi - s_lo
    pk = packed[ray_lo:ray_hi].clone()
    pk[:, 0] -= s_lo
    ridx = (ri[s_lo:s_hi] - ray_lo).contiguous()
    with torch.no_grad():
        y = fld.mlp_base_grid.forward_levelmajor(x01[s_lo:s_hi].contiguous())
    y = y.detach().requires_grad_(True)
    h, sigma = fld._base_mlp(y, sel[s_lo:s_hi].contiguous(), m)
    outs, ins, gos = [], [], []
    if "base" in which:
        outs += [h, sigma]; ins += [y, fld.mlp_base_mlp.params]; gos += [torch.randn_like(h), torch.randn_like(sigma)]
    if "head" in which:
        h2 = h.detach().requires_grad_(True)
        table_e = fld._train_emb_table()
        eidx = torch.zeros(ray_hi - ray_lo, dtype=torch.int32, device=dev)
        rgb = fld.rgb_packed(h2, rb.directions.detach()[ray_lo:ray_hi].contiguous(), eidx, ridx, pk, table_e)
        outs_h, ins_h, gos_h = [rgb], [h2, fld.mlp_head.params], [torch.randn_like(rgb)]
    keep = (y, h, sigma)

    def run():
        if "head" in which:
            torch.autograd.grad(outs_h, ins_h, gos_h, retain_graph=True)
        if "base" in which:
            torch.autograd.grad(outs, ins, gos, retain_graph=True)
    return run, keep


full = masked_stream(range(N_CU))
hb_full, _k1 = make_hash_bwd(0, n)
mlp_base_full, _k2 = make_mlp_bwd(0, R, which=("base",))
mlp_both_full, _k3 = make_mlp_bwd(0, R)
say(f"plain stream: hash_bwd {timed(hb_full):.3f} ms   base-MLP bwd {timed(mlp_base_full):.3f} ms   head+base MLP bwd {timed(mlp_both_full):.3f} ms")
say("\n1/2. whole batch on CU-masked streams (prefix masks = the same share of every XCD)")
say(f"{'CUs':>5} {'hash_bwd ms':>12} {'vs 256':>7} {'base-MLP bwd ms':>16} {'vs 256':>7} {'head+base ms':>13} {'vs 256':>7}")
ref = None
for k in (256, 224, 192, 160, 128, 96, 64):
    st = masked_stream(range(k))
    mb_, _ka = make_mlp_bwd(0, R, which=("base",), stream=st)
    mh_, _kb = make_mlp_bwd(0, R, stream=st)
    a, b, c = timed(hb_full, st), timed(mb_, st), timed(mh_, st)
    del mb_, mh_, _ka, _kb
    ref = ref or (a, b, c)
    say(f"{k:>5} {a:>12.3f} {a / ref[0]:>7.2f} {b:>16.3f} {b / ref[1]:>7.2f} {c:>13.3f} {c / ref[2]:>7.2f}")
say("\n   other mask layouts at 128 CUs (hash_bwd): prefix = 16 CUs of every XCD; 'xcd0-3' = all CUs of four XCDs; 'alt' = every other bit pair")
for name, bits in (("prefix", range(128)), ("xcd0-3", [b for b in range(256) if b % 8 < 4]), ("alt", [b for b in range(256) if (b // 8) % 2 == 0])):
    st = masked_stream(bits)
    mh_, _kb = make_mlp_bwd(0, R, stream=st)
    say(f"   {name:>8}: hash_bwd {timed(hb_full, st):.3f} ms   head+base MLP bwd {timed(mh_, st):.3f} ms")
    del mh_, _kb

# ---- 3. the overlap the experiment is about ---------------------------------------------------------------------------------
say("\n3. half A scatters on X CUs while half B's MLP backward (head + base) runs on the other 256 - X")
half = R // 2
n_half = int(packed[half, 0])
hb_A, _k4 = make_hash_bwd(0, n_half)
hb_B, _k5 = make_hash_bwd(n_half, n)
mlp_A, _k6 = make_mlp_bwd(0, half)
mlp_B, _k7 = make_mlp_bwd(half, R)
t_hA, t_hB, t_mA, t_mB = timed(hb_A), timed(hb_B), timed(mlp_A), timed(mlp_B)
say(f"   whole chip, one after the other: hash_bwd A {t_hA:.3f}  B {t_hB:.3f}   MLP bwd A {t_mA:.3f}  B {t_mB:.3f}   "
    f"(today's order MLP(A+B) -> hash(A+B): {timed(mlp_both_full) + timed(hb_full):.3f} ms)")


def seq_today():
    mlp_both_full(); hb_full()


t_today = timed(seq_today)
say(f"   today, measured as one sequence on one stream: {t_today:.3f} ms")
for X in (64, 96, 128, 160, 192):
    sA, sB = masked_stream(range(X)), masked_stream(range(X, N_CU))
    main = torch.cuda.current_stream()
    mlp_B, _k7 = make_mlp_bwd(half, R, stream=sB)        # (its backward runs on the stream of its forward: sB)

    def pipelined():
        # MLP_A (whole chip) -> [hash_A on X || MLP_B on the rest] -> hash_B (whole chip)
        mlp_A()
        sA.wait_stream(main); sB.wait_stream(main)
        with torch.cuda.stream(sA):
            hb_A()
        with torch.cuda.stream(sB):
            mlp_B()
        main.wait_stream(sA); main.wait_stream(sB)
        hb_B()

    def pair_only():
        sA.wait_stream(main); sB.wait_stream(main)
        with torch.cuda.stream(sA):
            hb_A()
        with torch.cuda.stream(sB):
            mlp_B()
        main.wait_stream(sA); main.wait_stream(sB)
    t_pair, t_pipe = timed(pair_only), timed(pipelined)
    t_hx, t_mx = timed(hb_A, sA), timed(mlp_B, sB)
    say(f"   X = {X:>3}: hash_A alone on X {t_hx:.3f}  MLP_B alone on 256-X {t_mx:.3f}  both together {t_pair:.3f}  "
        f"(sum on the whole chip {t_hA + t_mB:.3f});  MLP_A -> pair -> hash_B = {t_pipe:.3f} ms vs today {t_today:.3f}")
# the same pair WITHOUT masks (two plain streams): what round 1 measured as a loss
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
main = torch.cuda.current_stream()
mlp_B, _k7 = make_mlp_bwd(half, R, stream=s2)


def pair_plain():
    s1.wait_stream(main); s2.wait_stream(main)
    with torch.cuda.stream(s1):
        hb_A()
    with torch.cuda.stream(s2):
        mlp_B()
    main.wait_stream(s1); main.wait_stream(s2)


say(f"   two plain streams, no masks: both together {timed(pair_plain):.3f} ms (sum one after the other {t_hA + t_mB:.3f})")
if len(sys.argv) > 1:
    with open(sys.argv[1], "w") as f:
        f.write("\n".join(out_lines) + "\n")
