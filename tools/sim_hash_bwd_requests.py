"""CPU model of hash_bwd atomic line requests (64-B lines) per sample for different lane<->sample mappings."""
import numpy as np, sys
sys.path.insert(0, '.')
import torch
from oracle import hashgrid as hg
from oracle.field import contract_inf
meta = hg.tcnn_grid_meta()
R, S = 48, 1024
g = torch.Generator().manual_seed(1)
o = torch.rand(R, 3, generator=g) - 0.5
d = torch.randn(R, 3, generator=g); d = d / d.norm(dim=-1, keepdim=True)
step = 2 * 3 ** 0.5 / 1000
t = 0.05 + step * (torch.arange(S) + 0.5)
pos = o[:, None, :] + d[:, None, :] * t[None, :, None]
x01 = ((contract_inf(pos.reshape(-1, 3)) + 2) / 4).float()
N = x01.shape[0]
tot = {"blocked": 0, "interleaved": 0, "hybrid": 0, "lanes": 0}
per_level = []
for l in range(16):
    idx = hg.tcnn_corner_indices(x01, meta, l).numpy()          # [N,8] absolute entry index
    line = idx >> 3
    res = {}
    for name in ("blocked", "interleaved"):
        req = 0; lanes = 0
        for w in range(N // 64):
            base = w * 64
            # sample handled by group gi at round r
            if name == "blocked":
                samp = np.array([[16 * gi + r for gi in range(4)] for r in range(16)]) + base
            else:
                samp = np.array([[4 * r + gi for gi in range(4)] for r in range(16)]) + base
            cur = np.full((4, 8), -1, dtype=np.int64)
            for r in range(16):
                new = idx[samp[r]]                                 # [4,8]
                flush = (cur != new) & (cur >= 0)
                if flush.any():
                    req += len(np.unique(cur[flush] >> 3)); lanes += int(flush.sum())
                cur = new
            req += len(np.unique(cur >> 3)); lanes += 32
        res[name] = req / N
        if name == "blocked": tot["lanes"] += lanes / N
    per_level.append(res)
    tot["blocked"] += res["blocked"]; tot["interleaved"] += res["interleaved"]; tot["hybrid"] += min(res.values())
    print(f"level {l:2d} res {meta.resolutions[l]:5d}: requests/sample blocked {res['blocked']:.3f}  interleaved {res['interleaved']:.3f}")
print("total requests/sample:", {k: round(v, 2) for k, v in tot.items()})
print("predicted atomic time at 20.7 G req/s for N=4.19M: blocked %.2f ms, hybrid %.2f ms" % (tot["blocked"] * 4194304 / 20.7e9 * 1e3, tot["hybrid"] * 4194304 / 20.7e9 * 1e3))

# lower bounds: every distinct line (or entry) requested once per chunk of C consecutive samples
for C in (64, 256, 1024):
    tl = te = 0.0
    for l in range(16):
        idx = hg.tcnn_corner_indices(x01, meta, l).numpy()
        for w in range(N // C):
            blk = idx[w * C:(w + 1) * C].reshape(-1)
            tl += len(np.unique(blk >> 3)); te += len(np.unique(blk))
    print(f"chunk {C:5d} samples: distinct lines/sample {tl / N:.2f}, distinct entries/sample {te / N:.2f}")

# ---- cache policy of hash_bwd_cached_kernel: 64-sample chunks, run ends only, one probe.  Cost unit = 32-B sectors
# reaching the memory-side atomic units (rocprof WRITE_SIZE / 32 B matches this count: 16.5 per sample for the
# 256 x 64-B variant, measured 2.17 GKiB per launch).
def cache_policy(ent_log2, n_slots, name, slotfn=None):
    tl = tf = 0.0
    for l in range(16):
        scale = np.float32(meta.scales[l])
        p = np.floor((x01.numpy() * scale + np.float32(0.5)).astype(np.float32)).astype(np.int64)
        idx = hg.tcnn_corner_indices(x01, meta, l).numpy()
        sect = fb = 0
        for w in range(N // 64):
            pp, ii = p[w * 64:(w + 1) * 64], idx[w * 64:(w + 1) * 64]
            same = np.concatenate([[False], (pp[1:] == pp[:-1]).all(1)])
            ends = np.concatenate([~same[1:], [True]])
            pe, ie = pp[ends], ii[ends]
            keys, flushed = {}, set()
            for c in range(8):
                gran = ie[:, c] >> ent_log2
                sl = slotfn(pe[:, 0] + (c & 1), pe[:, 1] + ((c >> 1) & 1), pe[:, 2] + (c >> 2), gran) if slotfn else \
                    ((gran * 0x9E3779) >> (24 - int(np.log2(n_slots)))) & (n_slots - 1)
                for s_, gk, e in zip(sl, gran, ie[:, c]):
                    k = keys.get(s_)
                    if k is None: keys[s_] = gk; flushed.add(e >> 2)
                    elif k == gk: flushed.add(e >> 2)
                    else: fb += 1
            sect += len(flushed)
        tl += sect / N; tf += fb / N
    print(f"cache {name}: flushed sectors/sample {tl:.2f} + direct-to-memory updates/sample {tf:.2f} = {tl + tf:.2f}")
cache_policy(3, 256, "256 slots x 64-B line, multiplicative hash")
cache_policy(3, 256, "256 slots x 64-B line, slot linear in cell coordinates", lambda x, y, z, g: ((x >> 3) * 7 + y * 19 + z * 83) & 255)
cache_policy(2, 512, "512 slots x 32-B sector, multiplicative hash (default)")
