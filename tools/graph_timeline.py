#!/usr/bin/env python3
"""Timeline of ONE replay of the captured 3-bundle training step (bench.py cfg2 composition) from a rocprofv3 kernel trace:
every kernel in launch order with its duration and the idle gap in front of it -- where a replayed step's time goes beyond the sum
of its kernels.
   run:      rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 tools/graph_timeline.py run
   analyse:  python tools/graph_timeline.py analyse <dir> [out.txt]"""
import csv, glob, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

if sys.argv[1] == "run_headline":      # the eager M-march step bench.py's `value` is measured on
    import torch
    import bench
    from lsenerf_amd.optim import FlatAdam, FlatParams
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    model, _sets, _ = bench.build_workload(dev, 1000)      # (one ray set: the round-1..4 fixed draw)
    rb, target, jitter = _sets[0]
    opt = FlatAdam(FlatParams(model.get_param_groups()["fields"]), lr=1e-2, eps=1e-15, lr_final=1e-4, max_steps=200000)
    for _ in range(12):
        bench.train_step(model, rb, target, jitter, opt, 1)
    torch.cuda.synchronize()
    sys.exit(0)

if sys.argv[1] == "run":
    import torch
    import bench
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    res = bench.context_composition(dev, "cfg2", steps=6, warmup=3)
    print("graphed ms/step", res["graphed"]["ms_per_step"], "prefetched", res["graphed"]["marcher_prefetched"]["ms_per_step"])
    sys.exit(0)

d = sys.argv[2]
rows = []
for f in glob.glob(f"{d}/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", "?")))
rows.sort()
# the graphed replays are the LAST dense runs of the trace: find the last occurrence of the Adam kernel that closes a step, then walk
# back to the previous one -> one full step in between
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r[2]]
out = []
def say(s):
    print(s); out.append(s)
# pick a replay in the middle of the second-to-last timing block (plain graphed) and of the last block (prefetched)
def step_between(i0, i1, title):
    seg = rows[i0 + 1:i1 + 1]
    t0 = rows[i0][1]
    say(f"== {title}: {len(seg)} kernels, {(seg[-1][1] - t0) / 1e3:.1f} us from the end of the previous step's Adam to the end of this one's")
    busy = 0
    prev_end = t0
    gaps = 0
    for s, e, name, st in seg:
        nm = name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:58]
        gap = s - prev_end
        say(f"   {nm:58s} stream {st:>3}  {(e - s) / 1e3:8.1f} us   gap before {gap / 1e3:7.1f} us")
        busy += e - s
        gaps += max(gap, 0)
        prev_end = max(prev_end, e)
    say(f"   sum of kernel durations {busy / 1e3:.1f} us, sum of positive gaps {gaps / 1e3:.1f} us")
n = len(adam)
# blocks: ... the last 9 Adam launches belong to the prefetched timing (3 warm-up + 6 timed), the 9 before to the plain graphed timing
if sys.argv[1] == "analyse_headline":
    step_between(adam[-3], adam[-2], "eager M-march step (bench.py headline)")
elif n >= 20:
    step_between(adam[-14], adam[-13], "plain graphed replay")
    step_between(adam[-3], adam[-2], "replay with the next step's marcher on a side stream")
if len(sys.argv) > 3:
    open(sys.argv[3], "w").write("\n".join(out) + "\n")
