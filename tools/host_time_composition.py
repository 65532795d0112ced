#!/usr/bin/env python3
"""Host-side accounting of the reference's step composition (bench.context_composition): Python time vs time blocked in the
sampler's count read-backs vs GPU kernel time.  usage: python tools/host_time_composition.py [cfg2|cfg4]"""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from lsenerf_amd import _lib
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
_lib.load()
r = bench.context_composition(dev, sys.argv[1] if len(sys.argv) > 1 else "cfg2")
r.pop("kernel_ms_per_step")
print(json.dumps(r, indent=1))
