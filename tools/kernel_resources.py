#!/usr/bin/env python3
"""Print per-kernel register/LDS/spill usage of a .hip file (hipcc -Rpass-analysis=kernel-resource-usage)."""
import re, subprocess, sys
src = sys.argv[1]
extra = sys.argv[2:]
out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950",
                      "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"] + extra,
                     capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark: (?:\s*)([A-Za-z ]+?)(?: \[bytes/\w+\]| \[waves/SIMD\])?: (\S+) \[-Rpass", line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2)
    if k == "Function Name":
        cur = subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()
        cur = re.sub(r"\(anonymous namespace\)::", "", cur)
        cur = re.sub(r"\(.*", "", cur).replace("void ", "")
        rows[cur] = {}
    elif cur:
        rows[cur][k] = v
print(f"{'kernel':60s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'spill':>5s} {'scratch':>7s} {'occ':>3s} {'LDS':>6s}")
for k, r in rows.items():
    print(f"{k[:60]:60s} {r.get('VGPRs','?'):>5s} {r.get('AGPRs','?'):>5s} {r.get('SGPRs','?'):>5s} "
          f"{r.get('VGPRs Spill','?'):>5s} {r.get('ScratchSize','?'):>7s} {r.get('Occupancy','?'):>3s} {r.get('LDS Size','?'):>6s}")
