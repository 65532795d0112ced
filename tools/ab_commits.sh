#!/bin/bash
# A/B the bench of several checked-out worktrees on ONE box: tools/ab_commits.sh <dir> <dir> ...   (first run = box warm-up)
run() {
  local c=$1; local flags="--no-cpu-baseline"
  grep -q "no-context" $c/bench.py && flags="$flags --no-context"
  (cd $c && timeout -k 10 300 python bench.py $flags 2>/dev/null | tail -1) > /tmp/ab_x.json
  python - "$c" <<'PY'
import json, sys
d = json.loads(open("/tmp/ab_x.json").read())
print(sys.argv[1], "ms_per_step", round(d["ms_per_step"], 3), "kernels", round(sum(d["kernel_ms_per_step"].values()), 3))
PY
}
run $1 > /dev/null
for c in "$@"; do run $c; done
