#!/bin/bash
# One GPU-box round: parity tests, smoke, bench, rocprofv3 kernel stats.  Usage: tools/gpu_round.sh <tag> [pytest-args]
TAG=${1:-r}; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
if [ -z "$SKIP_PYTEST" ]; then
  # unbuffered + verbose + faulthandler: a test that hangs is named in the log (a buffered -q log of a killed run is empty)
  PYTHONUNBUFFERED=1 timeout -k 10 700 python -u -m pytest tests -m gpu -v --timeout 240 -o faulthandler_timeout=200 "$@" > $OUT/pytest.log 2>&1; rc=$?
  echo "pytest rc=$rc"; tail -4 $OUT/pytest.log
  if [ $rc -ge 124 ]; then echo "pytest killed; stopping"; exit $rc; fi
fi
timeout -k 10 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; rc=$?; echo "smoke rc=$rc"; tail -2 $OUT/smoke.log
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 600 python bench.py > $OUT/bench.log 2>&1; rc=$?; echo "bench rc=$rc"; tail -3 $OUT/bench.log | cut -c1-1800
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-context --no-atomic-floor > $OUT/rocprof.log 2>&1; rc=$?
echo "rocprof rc=$rc"; find $OUT/prof -name "*kernel_stats*" | head
# HBM traffic counters: separate passes (FETCH_SIZE and WRITE_SIZE do not fit one pass), kernel-trace only; TCC_EA0_ATOMIC_sum = the
# memory-side float-atomic REQUESTS the hash backward actually sends (priced against the request rate in bench.py's roofline.atomic)
for C in FETCH_SIZE WRITE_SIZE TCC_EA0_ATOMIC_sum; do
  timeout -k 10 400 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$C -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-context --no-atomic-floor > $OUT/pmc_$C.log 2>&1; rc=$?
  echo "pmc $C rc=$rc"; if [ $rc -ge 124 ]; then exit $rc; fi
done
find $OUT -name "*counter_collection*" | head
