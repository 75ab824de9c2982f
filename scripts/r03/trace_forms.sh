#!/bin/bash
# Kernel trace of the frame's launch forms, one frame in flight: per-kernel durations.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03/forms; rm -rf $OUT; mkdir -p $OUT
for form in ${FORMS:-0 5 4 2}; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/f$form -o t -- python3 bench.py --steps 40 --warmup 5 --fused $form --frames-in-flight 1 --no-cpu-baseline --no-paths --no-poses ${EXTRA} > $OUT/f$form.log 2>&1
  echo "== form $form"; python3 - <<PY
import csv
rows = list(csv.DictReader(open("$OUT/f$form/t_kernel_stats.csv")))
for r in rows[:6]:
    print(f"  {r['Name'][:90]:90s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us  min {float(r['MinNs'])/1e3:8.1f}  max {float(r['MaxNs'])/1e3:8.1f}")
PY
done
