#!/bin/bash
# On the GPU box: default bench with tile ordering on (re-sort every 8 frames) / off, static and orbiting camera, interleaved twice.
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
  for orbit in 0 1.0; do
    for ord in 8 0; do
      timeout -k 10 120 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-paths --no-poses --tile-ordering $ord --orbit $orbit "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print('ordering=$ord orbit=$orbit rep$rep', round(d['value']), 'Mrays/s', round(d['ms_per_step'], 4), 'ms', 'alone', round(d['config']['kernel_ms_alone'], 4))"
    done
  done
done
