#!/bin/bash
# On the GPU box: one-launch frame vs two-launch form, frames in flight 1 and 3, twice, interleaved.
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
  for fused in 1 0; do
    for fif in 1 3; do
      timeout -k 10 120 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-paths --fused $fused --frames-in-flight $fif "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print('fused=$fused in_flight=$fif rep$rep', round(d['value']), 'Mrays/s', round(d['ms_per_step'], 4), 'ms', 'alone', round(d['config']['kernel_ms_alone'], 4))"
    done
  done
done
