#!/bin/bash
# On the GPU box: bench every variant library (and the default) back to back, twice, interleaved.
# usage: ab_bench.sh [extra bench.py flags, e.g. --pose 1]
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
  for lib in blok_amd/libblok_hip.so blok_amd/variants/*.so; do
    BLOK_HIP_LIB=$PWD/$lib timeout -k 10 120 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-paths "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print('$lib', 'rep$rep', round(d['value']), 'Mrays/s', round(d['ms_per_step'], 4), 'ms', 'alone', round(d['config']['kernel_ms_alone'], 4))"
  done
done
