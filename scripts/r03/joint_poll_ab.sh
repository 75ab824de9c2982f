#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03
for lib in blok_amd/libblok_hip.so blok_amd/variants/*.so; do
  for form in 4 5 2; do
    BLOK_HIP_LIB=$PWD/$lib timeout -k 10 120 python3 bench.py --steps 60 --warmup 5 --fused $form --frames-in-flight 1 --no-cpu-baseline --no-paths --no-poses 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print('$lib form $form: alone', round(d['config']['kernel_ms_alone'], 4), 'ms, gave up', d['config']['walk_waves_that_gave_up_waiting'])"
  done
done 2>&1 | tee gpurun_out/r03/joint_poll_ab.txt
