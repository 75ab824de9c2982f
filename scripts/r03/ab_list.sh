#!/bin/bash
# On the GPU box: the bench line of this build and of the round-2 library, static and orbiting camera, interleaved.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03
OUT=gpurun_out/r03/ab_list.txt; : > $OUT
show() { python3 -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); c = d['config']
print('$1', round(d['value']), 'Mrays/s', round(d['ms_per_step'], 4), 'ms/frame; alone', round(c['kernel_ms_alone'], 4), 'ms;', {k: (round(v['Mrays_per_s']), round(v['ms_per_frame_alone'], 4)) for k, v in c['poses'].items()}, 'gave up', c['walk_waves_that_gave_up_waiting'])" | tee -a $OUT; }
for rep in 1 2; do
  timeout -k 10 200 python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-paths 2>/dev/null | show "new static rep$rep"
  BLOK_HIP_LIB=$PWD/blok_amd/variants/libblok_hip_r2.so timeout -k 10 200 python3 bench.py --steps 100 --warmup 10 --settle 32 --no-cpu-baseline --no-paths --no-poses 2>/dev/null | show "r2  static rep$rep (settle 32; its alone figure has not settled)"
  timeout -k 10 200 python3 bench.py --steps 100 --warmup 10 --orbit 1 --no-cpu-baseline --no-paths --no-poses 2>/dev/null | show "new orbit1 rep$rep"
  BLOK_HIP_LIB=$PWD/blok_amd/variants/libblok_hip_r2.so timeout -k 10 200 python3 bench.py --steps 100 --warmup 10 --orbit 1 --no-cpu-baseline --no-paths --no-poses 2>/dev/null | show "r2  orbit1 rep$rep"
done
for form in 4 5 2 0; do
  timeout -k 10 200 python3 bench.py --steps 100 --warmup 10 --fused $form --frames-in-flight 1 --no-cpu-baseline --no-paths --no-poses 2>/dev/null | show "new form $form, 1 in flight"
done
timeout -k 10 200 python3 bench.py --steps 100 --warmup 10 --fused 4 --no-cpu-baseline --no-paths --no-poses 2>/dev/null | show "new form 4, 3 in flight"
timeout -k 10 200 python3 bench.py --steps 100 --warmup 10 --fused 5 --no-cpu-baseline --no-paths --no-poses 2>/dev/null | show "new form 5, 3 in flight"
