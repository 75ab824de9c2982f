#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03
show() { python3 -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); c = d['config']
print('$1:', round(d['value']), 'Mrays/s', round(d['ms_per_step'], 4), 'ms/frame; alone', round(c['kernel_ms_alone'], 4), 'ms;', {k: (round(v['Mrays_per_s']), round(v['ms_per_frame_alone'], 4)) for k, v in c['poses'].items()}, 'gave up', c['walk_waves_that_gave_up_waiting'])"; }
{
for form in 4 5 2 0; do
  for cls in 1 0; do
    [ $cls = 0 ] && [ $form != 4 ] && [ $form != 5 ] && continue
    timeout -k 10 120 python3 bench.py --steps 60 --warmup 5 --fused $form --list-classes $cls --frames-in-flight 1 --no-cpu-baseline --no-paths --no-poses 2>/dev/null | show "form $form classes $cls static, 1 in flight"
    timeout -k 10 120 python3 bench.py --steps 60 --warmup 5 --fused $form --list-classes $cls --frames-in-flight 1 --orbit 1 --no-cpu-baseline --no-paths --no-poses 2>/dev/null | show "form $form classes $cls orbit 1, 1 in flight"
  done
done
timeout -k 10 200 python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-paths 2>/dev/null | show "default (form 3), 3 in flight"
timeout -k 10 200 python3 bench.py --steps 100 --warmup 10 --orbit 1 --no-cpu-baseline --no-paths --no-poses 2>/dev/null | show "default (form 3), 3 in flight, orbit 1"
timeout -k 10 200 python3 bench.py --steps 100 --warmup 10 --fused 0 --no-cpu-baseline --no-paths --no-poses 2>/dev/null | show "form 0, 3 in flight"
} 2>&1 | tee gpurun_out/r03/list_classes_ab.txt
