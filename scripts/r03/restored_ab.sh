#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03
show() { python3 -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); c = d['config']
print('$1:', round(d['value']), 'Mrays/s', round(d['ms_per_step'], 4), 'ms/frame; alone', round(c['kernel_ms_alone'], 4), 'ms;', {k: (round(v['Mrays_per_s']), round(v['ms_per_frame_alone'], 4)) for k, v in c['poses'].items()}, 'gave up', c['walk_waves_that_gave_up_waiting'])"; }
{
timeout -k 10 200 python3 bench.py --steps 100 --warmup 10 --settle 32 --no-cpu-baseline --no-paths 2>/dev/null | show "default, settle 32"
timeout -k 10 200 python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-paths --no-poses 2>/dev/null | show "default, settle 0"
timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-paths --no-poses 2>/dev/null | show "default, steps 20 warmup 5"
timeout -k 10 200 python3 bench.py --steps 100 --warmup 10 --orbit 1 --no-cpu-baseline --no-paths --no-poses 2>/dev/null | show "default, orbit 1"
timeout -k 10 200 python3 bench.py --steps 100 --warmup 10 --frames-in-flight 1 --no-cpu-baseline --no-paths --no-poses 2>/dev/null | show "default, 1 in flight"
timeout -k 10 200 python3 bench.py --steps 100 --warmup 10 --frames-in-flight 1 --orbit 1 --no-cpu-baseline --no-paths --no-poses 2>/dev/null | show "default, 1 in flight, orbit 1"
} 2>&1 | tee gpurun_out/r03/restored_ab.txt
