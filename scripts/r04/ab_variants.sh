#!/bin/bash
# On the GPU box: default library and every variant under blok_amd/variants/: pipelined rate and the launch alone for the three poses, twice.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04
for rep in 1 2; do
for lib in blok_amd/libblok_hip.so blok_amd/variants/*.so; do
  [ -f "$lib" ] || continue
  BLOK_HIP_LIB=$PWD/$lib timeout -k 10 300 python3 bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-paths ${EXTRA} 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); c = d['config']
print('$lib rep$rep:', round(d['value']), 'Mrays/s; alone', round(c['kernel_ms_alone'], 4), 'ms; moving', round(c['kernel_ms_alone_moving'] or 0, 4), {k: (round(v['Mrays_per_s']), round(v['ms_per_frame_alone'], 4)) for k, v in c['poses'].items()})"
done; done 2>&1 | tee gpurun_out/r04/ab_variants_${TAG:-x}.txt
