#!/bin/bash
# On the GPU box: round 3's library (blok_amd/variants/libblok_hip_r3.so, built from 41941c5) against this tree's, same box: the launch alone for a
# camera orbiting by 1 degree per frame (scripts/r03/solitary_orbit.py), and the default bench line (pipelined rate, launch alone at rest).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04
for rep in 1 2; do
for lib in blok_amd/variants/libblok_hip_r3.so blok_amd/libblok_hip.so; do
  echo "== $lib rep $rep"
  BLOK_HIP_LIB=$PWD/$lib timeout -k 10 200 python3 scripts/r03/solitary_orbit.py 1 64 1 2>&1 | grep -v amdgpu.ids | tail -3
  BLOK_HIP_LIB=$PWD/$lib timeout -k 10 300 python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-paths --no-poses 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); r = d['roofline']
print('bench:', round(d['value']), 'Mrays/s', round(d['ms_per_step'], 4), 'ms/frame; alone', round(r['kernel_ms'], 4), 'moving', round(r['kernel_ms_moving'], 4), 'alternating', r.get('kernel_ms_alternating'), 'cold', r.get('kernel_ms_cold'))"
done; done 2>&1 | tee gpurun_out/r04/ab_r3.txt
