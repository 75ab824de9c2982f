#!/bin/bash
# A/B on the GPU box: frames in flight with the searches of consecutive frames one behind the other (variant library, -DBLOK_EXP_SERIAL_BEAMS)
# against the shipped library; the default bench and the driver's arguments, two repetitions each, alternating.
set -o pipefail
mkdir -p gpurun_out/r04
OUT=gpurun_out/r04/serial_beams_ab.txt; : > $OUT
for rep in 1 2; do
  for lib in blok_amd/libblok_hip.so blok_amd/variants/libblok_hip_serialbeams.so; do
    for a in "--steps 200 --warmup 10" "--steps 20 --warmup 5"; do
      BLOK_HIP_LIB=$PWD/$lib python3 bench.py $a --no-cpu-baseline --no-paths --no-poses 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1])
print('$lib', '$a', 'Mrays/s %.0f' % d['value'], 'ms/step %.4f' % d['ms_per_step'], 'alone %.4f' % d['config']['kernel_ms_alone'])" | tee -a $OUT || exit 1
    done
  done
done
