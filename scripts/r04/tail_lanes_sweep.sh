#!/bin/bash
# The tail pool cut by the number of lanes still walking: BLOK_TAIL_CAPS=a,b,N,M (hard caps in trips for bounce rounds / rounds over parked rays; a round also
# ends once N lanes or fewer walk, from M trips on), 64 spp, poses A and B.
set -o pipefail
mkdir -p gpurun_out/r04
OUT=gpurun_out/r04/tail_lanes_sweep.txt; : > $OUT
for caps in "24,32,0,12" "48,48,16,12" "48,48,8,12" "48,48,24,12" "48,48,32,12" "32,32,16,8" "64,64,16,16" "24,32,0,12"; do
  echo "== caps $caps" | tee -a $OUT
  BLOK_TAIL_CAPS=$caps timeout -k 5 200 python3 scripts/r04/tail_pool_check.py 64 0,1 2>&1 | grep -v amdgpu.ids | grep -E "mode 3|deviation" | awk 'NR%3!=1' | tee -a $OUT || exit 1
done
