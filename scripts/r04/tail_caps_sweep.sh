#!/bin/bash
# The tail pool's two caps swept: BLOK_TAIL_CAPS=a,b (trips of a bounce round / of a round over parked rays), 64 spp, poses A and B.
set -o pipefail
mkdir -p gpurun_out/r04
OUT=gpurun_out/r04/tail_caps_sweep.txt; : > $OUT
for caps in ${CAPS:-"24,32" "16,24" "16,32" "24,24" "32,32" "24,48" "32,48" "48,48" "12,24"}; do
  echo "== caps $caps" | tee -a $OUT
  BLOK_TAIL_CAPS=$caps timeout -k 5 200 python3 scripts/r04/tail_pool_check.py 64 0,1 2>&1 | grep -v amdgpu.ids | grep -E "mode 3|deviation" | awk 'NR%3!=1' | tee -a $OUT || exit 1
done
