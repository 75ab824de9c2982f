#!/bin/bash
# On the GPU box: scripts/r04/paths_start_ab.py for the default library and every variant under blok_amd/variants/.  TAG names the output.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04
for lib in blok_amd/libblok_hip.so blok_amd/variants/*.so; do
  [ -f "$lib" ] || continue
  echo "== $lib"
  BLOK_HIP_LIB=$PWD/$lib timeout -k 10 400 python3 scripts/r04/paths_start_ab.py ${SPP:-64} ${COMBOS:-00,10,01,11} ${POSES:-0,1,2} 2>&1 | grep -v "amdgpu.ids"
done 2>&1 | tee gpurun_out/r04/paths_start_${TAG:-x}.txt
