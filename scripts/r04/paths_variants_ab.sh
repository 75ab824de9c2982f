#!/bin/bash
# On the GPU box: the path kernel at BASELINE configs[4] (4K, 64 spp, 2 bounces) for the default library and every variant under
# blok_amd/variants/ (scripts/build_variant.sh), poses A/B/C, HIP events around the kernel.  TAG names the output file.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04
SPP=${SPP:-64}
for lib in blok_amd/libblok_hip.so blok_amd/variants/*.so; do
  echo "== $lib"
  BLOK_HIP_LIB=$PWD/$lib timeout -k 10 240 python3 scripts/r03/paths_ab.py $SPP 2 2>&1 | grep -v Warning
done 2>&1 | tee gpurun_out/r04/paths_variants_ab_${TAG:-x}.txt
