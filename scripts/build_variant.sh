#!/bin/bash
# Builds an A/B variant of libblok_hip.so: scripts/build_variant.sh <name> [extra hipcc flags...]
set -e
NAME=$1; shift
cd "$(dirname "$0")/.."
hipcc --offload-arch=gfx950 -O3 -std=c++20 -ffp-contract=off -fPIC -shared -Iinclude -Iblok_amd/csrc/hip "$@" \
  -o blok_amd/variants/libblok_hip_$NAME.so blok_amd/csrc/hip/api.hip blok_amd/csrc/hip/api_post.hip blok_amd/csrc/hip/api_volume.hip blok_amd/csrc/hip/api_multi.hip blok_amd/csrc/hip/trace_kernels.hip blok_amd/csrc/hip/dense_kernels.hip blok_amd/csrc/hip/tile_order.hip blok_amd/csrc/hip/gpu_build.hip blok_amd/csrc/hip/post_kernels.hip blok_amd/csrc/hip/tree_build.cpp
echo built blok_amd/variants/libblok_hip_$NAME.so
