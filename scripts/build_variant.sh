#!/bin/bash
# Builds an A/B variant of libblok_hip.so: scripts/build_variant.sh <name> [extra hipcc flags...]
# Only the files named in VARIANT_SRC (default: trace_kernels.hip) are recompiled with the extra flags; the other translation units are
# compiled once with the product's flags into build/variant_obj/ and reused (no device code crosses a translation unit).
set -e
NAME=$1; shift
cd "$(dirname "$0")/.."
SRC_DIR=blok_amd/csrc/hip
ALL="api.hip api_launch.hip api_debug.hip api_post.hip api_volume.hip api_multi.hip trace_kernels.hip dense_kernels.hip tile_order.hip gpu_build.hip post_kernels.hip tree_build.cpp"
VARIANT_SRC=${VARIANT_SRC:-trace_kernels.hip}
FLAGS="--offload-arch=gfx950 -O3 -std=c++20 -ffp-contract=off ${BASE_EXTRA--fno-slp-vectorize} -fPIC -Iinclude -Iblok_amd/csrc/hip"
OBJ=build/variant_obj; mkdir -p $OBJ blok_amd/variants
pids=()
for f in $ALL; do
  case " $VARIANT_SRC " in
    *" $f "*) hipcc $FLAGS "$@" -c $SRC_DIR/$f -o $OBJ/${f%.*}_$NAME.o & pids+=($!) ;;
    *) if [ ! -f $OBJ/${f%.*}.o ] || [ -n "$(find $SRC_DIR include blok_amd/csrc/common -newer $OBJ/${f%.*}.o \( -name '*.h' -o -name '*.hip' -o -name '*.cpp' -o -name '*.hpp' \) | head -1)" ]; then
         hipcc $FLAGS -c $SRC_DIR/$f -o $OBJ/${f%.*}.o & pids+=($!)
       fi ;;
  esac
done
for p in "${pids[@]}"; do wait $p; done
objs=""
for f in $ALL; do
  case " $VARIANT_SRC " in *" $f "*) objs="$objs $OBJ/${f%.*}_$NAME.o" ;; *) objs="$objs $OBJ/${f%.*}.o" ;; esac
done
hipcc --offload-arch=gfx950 -shared -fPIC -o blok_amd/variants/libblok_hip_$NAME.so $objs -ldl
echo built blok_amd/variants/libblok_hip_$NAME.so
