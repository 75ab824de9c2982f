#!/bin/bash
# On the GPU box: kernel trace of a camera orbiting 1 degree per frame with ONE frame in flight — the frame's kernel and what keeps its
# order up to date behind it (snapshot, dilation, sort, finish), with the carried order on and off.  Output under gpurun_out/r03d/.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03d; mkdir -p $OUT
for mo in 1 0; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/orbit_trace_mo$mo -o t -- python3 bench.py --steps 120 --warmup 10 --frames-in-flight 1 --no-cpu-baseline --no-paths --no-poses --orbit 1 --moving-order $mo > $OUT/orbit_trace_mo$mo.json 2> $OUT/orbit_trace_mo$mo.err || { echo failed; tail -5 $OUT/orbit_trace_mo$mo.err; exit 1; }
  python3 scripts/r03/bench_fields.py "orbit 1, one frame in flight, moving-order $mo (under rocprofv3)" < $OUT/orbit_trace_mo$mo.json
  f=$(find $OUT/orbit_trace_mo$mo -name '*kernel_stats.csv' | head -1); cut -d, -f1-4,6-8 "$f" | sed 's/(anonymous namespace):://; s/void blok:://' | cut -c1-150 | head -14
done
