#!/bin/bash
# On the GPU box: rocprofv3 kernel trace + separate PMC passes of the default bench workload (one frame in flight, so kernel durations and
# counters are those of a launch running alone) and of the path kernel at BASELINE configs[4].  Output: gpurun_out/prof_<tag>/; then
# scripts/r03/summarize.py <tag> turns it into profiles/<tag>_*.
set -o pipefail
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
CMD="python3 bench.py --steps 60 --warmup 10 --settle 32 --frames-in-flight 1 --no-cpu-baseline --no-paths --no-poses"
echo "== kernel trace, one frame in flight" && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1 || { echo trace failed; tail -5 $OUT/trace.log; exit 1; }
echo "== kernel trace, default bench configuration" && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace2 -o trace2 -- python3 bench.py --steps 60 --warmup 10 --settle 32 --no-cpu-baseline --no-paths --no-poses > $OUT/trace2.log 2>&1 || echo "trace2 failed"
echo "== kernel trace, camera orbiting 1 degree per frame, frames alone one at a time (the carried order and its upkeep kernels)" && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace3 -o trace3 -- python3 scripts/r03/solitary_orbit.py 1 96 1 > $OUT/trace3.log 2>&1 || echo "trace3 failed"
echo "== kernel trace, path kernel 64 spp" && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/paths_trace -o paths -- python3 scripts/r03/profile_paths64.py 64 3 > $OUT/paths_trace.log 2>&1 || echo "paths trace failed"
i=0
for PMC in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU" \
           "GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  echo "== pmc pass $i: $PMC"
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/pmc$i -o pmc -- $CMD > $OUT/pmc$i.log 2>&1 || { echo "pmc pass $i failed"; tail -5 $OUT/pmc$i.log; }
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/paths_pmc$i -o pmc -- python3 scripts/r03/profile_paths64.py 8 3 > $OUT/paths_pmc$i.log 2>&1 || { echo "paths pmc pass $i failed"; tail -5 $OUT/paths_pmc$i.log; }
done
du -sh $OUT
