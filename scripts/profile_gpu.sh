#!/bin/bash
# Runs on the GPU box (via gpurun): kernel trace + separate PMC passes over the default bench command.
# Output: gpurun_out/prof_<tag>/...   then scripts/summarize_profile.py turns it into profiles/<tag>_*.
set -o pipefail
TAG=${1:-r02}
STEPS=${2:-20}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
# one frame in flight: launches do not overlap, so per-kernel durations and counters are those of a launch running alone
CMD="python3 bench.py --steps $STEPS --warmup 3 --frames-in-flight 1 --no-cpu-baseline --no-paths --no-poses"
echo "== kernel trace" && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1 || { echo trace failed; tail -5 $OUT/trace.log; exit 1; }
echo "== kernel trace, default bench configuration (frames in flight)" && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace2 -o trace2 -- python3 bench.py --steps $STEPS --warmup 3 --no-cpu-baseline --no-paths --no-poses > $OUT/trace2.log 2>&1 || echo "trace2 failed"
i=0
for PMC in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM" \
           "GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  echo "== pmc pass $i: $PMC"
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/pmc$i -o pmc -- $CMD > $OUT/pmc$i.log 2>&1 || { echo "pmc pass $i failed"; tail -5 $OUT/pmc$i.log; }
done
find $OUT -name "*.csv" | head -40
du -sh $OUT
