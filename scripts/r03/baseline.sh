#!/bin/bash
# Round-3 baseline on the GPU box: tests, bench lines (default, orbit), path-kernel kernel trace + PMC passes.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03_base; rm -rf $OUT; mkdir -p $OUT
echo "== tests" && timeout -k 10 400 python3 -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/tests.log
echo "== bench" && timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "rc=$?"
echo "== bench orbit" && timeout -k 10 200 python3 bench.py --steps 60 --warmup 5 --orbit 1 --settle 0 --no-poses --no-paths --no-cpu-baseline > $OUT/bench_orbit.json 2> $OUT/bench_orbit.err; echo "rc=$?"
echo "== paths trace" && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/paths_trace -o paths -- python3 scripts/r03/profile_paths64.py 64 3 > $OUT/paths_trace.log 2>&1; echo "rc=$?"
i=0
for PMC in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum SQ_INSTS_FLAT SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  echo "== paths pmc $i: $PMC"
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/paths_pmc$i -o pmc -- python3 scripts/r03/profile_paths64.py 8 3 > $OUT/paths_pmc$i.log 2>&1 || tail -3 $OUT/paths_pmc$i.log
done
python3 - <<'PY'
import csv, glob, collections, json
acc = collections.defaultdict(list)
for f in glob.glob('gpurun_out/r03_base/paths_pmc*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'path_kernel' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
out = {k: sum(v) / len(v) for k, v in sorted(acc.items())}
json.dump(out, open('gpurun_out/r03_base/paths_pmc_8spp.json', 'w'), indent=1)
print(out)
PY
find $OUT -name "*stats*.csv" | head; du -sh $OUT
