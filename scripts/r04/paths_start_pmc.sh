#!/bin/bash
# On the GPU box: rocprofv3 counters of the path kernel (8 spp, 4K, pose A) for three settings of blok_hip_set_path_start — round 3's starts (from the
# root, 32x32 beam only), + the wave tile's own beam, + rays entered from the anchor — one --pmc pass per counter group (program directly after --).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04/paths_start_pmc; rm -rf $OUT; mkdir -p $OUT
for combo in "0 0" "0 1" "1 1"; do
  tag=$(echo $combo | tr -d ' ')
  i=0
  for PMC in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/c${tag}_p$i -o pmc -- python3 scripts/r04/paths_start_one.py 8 3 $combo > $OUT/c${tag}_p$i.log 2>&1 || echo "pass $tag $i failed"
  done
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c${tag}_trace -o trace -- python3 scripts/r04/paths_start_one.py 64 3 $combo > $OUT/c${tag}_trace.log 2>&1 || echo "trace $tag failed"
done
python3 - <<'PY'
import csv, glob, json
from collections import defaultdict
out = {}
for tag in ("00", "01", "11"):
    c = defaultdict(list)
    for f in glob.glob(f"gpurun_out/r04/paths_start_pmc/c{tag}_p*/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "path_kernel" in row["Kernel_Name"]: c[row["Counter_Name"]].append(float(row["Counter_Value"]))
    m = {k: sum(v) / len(v) for k, v in c.items()}
    d = []
    for f in glob.glob(f"gpurun_out/r04/paths_start_pmc/c{tag}_trace/**/*kernel_trace.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "path_kernel" in row["Kernel_Name"]: d.append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    e = {"resume": int(tag[0]), "wave_tile_beam": int(tag[1]), "counters_8spp": m, "duration_ms_64spp": sum(d) / len(d) / 1e6 if d else None}
    if m.get("SQ_ACTIVE_INST_VALU"): e["lane_utilisation"] = m["SQ_THREAD_CYCLES_VALU"] / (64.0 * m["SQ_ACTIVE_INST_VALU"])
    if "WRITE_SIZE" in m: e["write_GB_per_8spp_launch"] = m["WRITE_SIZE"] * 1024 / 1e9; e["fetch_GB_per_8spp_launch_raw"] = m.get("FETCH_SIZE", 0) * 1024 / 1e9
    out[tag] = e
json.dump(out, open("gpurun_out/r04/paths_start_pmc.json", "w"), indent=1)
for tag, e in out.items():
    m = e["counters_8spp"]
    print(tag, "64 spp", e["duration_ms_64spp"], "ms; VALU", m.get("SQ_INSTS_VALU"), "SALU", m.get("SQ_INSTS_SALU"), "VMEM", m.get("SQ_INSTS_VMEM"), "LDS", m.get("SQ_INSTS_LDS"), "lane util", e.get("lane_utilisation"), "write GB", e.get("write_GB_per_8spp_launch"))
PY
