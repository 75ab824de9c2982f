#!/usr/bin/env python3
"""Condenses gpurun_out/prof_<tag>/ (scripts/r03/profile.sh) into profiles/<tag>_kernel_stats*.csv, profiles/<tag>_pmc.json (raw counter
means per kernel) and profiles/<tag>_summary.json — the derived figures bench.py quotes: HBM bytes per launch from FETCH_SIZE / WRITE_SIZE
(KiB; FETCH_SIZE raw and with the guide's 2x wide-read correction as lower / upper bound, /opt/skills/guides/MI355X_MICROARCH.md §HBM),
VALU wave-instructions per launch, lane utilisation = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU), and the VALU issue interval
= SIMDs x kernel cycles / VALU instructions."""
import csv, json, re, statistics, subprocess, sys
from collections import defaultdict
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
src = ROOT / "gpurun_out" / f"prof_{tag}"
dst = ROOT / "profiles"
N_SIMD, CLOCK_GHZ = 1024, 2.4       # 256 CUs x 4 SIMDs; peak engine clock (MI355X_MICROARCH.md)

def copy_stats(sub, name):
    f = list(src.glob(f"{sub}/**/*kernel_stats.csv"))
    if f:
        (dst / f"{tag}_{name}.csv").write_text(f[0].read_text())
        return list(csv.DictReader(f[0].open()))
    return []

stats = copy_stats("trace", "kernel_stats")
copy_stats("trace2", "kernel_stats_pipelined")
orbit = copy_stats("trace3", "kernel_stats_orbit1")
paths = copy_stats("paths_trace", "paths_kernel_stats")

def durations(sub):
    d = defaultdict(list)
    for f in src.glob(f"{sub}/**/*kernel_trace.csv"):
        for row in csv.DictReader(f.open()):
            d[row["Kernel_Name"]].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    return d

def counters(prefix):
    pmc = defaultdict(lambda: defaultdict(list))
    for f in sorted(src.glob(f"{prefix}*/**/*counter_collection.csv")):
        for row in csv.DictReader(f.open()):
            pmc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in pmc.items()}

def derive(kernel, c, dur_ns):
    out = {"kernel": kernel, "duration_us_unprofiled": dur_ns / 1e3 if dur_ns else None}
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        out["hbm_bytes_per_launch"] = (c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
        out["hbm_bytes_per_launch_upper"] = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
        out["fetch_kib_raw"], out["write_kib"] = c["FETCH_SIZE"], c["WRITE_SIZE"]
    if "SQ_INSTS_VALU" in c:
        out["valu_wave_instructions_per_launch"] = c["SQ_INSTS_VALU"]
        out["salu_wave_instructions_per_launch"] = c.get("SQ_INSTS_SALU")
        out["vmem_wave_instructions_per_launch"] = c.get("SQ_INSTS_VMEM")
        out["waves_per_launch"] = c.get("SQ_WAVES")
        if dur_ns:
            cycles = dur_ns * CLOCK_GHZ
            out["valu_issue_interval_cycles_per_simd"] = N_SIMD * cycles / c["SQ_INSTS_VALU"]
            # against the interval the loop's instruction mix would need alone on a SIMD (static mix of the walk loop's ISA: 62 full-rate
            # at 2.2 cycles + 61 half-rate at 4.2 since the build without the SLP pass, 54 + 66 before it: scripts/microbench/valu_rate*.hip,
            # DESIGN.md §9): 1 = the VALU never waits.  A static mix — the frames in flight run below this interval — so a figure near 1 says
            # "the vector pipe is what the launch waits for", not more
            fr, hr = (62, 61) if tag >= "r04d" else (54, 66)
            out["valu_issue_frac"] = ((fr * 2.2 + hr * 4.2) / float(fr + hr)) / out["valu_issue_interval_cycles_per_simd"]
    if "SQ_THREAD_CYCLES_VALU" in c and c.get("SQ_ACTIVE_INST_VALU"):
        out["lane_utilisation"] = c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"])
    if c.get("TCC_REQ_sum"):
        out["l2_hit_rate"] = c.get("TCC_HIT_sum", 0.0) / c["TCC_REQ_sum"]
    return out

head = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
dirty = bool(subprocess.run(["git", "status", "--porcelain", "--", "blok_amd", "include", "bench.py"], cwd=ROOT, capture_output=True, text=True).stdout.strip())
summary = {"tag": tag, "commit": head + ("+uncommitted changes" if dirty else ""),
           "command": "python3 bench.py --steps 60 --warmup 10 --settle 32 --frames-in-flight 1 --no-cpu-baseline --no-paths --no-poses (scripts/r03/profile.sh)",
           "kernels": {}, "paths": {}}
raw = {}
dur = durations("trace")
for kernel, c in counters("pmc").items():
    if not any(n in kernel for n in ("trace_kernel", "beam_kernel", "joint_kernel", "list_")):
        continue
    raw[kernel] = c
    d = dur.get(kernel)
    short = re.search(r"(\w+_kernel)", kernel).group(1)
    # the median launch: the first launches of a run (before a view's order is in force, api.hip) are not the steady state the counters average over either
    summary["kernels"][short] = derive(kernel, c, statistics.median(d) if d else None) | {"launches_traced": len(d) if d else 0, "duration_us_mean": sum(d) / len(d) / 1e3 if d else None,
                                                                                          "duration_us_min": min(d) / 1e3 if d else None}
pdur = durations("paths_trace")
for kernel, c in counters("paths_pmc").items():
    if "path_kernel" not in kernel:
        continue
    raw[kernel + " [8 spp]"] = c
    d = pdur.get(kernel)
    e = derive(kernel, c, None)
    e["note"] = "counters of 8-spp launches (4K, 2 bounces); duration of 64-spp launches"
    e["duration_us_64spp"] = sum(d) / len(d) / 1e3 if d else None
    summary["paths"]["path_kernel"] = e
for row in orbit:
    if any(n in row["Name"] for n in ("joint_kernel", "trace_kernel", "beam_kernel", "order_class_kernel", "order_rows_kernel", "order_scatter_kernel")):
        summary.setdefault("orbit1", {})[re.search(r"(\w+_kernel)", row["Name"]).group(1)] = {"calls": int(row["Calls"]), "average_us": float(row["AverageNs"]) / 1e3}
(dst / f"{tag}_pmc.json").write_text(json.dumps(raw, indent=1))
(dst / f"{tag}_summary.json").write_text(json.dumps(summary, indent=1))
print(json.dumps(summary, indent=1))
