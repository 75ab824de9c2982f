#!/usr/bin/env python3
"""Condenses gpurun_out/prof_<tag>/ (rocprofv3 csv output) into profiles/<tag>_kernel_stats.csv,
profiles/<tag>_pmc.json and profiles/pmc_traffic.json (HBM bytes per launch of the dominant kernel, with the
gfx950 FETCH_SIZE correction of /opt/skills/guides/MI355X_MICROARCH.md §HBM applied)."""
import csv
import json
import sys
from collections import defaultdict
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = ROOT / "gpurun_out" / f"prof_{tag}"
dst = ROOT / "profiles"
dst.mkdir(exist_ok=True)

stats = list(src.glob("trace/**/*kernel_stats.csv"))
if stats:
    text = stats[0].read_text()
    (dst / f"{tag}_kernel_stats.csv").write_text(text)
    print(text)
stats2 = list(src.glob("trace2/**/*kernel_stats.csv"))
if stats2:
    (dst / f"{tag}_kernel_stats_pipelined.csv").write_text(stats2[0].read_text())
trace = list(src.glob("trace/**/*kernel_trace.csv"))
durations = defaultdict(list)
if trace:
    for row in csv.DictReader(trace[0].open()):
        durations[row["Kernel_Name"]].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))

pmc = defaultdict(lambda: defaultdict(list))
for f in sorted(src.glob("pmc*/**/*counter_collection.csv")):
    for row in csv.DictReader(f.open()):
        pmc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
summary = {}
for kernel, counters in pmc.items():
    if not any(name in kernel for name in ("trace_kernel", "beam_kernel", "frame_kernel", "joint_kernel")):
        continue
    summary[kernel] = {c: {"mean": sum(v) / len(v), "n": len(v)} for c, v in counters.items()}
    d = durations.get(kernel)
    if d:
        summary[kernel]["duration_ns_mean_unprofiled_pass"] = sum(d) / len(d)
        summary[kernel]["launches"] = len(d)
(dst / f"{tag}_pmc.json").write_text(json.dumps(summary, indent=1))
print(json.dumps(summary, indent=1))
# HBM traffic per FRAME: FETCH_SIZE + WRITE_SIZE of every kernel of the frame's launch sequence (beam_kernel + trace_kernel of the
# Rect mode, frame_kernel in the one-launch form, or joint_kernel — whichever the profiled run launched), per launch.  rocprofv3 reports
# both in KiB.
frame_kernels = {k: s for k, s in summary.items() if ("RayModeE0" in k or "(blok::RayMode)0" in k) and "FETCH_SIZE" in s and "WRITE_SIZE" in s}
if frame_kernels:
    fetch_kb = sum(s["FETCH_SIZE"]["mean"] for s in frame_kernels.values())
    write_kb = sum(s["WRITE_SIZE"]["mean"] for s in frame_kernels.values())
    out = {"kernels": {k: {"fetch_size_kb_raw": s["FETCH_SIZE"]["mean"], "write_size_kb": s["WRITE_SIZE"]["mean"],
                           "duration_ns_mean": s.get("duration_ns_mean_unprofiled_pass")} for k, s in frame_kernels.items()},
           "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md §HBM only for wide coalesced streams; these kernels' reads are "
                   "16-B node gathers (uncalibrated width), so the raw value is reported as a lower bound and 2x as an upper bound",
           "hbm_bytes_per_frame": (fetch_kb + write_kb) * 1024.0,
           "hbm_bytes_per_frame_upper": (2 * fetch_kb + write_kb) * 1024.0,
           "frame_kernel_ns_sum": sum(s.get("duration_ns_mean_unprofiled_pass", 0.0) for s in frame_kernels.values())}
    (dst / "pmc_traffic.json").write_text(json.dumps(out, indent=1))
    print(json.dumps(out, indent=1))
