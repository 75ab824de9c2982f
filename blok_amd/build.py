"""In-tree builds: libblok_host.so (g++), libblok_hip.so (hipcc, gfx950 only)."""
from __future__ import annotations

import os
import shutil
import subprocess
from pathlib import Path

PKG = Path(__file__).resolve().parent
ROOT = PKG.parent
INCLUDE = ROOT / "include"

HOST_SRC = [PKG / "csrc/host/world.cpp", PKG / "csrc/host/scene.cpp", PKG / "csrc/host/vox.cpp"]
HIP_SRC = [PKG / "csrc/hip/api.hip", PKG / "csrc/hip/api_launch.hip", PKG / "csrc/hip/api_debug.hip", PKG / "csrc/hip/api_post.hip", PKG / "csrc/hip/api_volume.hip", PKG / "csrc/hip/api_multi.hip", PKG / "csrc/hip/trace_kernels.hip", PKG / "csrc/hip/dense_kernels.hip", PKG / "csrc/hip/tile_order.hip", PKG / "csrc/hip/gpu_build.hip",
           PKG / "csrc/hip/post_kernels.hip", PKG / "csrc/hip/tree_build.cpp"]
HIP_HDR = sorted((PKG / "csrc/hip").glob("*.h")) + sorted((PKG / "csrc/common").glob("*.h")) + [INCLUDE / "blok_hip.h", INCLUDE / "blok_hip_debug.h", INCLUDE / "blok_world.h"]


def _stale(target: Path, deps) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(Path(d).stat().st_mtime > t for d in deps)


def _run(cmd):
    proc = subprocess.run([os.fspath(c) for c in cmd], capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError("build failed: " + " ".join(map(str, cmd)) + "\n" + proc.stdout + proc.stderr)


def hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found")


def build_host(force: bool = False) -> Path:
    out = PKG / "libblok_host.so"
    if force or _stale(out, HOST_SRC + sorted((PKG / "csrc/common").glob("*.h")) + [INCLUDE / "blok_world.h", INCLUDE / "blok_hip.h"]):
        _run(["g++", "-O2", "-std=c++20", "-fPIC", "-ffp-contract=off", "-Wall", "-Wextra", f"-I{INCLUDE}",
              "-shared", "-o", out, *HOST_SRC])
    return out


def build_hip(force: bool = False) -> Path:
    """One object per translation unit (no device code crosses one), compiled in parallel into build/obj/, then linked."""
    from concurrent.futures import ThreadPoolExecutor
    out = PKG / "libblok_hip.so"
    if not (force or _stale(out, HIP_SRC + HIP_HDR)):
        return out
    obj_dir = ROOT / "build" / "obj"
    obj_dir.mkdir(parents=True, exist_ok=True)
    # -fno-slp-vectorize: left to itself the SLP pass pairs the walk's float adds / multiplies / fmas into v_pk_*_f32; on gfx950 a packed
    # instruction issues at 1.64x the cost of a plain one for its two results (scripts/microbench/valu_rate2.hip), and gathering operands into
    # register pairs costs moves and registers (joint_kernel 67 -> 61 VGPRs: eight waves per SIMD instead of seven).  Measured: frame alone
    # 0.198 -> 0.191 ms, three in flight 0.177 -> 0.167 ms, configs[4] 45.0 -> 40.3 ms (profiles/r04_no_slp_ab.txt).  Results are bit-identical.
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++20", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", f"-I{INCLUDE}", f"-I{PKG / 'csrc/hip'}"]
    jobs = []
    for src in HIP_SRC:
        obj = obj_dir / (src.stem + ".o")
        if force or _stale(obj, [src] + HIP_HDR):
            jobs.append([hipcc(), *flags, "-c", src, "-o", obj])
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        list(pool.map(_run, jobs))
    _run([hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *[obj_dir / (src.stem + ".o") for src in HIP_SRC], "-ldl"])
    return out


def build_oracle() -> Path:
    """Test infrastructure (tests/, smoke, cpu_baseline only)."""
    _run(["make", "-C", ROOT / "oracle"])
    return ROOT / "oracle" / "liboracle.so"


def build_tools(force: bool = False) -> Path:
    """Headless C++ driver (tools/blok_headless.cpp) over include/blok/hip_tracer.hpp."""
    out = ROOT / "tools" / "blok_headless"
    src = ROOT / "tools" / "blok_headless.cpp"
    if force or _stale(out, [src, INCLUDE / "blok" / "hip_tracer.hpp", INCLUDE / "blok_hip.h", INCLUDE / "blok_world.h"]):
        _run(["g++", "-O2", "-std=c++20", "-Wall", "-Wextra", f"-I{INCLUDE}", "-o", out, src, f"-L{PKG}",
              "-lblok_hip", "-lblok_host", "-Wl,-rpath,$ORIGIN/../blok_amd"])
    return out


def build_all(force: bool = False):
    host, hip = build_host(force), build_hip(force)
    build_tools(force)
    return host, hip, build_oracle()
