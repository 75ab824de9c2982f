#!/usr/bin/env python3
"""Benchmark of the hot path: primary first-hit rays at 3840x2160 over the synthetic 1024^3 SVO
(BASELINE.json configs[2], the configuration the metric is quoted on).

One step = one full frame.  N = 1: one launch of the trace kernel over the whole frame.
A step produces the 16-B first-hit records and the RGBA8 framebuffer of the frame.
N > 1 (one process per GPU, launched by torch.distributed.run): the frame is cut into 32x32 tiles dealt
round-robin to the ranks, every rank traces its tiles from its own replica of the world (hit records stay on
the rank), one RCCL gather per frame brings the RGBA8 tiles to rank 0, which un-permutes them into the
framebuffer; the gather of frame k runs under the trace of frame k+1 (blok_amd/multi_gpu.py; the reference
has no multi-GPU path).  Total work per step is fixed as N grows -> "scaling": "strong".

Prints ONE JSON line on rank 0.  Inputs are synthetic (generator G(N, seed) of SURVEY.md §8(d)) and are
resident in HBM before the timed region.

`python bench.py --gpus N` from a bare shell (no WORLD_SIZE in the environment) starts the N ranks itself, as child
processes under torch.distributed.run, BEFORE this process touches the GPU, and relays rank 0's JSON line.

roofline.frac is quoted on the frame's launch running ALONE on the chip (HIP events around single launches, one at a
time); the throughput-derived figure with several frames in flight is reported beside it as frac_overlapped.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--n", type=int, default=1024, help="world edge in voxels")
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--pose", type=int, default=0, help="camera pose A/B/C = 0/1/2 (SURVEY.md §8(d))")
    ap.add_argument("--tile", type=int, default=32)
    ap.add_argument("--settle", type=int, default=32, help="untimed frames before the warmup steps: the longest-first order of a view at rest is per-view state (measured from the second "
                    "frame of a view, sorted behind it, adopted a few frames later, re-sorted ever less often), prepared like the world upload; 0 = start cold "
                    "(-4 %% over 20 frames: profiles/r03c_settle_frames.txt).  Reported in config.settle_frames and roofline.timing; roofline.frac_moving is the figure that uses no such state")
    ap.add_argument("--seed", type=lambda s: int(s, 0), default=0xB10C0001)
    ap.add_argument("--cpu-stride", type=int, default=1, help="CPU baseline traces every stride-th pixel in x and y")
    ap.add_argument("--frames-in-flight", type=int, default=0,
                    help="pipeline depth; 0 = 3 for N <= 2, else 4 (reference: MAX_FRAMES_IN_FLIGHT = 2; at N = 1 three measure +3-4 %% over two). "
                         "Measured on one GPU with a rank's tile share (scripts/tile_depth_probe.py): a frame-share is a "
                         "beam + trace launch pair whose latency (~130 us alone) far exceeds its work (26-105 us), so "
                         "3-4 frames must be in flight to hide it")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only for "
                    "rehearsing the N > 1 code path with several ranks on one GPU, which RCCL refuses)")
    ap.add_argument("--beam", type=int, default=32, help="beam pre-pass tile in pixels (0 = off)")
    ap.add_argument("--fused", type=int, default=3, help="launch form of a frame (blok_hip_set_fused): 0 = beam kernel then trace kernel, 1 = one persistent launch with work queues (measured slower), "
                    "2 = joint launch (searches and one walk wave per wave tile in one grid), 4 = list-fed joint launch (walk waves take the live wave tiles from the list the frame's searches publish), "
                    "5 = beam kernel, then list-fed walk, 3 = automatic: 2 when a launch has the device to itself, else 0 (either over the live prefix of the view's order when one is in force)")
    ap.add_argument("--beam-budget", type=int, default=0, help="node visits a beam search may spend (0 = the library's default, 256); running out is answered conservatively")
    ap.add_argument("--moving-order", type=int, default=1, help="camera in motion, launch alone on the device: walk in the previous frame's dilated order carried over by a whole-tile shift (0 = row-major)")
    ap.add_argument("--tile-ordering", type=int, default=8, help="camera at rest: longest-first scheduling of the walk from earlier frames' per-wave clocks, re-sorted every N launches (0 = off)")
    ap.add_argument("--list-classes", type=int, default=1, help="list launches: order the walk by the previous frame's measured cost in four classes (0 = the order the searches finish in)")
    ap.add_argument("--orbit", type=float, default=0.0, help="degrees the camera turns around the world's centre per frame (0 = static camera)")
    ap.add_argument("--dense-dda", action="store_true", help="BASELINE configs[1]: upload the scene as a dense id grid and trace with the dense-grid kernel (N = 1, --n <= 512)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-paths", action="store_true", help="skip the 64-spp path-tracing side measurement")
    ap.add_argument("--no-poses", action="store_true", help="skip the per-pose side measurements (poses A, B, C)")
    ap.add_argument("--gather-frames", type=int, default=0, help="N > 1: consecutive frames a rank traces in one launch pair and exchanges in one collective (0 = N, at most 8: every launch pair then has the size of a single-GPU frame)")
    ap.add_argument("--sparse-gather", type=int, default=2, help="N > 1: 0 = dense RGBA8 gather + un-permute; 1 = only the tiles with a non-sky pixel travel, as RGBA8; "
                    "2 = those tiles as 16-bit (material, face) codes that the root expands with the material table (half the bytes; falls back to 1 for very large tables)")
    return ap.parse_args()


def spawn_ranks(n: int) -> int:
    """Bare `python bench.py --gpus N`: run the N ranks as children (one per GPU) and relay rank 0's line.  The parent
    has not touched the GPU (torch is not even imported yet); nothing is exec'ed."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.fspath(Path(__file__).resolve()), *sys.argv[1:]]
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    if lines:
        print(lines[-1], flush=True)
    else:
        sys.stdout.write(proc.stdout)
    return proc.returncode if lines or proc.returncode else 1


def build_world(n: int, seed: int):
    from blok_amd import world as W
    cm = W.ChunkManager(128, 1.0)
    cm.generate_scene(n, seed)
    cm.rebuild_dirty_chunks()
    return cm.pack_chunks_to_gpu_svo(W.scene_materials(seed))


def build_world_shared(n: int, seed: int, dist, rank: int):
    """N > 1: ONE CPU build per job instead of one per rank — rank 0 builds the world and broadcasts the three arrays of WorldSvoGpu
    (149 MB of SvoNodes for 1024^3) over the process group; every rank then uploads its replica (replicated world)."""
    if dist is None:
        return build_world(n, seed)
    import torch
    from blok_amd import world as W
    from blok_amd._ffi import MATERIAL, SUB_CHUNK, SVO_NODE
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    arrays = None
    sizes = torch.zeros(3, dtype=torch.int64, device=dev)
    if rank == 0:
        packed = build_world(n, seed)
        arrays = [np.ascontiguousarray(a).view(np.uint8).reshape(-1) for a in (packed.nodes, packed.sub_chunks, packed.materials)]
        sizes = torch.tensor([a.size for a in arrays], dtype=torch.int64, device=dev)
    dist.broadcast(sizes, src=0)
    got = []
    for i in range(3):
        t = torch.from_numpy(arrays[i]).to(dev) if rank == 0 else torch.empty(int(sizes[i]), dtype=torch.uint8, device=dev)
        dist.broadcast(t, src=0)
        got.append(t.cpu().numpy())
    if rank == 0:
        return packed
    return W.PackedWorld(got[0].view(SVO_NODE), got[1].view(SUB_CHUNK), got[2].view(MATERIAL))


def usable_cores() -> int:
    """Host cores this process may actually use: the affinity mask, capped by the cgroup CPU quota (a GPU box of the pool shows
    256 hardware threads but grants a 16-CPU share per GPU; threads beyond the quota are only throttled)."""
    try:
        n = max(1, len(os.sched_getaffinity(0)))
    except AttributeError:
        n = max(1, os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            text = Path(path).read_text().split()
            if path.endswith("cpu.max"):
                quota, period = text[0], text[1]
            else:
                quota, period = text[0], Path("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read_text().split()[0]
            if quota not in ("max", "-1"):
                n = max(1, min(n, int(float(quota) / float(period) + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def oracle_counters(packed, cam, width, height, stride=1):
    """Per-ray algorithmic byte count of SURVEY.md §8(d), by the oracle's counting build (formulation 2: restated
    intersect.rint behind a front-to-back slot lattice) on the same scene and camera."""
    from tests import oracle_ffi as O
    lattice = O.Lattice(packed.nodes, packed.sub_chunks)
    _, c = lattice.trace_primary(cam, width, height, stride=stride, threads=usable_cores(), want_hits=False)
    rays = int(c["rays"])
    alg_bytes = 48 * int(c["sub_chunks_entered"]) + 16 * int(c["nodes_fetched"]) + 32 * int(c["hits"]) + 16 * rays
    return {"bytes_per_ray": alg_bytes / rays, "sub_chunks_per_ray": int(c["sub_chunks_entered"]) / rays,
            "nodes_per_ray": int(c["nodes_fetched"]) / rays, "hit_fraction": int(c["hits"]) / rays}


def cpu_baseline(packed, cam, width, height, stride, min_core_seconds=12.0):
    """the CPU restatement of the reference traversal (oracle/blok_oracle.cpp, per-visit counters
    compiled out, -O3 -march=native -ffp-contract=off, built on this machine) over the same frame, rows split over all
    usable host cores; whole frames until >= ~12 core-seconds.  Baseline only."""
    from tests import oracle_ffi as O
    threads = usable_cores()
    lattice = O.NativeLattice(packed.nodes, packed.sub_chunks)
    lattice.trace_primary(cam, width, height, stride=8, threads=threads, want_hits=False)       # warm caches
    frames, rays = 0, 0
    t0 = time.perf_counter()
    while True:
        _, c = lattice.trace_primary(cam, width, height, stride=stride, threads=threads, want_hits=False)
        frames += 1
        rays += int(c["rays"])
        dt = time.perf_counter() - t0
        if (dt * threads >= min_core_seconds and dt >= 1.0) or frames >= 256:
            break
    t1 = time.perf_counter()                      # one-thread rate on a strided sample of the same frame
    _, c1 = lattice.trace_primary(cam, width, height, stride=4, threads=1, want_hits=False)
    single = int(c1["rays"]) / (time.perf_counter() - t1) / 1e6
    return {
        "value": rays / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port", "single_thread_value": single,
        "build": "g++ -O3 -march=native -ffp-contract=off -DORC_NO_COUNTERS (oracle/Makefile: native), built on this host",
        "sample": f"{frames} pass(es) over every {stride}th pixel in x and y of the {width}x{height} frame "
                  f"({rays} rays, {dt:.2f} s wall on {threads} threads = {dt * threads:.0f} core-seconds); "
                  f"hardware threads on the host: {os.cpu_count()}, usable by this process (affinity and cgroup CPU quota): {threads}",
    }


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(spawn_ranks(args.gpus))
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    args.gpus = world_size
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a gfx950 GPU (no CPU fallback exists)")
    device_index = local_rank % torch.cuda.device_count()      # a launcher may expose one device per rank
    torch.cuda.set_device(device_index)
    dist = None
    if world_size > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime
        limit = datetime.timedelta(seconds=300)      # a collective that never completes ends the run with an error instead of hanging the node
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index), timeout=limit)
        else:
            dist.init_process_group(args.backend, timeout=limit)

    from blok_amd import world as W
    from blok_amd.tracer import HipTracer

    W_, H_ = args.width, args.height
    packed = build_world_shared(args.n, args.seed, dist, rank)
    cam = W.scene_camera(args.n, args.pose, W_, H_, args.seed)
    tracer = HipTracer(W_, H_, device=device_index).init()
    if args.dense_dda:
        tracer.set_dense_dda(True)
        stats = tracer.add_dense(W.scene_dense(args.n, args.seed), (0, 0, 0), W.scene_materials(args.seed))
        stats.n_ref_nodes, stats.n_sub_chunks = len(packed.nodes), len(packed.sub_chunks)
    else:
        stats = tracer.add_world(packed)                  # world resident in HBM from here on
    tracer.set_beam(args.beam)
    tracer.set_fused(args.fused)
    tracer.set_list_classes(bool(args.list_classes))
    tracer.set_tile_ordering(args.tile_ordering)
    tracer.set_moving_order(bool(args.moving_order))
    if args.beam_budget:
        tracer.set_beam_budget(args.beam_budget)

    if args.frames_in_flight <= 0:
        args.frames_in_flight = 3 if world_size <= 2 else 4
    from blok_amd.multi_gpu import FramePipeline, HipBackend
    stream = torch.cuda.current_stream()
    pipe = FramePipeline(HipBackend(tracer, cam), W_, H_, rank, world_size, dist, tile=args.tile, depth=args.frames_in_flight,
                         sparse=args.sparse_gather, batch=args.gather_frames or min(8, world_size))

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # --orbit: the camera of frame k stands on a circle around the vertical axis through the world's centre, orbit degrees
    # further per frame, there and back (ping-pong over 64 positions, so consecutive frames always differ by one step)
    def orbit_arc(step_deg):
        n_f = float(args.n)
        centre = np.array([0.5 * n_f, 0.25 * n_f, 0.5 * n_f]); start = np.array([-0.35 * n_f, 0.85 * n_f, -0.35 * n_f]) - centre
        arc = []
        for i in range(64):
            a = np.radians(step_deg * i)
            p = centre + np.array([start[0] * np.cos(a) - start[2] * np.sin(a), start[1], start[0] * np.sin(a) + start[2] * np.cos(a)])
            arc.append(W.camera_look_at(tuple(float(v) for v in p), tuple(float(v) for v in centre), 60.0, W_, H_))
        return arc + arc[-2:0:-1]

    orbit_cams = orbit_arc(args.orbit) if args.orbit > 0.0 else None
    frame_no = [0]

    def next_frame():
        if orbit_cams is not None:
            pipe.backend.cam = orbit_cams[frame_no[0] % len(orbit_cams)]
            frame_no[0] += 1
        pipe.step()

    for _ in range(args.settle + args.warmup):
        next_frame()
    pipe.flush()
    fence()
    # HIP events on the streams the kernels are launched on: one at the head of slot 0's stream, one at the tail of
    # every slot's stream; the device time of the region is the longest head->tail span
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in pipe.streams]
    t0 = time.perf_counter()
    ev0.record(pipe.streams[0])
    for _ in range(args.steps):
        next_frame()
    pipe.flush()                                 # every frame's trace, gather and un-permute are inside the timed region
    for e, st in zip(ev1, pipe.streams):
        e.record(st)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    fence()
    device_ms = max(ev0.elapsed_time(e) for e in ev1)
    if dist is not None:
        t = torch.tensor([elapsed, device_ms], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, device_ms = float(t[0]), float(t[1])

    # the frame's launch running ALONE: HIP events around single launches on one stream, one at a time (what a rocprofv3
    # kernel trace of `--frames-in-flight 1` shows; profiles/README.md).  One launch (frame_kernel) per frame in the
    # one-launch form, the beam_kernel + trace_kernel pair otherwise.
    def solitary_ms(backend, reps, cams=None):
        tracer.set_timing(True)
        ms = []
        cams = cams if cams is not None else orbit_cams
        for k in range(-2, reps):           # two unmeasured launches first
            if cams is not None:
                backend.cam = cams[k % len(cams)]
            if world_size == 1:
                backend.trace_full(pipe.hits, pipe._frame[0], stream.cuda_stream)
            else:
                backend.trace_tiles(args.tile, rank, world_size, pipe.hits, pipe.rgba[0], stream.cuda_stream)
            torch.cuda.synchronize()
            if k >= 0:
                ms.append(tracer.last_kernel_ms())
        tracer.set_timing(False)
        return float(np.mean(ms))

    gave_up = tracer.frame_queue_stalls()        # joint launches: walk waves that stopped waiting for their tile's search (they start at the ray origin: same frame, more work)
    kernel_ms_avg = solitary_ms(pipe.backend, min(args.steps, 20))
    # ... and for a camera in motion (1 degree per frame around the world's centre): no order from earlier frames applies there
    kernel_ms_moving = None
    if world_size == 1 and args.orbit == 0.0 and not args.dense_dda:
        moving = HipBackend(tracer, cam)
        kernel_ms_moving = solitary_ms(moving, 16, orbit_arc(1.0))
    # ... for a caller that alternates between two fixed views (stereo eyes, a cut back and forth): each view has its own cached order from its
    # second appearance on (round 4) — and for the first frame of a view never seen before (no scheduling state at all: row-major order)
    kernel_ms_alternating = kernel_ms_cold = None
    if world_size == 1 and args.orbit == 0.0 and not args.dense_dda:
        two = [cam, orbit_arc(6.0)[1]]
        alt = HipBackend(tracer, cam)
        solitary_ms(alt, 10, two)                                  # each view a few times, its sort, its adoption: untimed, like --settle for the one view
        kernel_ms_alternating = solitary_ms(alt, 16, two)
        tracer.set_timing(True)
        cold = []
        for k in range(4):
            alt.cam = orbit_arc(50.0 + 17.0 * k)[1]
            alt.trace_full(pipe.hits, pipe._frame[0], stream.cuda_stream); torch.cuda.synchronize()
            cold.append(tracer.last_kernel_ms())
        tracer.set_timing(False)
        kernel_ms_cold = float(np.mean(cold))
    local_hits = (pipe.hits[:, 3] >> 24).sum()
    # N > 1: what the process group actually was — every rank reports itself through the collective backend (so that the line cannot claim N
    # ranks unless N ranks answered): its device, its share of the tiles, its tile launch alone on its device
    ranks_info = backend_info = None
    if dist is not None:
        props = torch.cuda.get_device_properties(torch.cuda.current_device())
        mine = {"rank": rank, "local_rank": int(os.environ.get("LOCAL_RANK", "0")), "device": torch.cuda.current_device(), "device_name": props.name,
                "device_uuid": str(getattr(props, "uuid", "")), "pci_bus_id": getattr(props, "pci_bus_id", None),
                "tiles": int(tracer.tiles_for_rank(args.tile, rank, world_size)), "kernel_ms_alone": kernel_ms_avg, "order_in_use": int(tracer.last_order_use()[0]),
                "pid": os.getpid()}
        gathered = [None] * world_size
        dist.all_gather_object(gathered, mine)
        ranks_info = gathered
        nccl = None
        try:
            nccl = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:
            pass
        backend_info = {"backend": dist.get_backend(), "world_size_reported": dist.get_world_size(), "collective_library": (f"RCCL (torch.cuda.nccl.version) {nccl}" if dist.get_backend() == "nccl" else dist.get_backend()),
                        "hip": torch.version.hip, "distinct_devices": len({(r["device_uuid"], r["pci_bus_id"], r["device"]) for r in gathered})}
        dist.all_reduce(local_hits)
    hits = int(local_hits.item())
    sky = 0xFF000000 | (230 << 16) | (200 << 8) | 160
    lit_pixels = int((pipe.frame_rgba != (sky - (1 << 32))).sum().item()) if rank == 0 else 0

    # the other camera poses of SURVEY.md §8(d), same world and frame: a short pipelined run and the solitary launch each
    poses = {}
    if world_size == 1 and not args.no_poses:
        for pose in (0, 1, 2):
            pcam = W.scene_camera(args.n, pose, W_, H_, args.seed)
            pb = HipBackend(tracer, pcam)
            pipe.backend = pb
            for _ in range(24):
                pipe.step()
            pipe.flush(); torch.cuda.synchronize()
            k = max(10, min(args.steps, 60))
            t1 = time.perf_counter()
            for _ in range(k):
                pipe.step()
            pipe.flush(); torch.cuda.synchronize()
            dt = time.perf_counter() - t1
            entry = {"Mrays_per_s": W_ * H_ * k / dt / 1e6, "ms_per_frame": dt / k * 1e3, "ms_per_frame_alone": solitary_ms(pb, 10),
                     "hits_per_frame": int((pipe.hits[:, 3] >> 24).sum().item())}
            if not args.no_cpu_baseline:
                c = oracle_counters(packed, pcam, W_, H_)
                # nominal: the reference's per-visit byte count over the launch's duration — not a bound (pose B exceeds 1: the reference
                # fetches 22 nodes per ray there, this structure a few)
                entry.update({"hit_fraction": c["hit_fraction"], "algorithmic_bytes_per_ray": c["bytes_per_ray"],
                              "frac_nominal_alone": c["bytes_per_ray"] * W_ * H_ / (entry["ms_per_frame_alone"] * 1e-3) / 1e9 / HBM_PEAK_GBS})
            poses["ABC"[pose]] = entry
        pipe.backend = HipBackend(tracer, cam)

    # side measurement (not the headline value): BASELINE.json configs[4], the reference's sample/bounce loop at
    # 64 spp, 2 bounces, over the same world and frame — two frames, HIP events around the kernel
    paths = None
    if world_size == 1 and not args.no_paths:
        color = torch.empty((H_ * W_, 4), dtype=torch.float32, device="cuda")
        tracer.set_timing(True)
        ms = []
        for f in range(3):
            tracer.trace_paths_device(cam, color.data_ptr(), spp=64, max_bounces=2, frame_index=f, stream=stream.cuda_stream)
            torch.cuda.synchronize()
            ms.append(tracer.last_kernel_ms())
        tracer.set_timing(False)
        path_ms = float(np.mean(ms[1:]))
        paths = {"config": "3840x2160 x 64 spp, 2 bounces + sun shadow ray (raygen.rgen loop)", "ms_per_frame": path_ms,
                 "Gpaths_per_s": W_ * H_ * 64 / (path_ms * 1e-3) / 1e9}
        del color
        if rank == 0 and not args.no_cpu_baseline:
            # algorithmic bytes of the path loop, by the oracle's counters on every 8th pixel in x and y at the same 64 spp: per ray
            # segment (radiance or shadow) 48 B per sub-chunk descriptor entered + 16 B per node fetched, 32 B of material per
            # radiance hit, and per pixel the 48-B G-buffer of raygen.rgen:392-413
            from tests import oracle_ffi as O
            lat = O.Lattice(packed.nodes, packed.sub_chunks)
            _, pc = O.render_paths(lat, packed.materials, cam, W_, H_, spp=64, max_bounces=2, frame_index=1, stride=8, threads=usable_cores())
            px = ((W_ + 7) // 8) * ((H_ + 7) // 8)
            seg_bytes = 48 * int(pc["sub_chunks_entered"]) + 16 * int(pc["nodes_fetched"]) + 32 * int(pc["hits"])
            per_pixel = seg_bytes / px + 48.0
            achieved = per_pixel * W_ * H_ / (path_ms * 1e-3) / 1e9
            paths["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                                 "kernel": "path_kernel (behind its beam_kernel; every wave tile also behind its own 8x8 beam), one launch per frame, alone", "kernel_ms": path_ms,
                                 "algorithmic_bytes_per_pixel": per_pixel, "ray_segments_per_pixel": int(pc["rays"]) / px,
                                 "Gsegments_per_s": int(pc["rays"]) / px * W_ * H_ / (path_ms * 1e-3) / 1e9,
                                 "sample": f"oracle counters on every 8th pixel in x and y ({px} pixels, 64 spp)"}

    if rank == 0:
        rays_per_step = W_ * H_
        value = rays_per_step * args.steps / elapsed / 1e6
        if args.dense_dda:
            launch = "dense_kernel per frame (id grid in 8^3 tiles, tile bits in LDS, two-level DDA)"
        elif not args.beam:
            launch = "trace_kernel per frame"
        elif args.fused == 1:
            launch = "one persistent launch per frame (frame_kernel: beam pre-pass + walk, work queues)"
        elif args.fused == 2:
            launch = "joint_kernel per frame (searches and walk waves in one grid)"
        elif args.fused in (3, 4, 5):
            launch = {3: "joint_kernel per frame when the launch has the device to itself (searches and walk waves in one grid; walk waves for the live prefix of the longest-first order), "
                         "beam_kernel + trace_kernel over the same prefix with frames in flight",
                      4: "list_joint_kernel per frame (searches and list-fed walk waves in one grid; walk waves only for the wave tiles the frame's own searches found live)",
                      5: "beam_kernel + list_walk_kernel per frame (walk waves only for the wave tiles the frame's own searches found live)"}[args.fused]
        else:
            launch = "beam_kernel + trace_kernel per frame"
        out = {
            "metric": "Mrays/sec, primary first-hit rays at 4K over a 1024^3 SVO",
            "value": value, "unit": "Mrays/s", "n_gpus": world_size, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.n}^3 synthetic SVO (G(N,seed) terrain shell + 64 spheres, "
                                   f"{stats.n_voxels} voxels, {stats.n_ref_nodes} reference SvoNodes, "
                                   f"{stats.n_sub_chunks} sub-chunks), {W_}x{H_} primary rays, camera pose "
                                   f"{'ABC'[args.pose]}, first-hit records 16 B/ray",
                       "parallelism": f"single GPU, {launch}, {args.frames_in_flight} frame(s) in flight" if world_size == 1 else f"{args.tile}x{args.tile} screen tiles round-robin over {world_size} GPUs, {'RCCL' if args.backend == 'nccl' else args.backend} gather of {'the live tiles as 16-bit material/face codes' if pipe.codes else 'the live RGBA8 tiles' if args.sparse_gather else 'RGBA8 tiles'} to rank 0, {pipe.batch} frame(s) per launch pair and exchange, {args.frames_in_flight} batches in flight",
                       "hits_per_frame": hits, "framebuffer_pixels_hit": lit_pixels,
                       "outputs": "16-B first-hit records (kept on the tracing GPU) + RGBA8 framebuffer on rank 0",
                       "tile_records_gathered_per_frame_and_rank": (pipe.records_gathered / max(1, pipe.frames_done)) if world_size > 1 and args.sparse_gather else None,
                       "tiles_per_rank": pipe.per_rank if world_size > 1 else None,
                       "ranks": ranks_info, "process_group": backend_info,
                       "camera_orbit_deg_per_frame": args.orbit,
                       "frames_in_flight": args.frames_in_flight, "settle_frames": args.settle, "walk_waves_that_gave_up_waiting": gave_up, "device_ms_per_step": device_ms / args.steps, "kernel_ms_alone": kernel_ms_avg, "kernel_ms_alone_moving": kernel_ms_moving, "beam_tile": args.beam,
                       "poses": poses, "also_measured_paths": paths},
        }
        alg = None
        if not args.no_cpu_baseline:
            alg = oracle_counters(packed, cam, W_, H_, 1 if world_size == 1 else 4)
            if world_size == 1:
                out["cpu_baseline"] = cpu_baseline(packed, cam, W_, H_, args.cpu_stride)
        rays_per_launch = rays_per_step / world_size
        if alg is not None:
            # frac: algorithmic bytes of one launch / duration of that launch running alone.  The throughput-derived figure
            # (device time of the timed region / steps, launches of several frames overlapping) is frac_overlapped.
            achieved = alg["bytes_per_ray"] * rays_per_launch / (kernel_ms_avg * 1e-3) / 1e9
            overlapped_ms = device_ms / args.steps
            achieved_overlapped = alg["bytes_per_ray"] * rays_per_launch / (overlapped_ms * 1e-3) / 1e9
            # counters of the frame's kernel from the committed rocprofv3 summary (profiles/<tag>_summary.json, scripts/r03/summarize.py): the
            # run that measured them and its commit are named beside the figures; nothing is re-measured here
            traffic, physical = None, None
            summaries = sorted((ROOT / "profiles").glob("r0*_summary.json"))
            if summaries and world_size == 1 and (args.n, W_, H_, args.pose, args.orbit) == (1024, 3840, 2160, 0, 0.0) and not args.dense_dda:
                prof = json.loads(summaries[-1].read_text())
                k = prof["kernels"].get("joint_kernel") or next(iter(prof["kernels"].values()), None)
                if k:
                    traffic = k.get("hbm_bytes_per_launch")
                    physical = {"from": f"profiles/{summaries[-1].name} (tag {prof['tag']}, commit {prof['commit']}; rocprofv3 --pmc, separate passes, one frame in flight)",
                                "kernel": k["kernel"].split("(anonymous namespace)::")[-1][:60], "kernel_us_in_that_run": k.get("duration_us_unprofiled"),
                                "hbm_bytes_per_launch": traffic, "hbm_bytes_per_launch_upper": k.get("hbm_bytes_per_launch_upper"),
                                "hbm_traffic_frac": (traffic / (k["duration_us_unprofiled"] * 1e-6) / 1e9 / HBM_PEAK_GBS) if traffic and k.get("duration_us_unprofiled") else None,
                                "valu_wave_instructions_per_launch": k.get("valu_wave_instructions_per_launch"),
                                "valu_issue_interval_cycles_per_simd": k.get("valu_issue_interval_cycles_per_simd"), "valu_issue_frac": k.get("valu_issue_frac"),
                                "lane_utilisation": k.get("lane_utilisation"),
                                "note": "what bounds the kernel is VALU issue under divergence, not memory: hbm_traffic_frac is the counted HBM bytes over the launch's duration against 8 TB/s; "
                                        "valu_issue_frac = the issue interval the loop's instruction mix needs alone on a SIMD / the measured interval; lane_utilisation = SQ_THREAD_CYCLES_VALU / (64 SQ_ACTIVE_INST_VALU)"}
            out["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                               "frac_moving": (alg["bytes_per_ray"] * rays_per_launch / (kernel_ms_moving * 1e-3) / 1e9 / HBM_PEAK_GBS) if kernel_ms_moving else None,
                               "kernel_ms_moving": kernel_ms_moving,
                               "moving": ("the same launch alone with the camera turning 1 degree per frame around the world's centre, every frame a view never seen before (pose-A bytes per ray); "
                                          + ("walked in the previous frame's dilated order carried over by a whole-tile shift; the three 4-5 us launches that keep that order follow each frame on its stream and are "
                                             "NOT in kernel_ms_moving (wall-clock period of synchronised frames with them: 257 us against 274 us in row-major order, profiles/r03_moving_order_solitary_frames.txt)"
                                             if args.moving_order else "row-major order")) if kernel_ms_moving else None,
                               "frac_alternating": (alg["bytes_per_ray"] * rays_per_launch / (kernel_ms_alternating * 1e-3) / 1e9 / HBM_PEAK_GBS) if kernel_ms_alternating else None,
                               "kernel_ms_alternating": kernel_ms_alternating,
                               "frac_cold": (alg["bytes_per_ray"] * rays_per_launch / (kernel_ms_cold * 1e-3) / 1e9 / HBM_PEAK_GBS) if kernel_ms_cold else None,
                               "kernel_ms_cold": kernel_ms_cold,
                               "alternating_and_cold": ("the same launch alone (pose-A bytes per ray) for a caller alternating between two fixed views 6 degrees apart, each walking in its own cached "
                                                        "order (12 untimed frames first); and for the FIRST frame of four views never seen before, 17 degrees apart: no order, no live prefix, "
                                                        "row-major joint launch — the figure that uses no scheduling state at all") if kernel_ms_alternating else None,
                               "physical": physical,
                               "kernel": launch, "kernel_ms": kernel_ms_avg,
                               "timing": f"HIP events around single launches, one at a time on an otherwise idle chip; {args.settle} settle + {args.warmup} warmup frames before the timed region, 2 unmeasured launches before the single ones",
                               "frac_overlapped": achieved_overlapped / HBM_PEAK_GBS, "achieved_overlapped": achieved_overlapped,
                               "kernel_ms_overlapped": overlapped_ms, "frames_in_flight": args.frames_in_flight,
                               "beam_tile": args.beam, "algorithmic_bytes_per_ray": alg["bytes_per_ray"],
                               "rays_per_launch": rays_per_launch,
                               "sub_chunks_per_ray": alg["sub_chunks_per_ray"], "nodes_per_ray": alg["nodes_per_ray"],
                               "hit_fraction": alg["hit_fraction"]}
        print(json.dumps(out), flush=True)
    tracer.shutdown()
    if dist is not None:
        # rank 0 has been on the CPU (oracle byte counts) while the others got here at once: everybody leaves together, so no rank tears
        # the group down under a peer that still owes a collective
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
