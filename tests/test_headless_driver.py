"""The C++20 host side (include/blok/hip_tracer.hpp) through the App-shaped headless driver."""
import subprocess

import pytest

from blok_amd import build as b


def test_driver_builds_and_fails_loudly_without_gpu():
    import torch
    exe = b.build_tools()
    assert exe.exists()
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    proc = subprocess.run([str(exe), "--n", "64", "--frames", "1"], capture_output=True, text=True)
    assert proc.returncode == 1 and "[FATAL]" in proc.stderr and "HipTracer::init" in proc.stderr


@pytest.mark.gpu
def test_driver_renders_a_frame(tmp_path):
    exe = b.build_tools()
    out = tmp_path / "frame.ppm"
    proc = subprocess.run([str(exe), "--n", "64", "--size", "320x200", "--frames", "2", "--out", str(out)],
                          capture_output=True, text=True, timeout=300)
    assert proc.returncode == 0, proc.stderr
    assert "10082 voxels" in proc.stdout and "frame 1:" in proc.stdout
    data = out.read_bytes()
    assert data.startswith(b"P6\n320 200\n255\n") and len(data) == len(b"P6\n320 200\n255\n") + 320 * 200 * 3
    body = data[len(b"P6\n320 200\n255\n"):]
    assert len(set(body[i:i + 3] for i in range(0, len(body), 3))) > 20      # terrain colours, not a flat image


@pytest.mark.gpu
def test_driver_full_ray_tracing_path(tmp_path):
    """--rt: the C++ mirror's drawFrameRT (path trace -> denoise -> TAA -> tonemap -> sharpen) over several moving frames."""
    exe = b.build_tools()
    out = tmp_path / "rt.ppm"
    proc = subprocess.run([str(exe), "--n", "64", "--size", "320x200", "--frames", "4", "--rt", "--spp", "2", "--out", str(out)],
                          capture_output=True, text=True, timeout=300)
    assert proc.returncode == 0, proc.stderr
    assert "frame 3:" in proc.stdout and "denoise" in proc.stdout
    data = out.read_bytes()
    head = b"P6\n320 200\n255\n"
    assert data.startswith(head) and len(data) == len(head) + 320 * 200 * 3
    body = data[len(head):]
    assert len(set(body[i:i + 3] for i in range(0, len(body), 3))) > 200     # shaded, filtered colours


@pytest.mark.gpu
def test_driver_multi_device_tile_partition(tmp_path):
    """blok::HipMultiTracer (one process, several ranks): three ranks rehearsed on device 0 (peer-copy transport; RCCL refuses a
    repeated device), with the sparse-pull and the dense exchange — the driver itself compares the partitioned frame with the
    single-device one."""
    exe = b.build_tools()
    out = tmp_path / "multi.ppm"
    for extra, exchange in (([], "sparse-pull"), (["--dense-exchange"], "dense")):
        proc = subprocess.run([str(exe), "--n", "64", "--size", "328x200", "--frames", "3", "--devices", "0,0,0", "--out", str(out)] + extra,
                              capture_output=True, text=True, timeout=300)
        assert proc.returncode == 0, proc.stderr + proc.stdout
        assert f"3 ranks, exchange {exchange}, transport peer-copy" in proc.stdout and "0 pixels differ" in proc.stdout
