"""bench.py's command line: the help text renders (argparse expands % in help strings), the defaults are the contract's, and — on a
GPU — every combination of the side-measurement switches still prints its one JSON line (a stray variable once made --no-poses fail)."""
import json
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def test_help_and_defaults():
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "--gpus" in out.stdout and "--steps" in out.stdout and "--warmup" in out.stdout, out.stderr
    sys.path.insert(0, str(ROOT))
    import importlib
    bench = importlib.import_module("bench")
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        a = bench.parse()
    finally:
        sys.argv = argv
    assert (a.gpus, a.n, a.width, a.height, a.pose, a.settle, a.orbit) == (1, 1024, 3840, 2160, 0, 32, 0.0)
    assert a.steps > 0 and a.warmup >= 0 and a.fused == 3


@pytest.mark.gpu
@pytest.mark.parametrize("flags", [[], ["--no-poses"], ["--no-paths"], ["--no-cpu-baseline"], ["--no-poses", "--no-paths"], ["--no-poses", "--no-cpu-baseline"],
                                   ["--no-paths", "--no-cpu-baseline", "--orbit", "1"], ["--fused", "4", "--no-paths"], ["--fused", "0", "--tile-ordering", "0", "--no-paths", "--no-poses"]])
def test_flag_combinations_print_one_line(flags):
    cmd = [sys.executable, str(ROOT / "bench.py"), "--n", "64", "--width", "512", "--height", "256", "--steps", "3", "--warmup", "1", "--cpu-stride", "4", *flags]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["unit"] == "Mrays/s" and d["n_gpus"] == 1 and d["steps"] == 3 and d["value"] > 0
    assert ("cpu_baseline" in d) == ("--no-cpu-baseline" not in flags)
    assert ("roofline" in d) == ("--no-cpu-baseline" not in flags)
    assert (d["config"]["also_measured_paths"] is None) == ("--no-paths" in flags)
    assert (d["config"]["poses"] == {}) == ("--no-poses" in flags)
