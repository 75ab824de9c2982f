"""The launch decisions of the gfx950 backend as a table (blok_amd/csrc/hip/launch_policy.h: plan_launch, plan_order), on the host:
which kernels a frame is launched as — static / creeping / jumping camera x device busy / idle x launch form — how the list forms size
their walk grid, and when a camera at rest is measured, sorted for and walked in its order.  No GPU."""
import ctypes as C
import os
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
SRC = ROOT / "tests" / "host_harness" / "policy_shim.cpp"
HDR = ROOT / "blok_amd" / "csrc" / "hip" / "launch_policy.h"
LIB = ROOT / "tests" / "host_harness" / "libpolicy_shim.so"

WALK, TWO, QUEUES, JOINT, LIST_JOINT, LIST_TWO = range(6)


@pytest.fixture(scope="module")
def policy():
    if not LIB.exists() or LIB.stat().st_mtime < max(SRC.stat().st_mtime, HDR.stat().st_mtime):
        subprocess.run(["g++", "-O1", "-std=c++20", "-fPIC", "-Wall", "-Wextra", "-Werror", f"-I{HDR.parent}", "-shared", "-o", os.fspath(LIB), os.fspath(SRC)], check=True)
    L = C.CDLL(os.fspath(LIB))
    L.policy_plan_launch.argtypes = [C.c_int] * 4 + [C.c_uint, C.c_int, C.POINTER(C.c_uint), C.POINTER(C.c_uint), C.POINTER(C.c_uint)]
    L.policy_plan_order.argtypes = [C.c_int] * 5 + [C.c_uint] * 4 + [C.c_int] * 5 + [C.POINTER(C.c_uint)] * 2
    L.policy_plan_shift.argtypes = [C.POINTER(C.c_float)] * 2 + [C.c_float] * 2 + [C.c_uint] * 5 + [C.POINTER(C.c_uint)] * 2 + [C.POINTER(C.c_float)]
    L.policy_plan_dilation.argtypes = [C.c_int, C.c_float]
    L.policy_plan_dilation.restype = C.c_uint
    return L


def plan(L, form, busy=False, beam=True, one_wave=True, tiles=129600, hint=None):
    walkers = C.c_uint(0)
    per = (C.c_uint * 4)()
    h = (C.c_uint * 4)(*(hint or (0, 0, 0, 0)))
    r = L.policy_plan_launch(form, int(beam), int(one_wave), int(busy), tiles, int(hint is not None), h, C.byref(walkers), per)
    return r & 255, bool(r & 256), walkers.value, list(per)


def test_launch_kinds(policy):
    # no pre-pass: the walk alone, whatever was asked for
    for form in range(6):
        assert plan(policy, form, beam=False)[0] == WALK
    # explicit forms are what they say, busy or not
    for busy in (False, True):
        assert plan(policy, 0, busy)[0] == TWO
        assert plan(policy, 1, busy)[0] == QUEUES
        assert plan(policy, 2, busy)[0] == JOINT
        assert plan(policy, 4, busy)[0] == LIST_JOINT
        assert plan(policy, 5, busy)[0] == LIST_TWO
    # automatic: joint only with the device to itself — two contexts, or two streams, can never have two joint launches in flight
    assert plan(policy, 3, busy=False)[0] == JOINT
    assert plan(policy, 3, busy=True)[0] == TWO
    # a build with several waves per workgroup has neither the joint prefix nor the lists
    assert plan(policy, 3, one_wave=False)[0] == TWO
    assert plan(policy, 4, one_wave=False)[0] == JOINT and plan(policy, 5, one_wave=False)[0] == TWO
    # a list entry names its wave tile in 21 bits
    assert plan(policy, 4, tiles=(1 << 21))[0] == JOINT and plan(policy, 4, tiles=(1 << 21) - 1)[0] == LIST_JOINT


def test_prefix_only_where_search_waves_can_walk(policy):
    assert plan(policy, 2)[1] and plan(policy, 3)[1] and plan(policy, 3, busy=True)[1]
    assert not plan(policy, 0)[1]                       # the explicit two-launch form keeps one walk wave per wave tile
    assert not plan(policy, 4)[1] and not plan(policy, 5)[1] and not plan(policy, 1)[1]
    assert not plan(policy, 3, one_wave=False)[1]


def test_list_walk_grid_is_a_bounded_hint(policy):
    tiles = 129600
    kind, _, walkers, per = plan(policy, 4, tiles=tiles)
    assert per == [16, tiles // 8, 16, 16] and walkers == 8 * sum(per)      # no previous launch: every tile is of the unknown class
    kind, _, walkers, per = plan(policy, 4, tiles=tiles, hint=(4000, 900, 300, 40))
    assert per == [4000 + 500 + 16, 900 + 112 + 16, 300 + 37 + 16, 40 + 5 + 16] and walkers == 8 * sum(per)
    # never beyond one workgroup per wave tile of the segment, never none
    kind, _, walkers, per = plan(policy, 5, tiles=800, hint=(10 ** 6, 0, 0, 0))
    assert per == [100, 16, 16, 16]
    assert plan(policy, 5, tiles=8, hint=(0, 0, 0, 0))[3] == [1, 1, 1, 1]
    assert all(w % 8 == 0 for w in (plan(policy, 4, tiles=t, hint=(t // 9, 3, 2, 1))[2] for t in (4096, 129600, 129601)))


def order(L, have=False, near_order=False, near_last=False, pending=False, still=0, since=0, interval=8, now=8, enabled=True,
          moving=False, alone=True, alone_before=True, dilated=False, shift_ok=False, flags=False):
    s, n = C.c_uint(0), C.c_uint(0)
    r = L.policy_plan_order(int(enabled), int(have), int(near_order), int(near_last), int(pending), still, since, interval, now,
                            int(moving), int(alone), int(alone_before), int(dilated), int(shift_ok), C.byref(s), C.byref(n))
    if flags:
        return {"use": bool(r & 1), "measure": bool(r & 2), "sort": bool(r & 4), "shifted": bool(r & 8), "dilate": bool(r & 16)}
    return bool(r & 1), bool(r & 2), bool(r & 4), s.value, n.value


def test_order_of_a_camera_at_rest(policy):
    for moving in (False, True):                         # the order for a camera in motion changes nothing for one at rest
        assert order(policy, enabled=False, have=True, near_order=True, near_last=True, moving=moving) == (False, False, False, 0, 8)
        # second frame at rest: measured, and sorted for at once (there is no order for this view)
        assert order(policy, near_last=True, moving=moving) == (False, True, True, 1, 8)
        # ... but never two sorts in flight
        assert order(policy, near_last=True, still=1, pending=True, moving=moving) == (False, True, False, 2, 8)
        # adopted: used, measured; re-sorted only when it is `interval` launches old, and then ever less often
        assert order(policy, have=True, near_order=True, near_last=True, still=5, since=3, moving=moving) == (True, True, False, 6, 8)
        assert order(policy, have=True, near_order=True, near_last=True, still=9, since=7, moving=moving) == (True, True, True, 10, 16)
        assert order(policy, have=True, near_order=True, near_last=True, still=40, since=15, now=16, moving=moving) == (True, True, True, 41, 32)
        assert order(policy, have=True, near_order=True, near_last=True, still=40, since=14, now=16, moving=moving) == (True, True, False, 41, 16)
        assert order(policy, have=True, near_order=True, near_last=True, still=99, since=63, now=64, moving=moving) == (True, True, True, 100, 64)
        assert not order(policy, have=True, near_order=True, near_last=True, still=9, since=7, moving=moving, flags=True)["dilate"]
    # first frame of a view: nothing to use; without the moving order nothing is measured either (the camera has not rested yet)
    assert order(policy) == (False, False, False, 0, 8)


def test_order_of_a_camera_in_motion_without_the_carried_order(policy):
    # creeping away from the order's view but at rest from frame to frame: the old order is dropped, the new view is sorted for at once
    assert order(policy, have=True, near_order=False, near_last=True, still=3, since=2, now=32) == (False, True, True, 4, 8)
    # moving from frame to frame: natural order, not measured, not sorted for — whatever order exists
    assert order(policy, have=True, near_order=False, near_last=False, still=7, since=20) == (False, False, False, 0, 8)
    # a jump back into the order's view: used at once (it was measured here), measured again from the next frame on
    assert order(policy, have=True, near_order=True, near_last=False, still=0, since=3) == (True, False, False, 0, 8)
    # interval 0 = measure but never sort
    assert order(policy, near_last=True, interval=0, now=0)[2] is False


def test_carried_order_of_a_camera_in_motion(policy):
    # a frame alone on the device, camera moving: measured, and a dilated sort follows it — with or without an order to use
    p = order(policy, moving=True, alone=True, flags=True)
    assert p == {"use": False, "measure": True, "sort": True, "shifted": False, "dilate": True}
    # the next frame walks in that order, carried over, and leaves the next one
    p = order(policy, have=True, dilated=True, shift_ok=True, moving=True, alone=True, flags=True)
    assert p == {"use": True, "measure": True, "sort": True, "shifted": True, "dilate": True}
    # the shift does not reach (a jump, too much parallax): row-major order, but the frame still leaves an order behind
    p = order(policy, have=True, dilated=True, shift_ok=False, moving=True, alone=True, flags=True)
    assert p == {"use": False, "measure": True, "sort": True, "shifted": False, "dilate": True}
    # never two sorts in flight
    assert not order(policy, have=True, dilated=True, shift_ok=True, moving=True, alone=True, pending=True, flags=True)["sort"]
    # a launch that merely finds the device idle between the frames of a pipeline (the previous one did not): as with frames in flight
    p = order(policy, have=True, dilated=True, shift_ok=True, moving=True, alone=True, alone_before=False, flags=True)
    assert p == {"use": False, "measure": False, "sort": False, "shifted": False, "dilate": False}
    # beside frames in flight on other streams: as before — row-major, not measured, no sort
    p = order(policy, have=True, dilated=True, shift_ok=True, moving=True, alone=False, flags=True)
    assert p == {"use": False, "measure": False, "sort": False, "shifted": False, "dilate": False}
    # an order of one view's own clocks is not carried anywhere (undilated, it is no better than row-major a quarter of a degree away)
    assert not order(policy, have=True, dilated=False, near_order=False, shift_ok=True, moving=True, alone=True, flags=True)["use"]
    # the camera stops: the carried order (shift 0) serves until the view's own sort — started at once, undilated — is adopted
    p = order(policy, have=True, dilated=True, shift_ok=True, near_order=True, near_last=True, moving=True, alone=True, flags=True)
    assert p == {"use": True, "measure": True, "sort": True, "shifted": True, "dilate": False}
    # ... and the view's own order takes over
    p = order(policy, have=True, dilated=False, near_order=True, near_last=True, still=4, since=1, moving=True, alone=True, flags=True)
    assert p == {"use": True, "measure": True, "sort": False, "shifted": False, "dilate": False}
    # switched off (blok_hip_set_moving_order(0)): the previous behaviour
    assert order(policy, have=True, dilated=True, shift_ok=True, moving=False, alone=True) == (False, False, False, 0, 8)
    assert order(policy, have=True, dilated=True, shift_ok=True, moving=True, alone=True, interval=0, now=0)[2] is False


def camera(pos, target, fov_deg=60.0, w=3840, h=2160):
    import numpy as np
    pos, target = np.asarray(pos, dtype=np.float64), np.asarray(target, dtype=np.float64)
    fwd = target - pos; fwd /= np.linalg.norm(fwd)
    right = np.cross(fwd, [0.0, 1.0, 0.0]); right /= np.linalg.norm(right)
    up = np.cross(right, fwd)
    return (C.c_float * 14)(*pos, *fwd, *right, *up, float(np.tan(np.radians(fov_deg) / 2)), w / h)


def shift(L, then, now, inv_mean=1 / 1400.0, inv_sigma=0.0, radius=4, w=3840, h=2160):
    sx, sy, res = C.c_uint(0), C.c_uint(0), C.c_float(0)
    ok = L.policy_plan_shift(then, now, inv_mean, inv_sigma, w, h, w // 8, h // 8, radius, C.byref(sx), C.byref(sy), C.byref(res))
    tx, ty = w // 8, h // 8
    shift.measured = bool(ok & 2)
    return bool(ok & 1), (sx.value + tx // 2) % tx - tx // 2, (sy.value + ty // 2) % ty - ty // 2, res.value


def test_shift_between_two_views(policy):
    import numpy as np
    centre = (512.0, 256.0, 512.0)
    a = camera((-358.0, 870.0, -358.0), centre)
    # the same view: no shift, nothing left over
    ok, sx, sy, res = shift(policy, a, a)
    assert ok and (sx, sy) == (0, 0) and res < 1e-3
    # a pure sideways step of the camera, everything at one distance: the image moves the other way by step / distance / (radians per pixel) —
    # a little more towards the corners, whose points at that distance along their rays are nearer the camera plane
    right = np.array([a[6], a[7], a[8]]); pos = np.array([a[0], a[1], a[2]])
    step = 30.0
    b = (C.c_float * 14)(*a); b[0], b[1], b[2] = (pos + right * step).tolist()
    rad_per_px = 2 * np.tan(np.radians(30.0)) * (3840 / 2160) / 3840
    ok, sx, sy, res = shift(policy, a, b, inv_mean=1 / 1400.0)
    assert ok and sy == 0 and sx == -round(step / 1400.0 / rad_per_px / 8) and res < 2.5
    # with a depth range the near field moves more than the shift takes out: that is the residual
    ok, sx2, _, res2 = shift(policy, a, b, inv_mean=1 / 1400.0, inv_sigma=0.25 / 1400.0)
    assert ok and sx2 == sx and res + 0.5 * abs(sx) * 0.8 < res2 < res + 0.5 * abs(sx) * 1.6      # (mean + 2 sigma) / mean = 1.5
    # an orbit by one degree about the centre (what bench.py --orbit 1 does): a shift of a tile or two, a few tiles of parallax
    def orbit(deg):
        c, s0 = np.array(centre), np.array([-358.0, 870.0, -358.0]) - np.array(centre)
        r = np.radians(deg)
        return camera(c + np.array([s0[0] * np.cos(r) - s0[2] * np.sin(r), s0[1], s0[0] * np.sin(r) + s0[2] * np.cos(r)]), centre)
    ok, sx, sy, res = shift(policy, orbit(10), orbit(11), inv_mean=1 / 1300.0, inv_sigma=0.2 / 1300.0)
    assert ok and abs(sx) <= 3 and abs(sy) <= 1 and 0.5 < res < 8.5
    # a turn on the spot by 20 degrees: the shift would be a quarter of the screen and more — not carried
    yaw = np.radians(20.0)
    t = pos + np.array([np.cos(yaw) * a[3] - np.sin(yaw) * a[5], a[4], np.sin(yaw) * a[3] + np.cos(yaw) * a[5]]) * 100.0
    assert not shift(policy, a, camera(pos, t))[0]
    # about-face, a non-finite camera: never — and the residual then says nothing (it read 0, the smallest dilation for the next sort)
    assert not shift(policy, a, camera(pos, pos - (np.array(centre) - pos)))[0] and not shift.measured
    n = (C.c_float * 14)(*a); n[0] = float("nan")
    assert not shift(policy, a, n)[0] and not shift.measured
    # another lens (round 4): a zoom is a scale about the screen's centre, and what it does to the tiles is residual like any other stretch — a
    # slow zoom (60 -> 59.75 degrees: 0.5 % = 1.2 tiles at the 0.8-corner samples of 480 x 270 tiles) is carried, a cut to another lens is not
    ok, sx, sy, res = shift(policy, a, camera((-358.0, 870.0, -358.0), centre, fov_deg=59.75))
    assert ok and shift.measured and (sx, sy) == (0, 0) and 0.5 < res < 2.0, (ok, sx, sy, res)
    ok, sx, sy, res = shift(policy, a, camera((-358.0, 870.0, -358.0), centre, fov_deg=50.0))
    assert not ok and shift.measured and res > 20.0
    # what the shift leaves over sizes the next dilation: rounded up, 2 to 8 tiles, 4 when nothing has been seen yet
    assert [policy.policy_plan_dilation(1, r) for r in (0.0, 1.2, 2.0, 3.1, 7.9, 30.0)] == [2, 2, 2, 4, 8, 8]
    assert policy.policy_plan_dilation(0, 0.0) == 4 and policy.policy_plan_dilation(1, float("nan")) == 4
