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
    L.policy_plan_order.argtypes = [C.c_int] * 5 + [C.c_uint] * 4 + [C.POINTER(C.c_uint)] * 2
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


def order(L, have=False, near_order=False, near_last=False, pending=False, still=0, since=0, interval=8, now=8, enabled=True):
    s, n = C.c_uint(0), C.c_uint(0)
    r = L.policy_plan_order(int(enabled), int(have), int(near_order), int(near_last), int(pending), still, since, interval, now, C.byref(s), C.byref(n))
    return bool(r & 1), bool(r & 2), bool(r & 4), s.value, n.value


def test_order_of_a_camera_at_rest(policy):
    assert order(policy, enabled=False, have=True, near_order=True, near_last=True) == (False, False, False, 0, 8)
    # first frame of a view: nothing to use, nothing measured (the camera has not rested yet)
    assert order(policy) == (False, False, False, 0, 8)
    # second frame at rest: measured, and sorted for at once (there is no order for this view)
    assert order(policy, near_last=True) == (False, True, True, 1, 8)
    # ... but never two sorts in flight
    assert order(policy, near_last=True, still=1, pending=True) == (False, True, False, 2, 8)
    # adopted: used, measured; re-sorted only when it is `interval` launches old, and then ever less often
    assert order(policy, have=True, near_order=True, near_last=True, still=5, since=3) == (True, True, False, 6, 8)
    assert order(policy, have=True, near_order=True, near_last=True, still=9, since=7) == (True, True, True, 10, 16)
    assert order(policy, have=True, near_order=True, near_last=True, still=40, since=15, now=16) == (True, True, True, 41, 32)
    assert order(policy, have=True, near_order=True, near_last=True, still=40, since=14, now=16) == (True, True, False, 41, 16)
    assert order(policy, have=True, near_order=True, near_last=True, still=99, since=63, now=64) == (True, True, True, 100, 64)


def test_order_of_a_camera_in_motion(policy):
    # creeping away from the order's view but at rest from frame to frame: the old order is dropped, the new view is sorted for at once
    assert order(policy, have=True, near_order=False, near_last=True, still=3, since=2, now=32) == (False, True, True, 4, 8)
    # moving from frame to frame: natural order, not measured, not sorted for — whatever order exists
    assert order(policy, have=True, near_order=False, near_last=False, still=7, since=20) == (False, False, False, 0, 8)
    # a jump back into the order's view: used at once (it was measured here), measured again from the next frame on
    assert order(policy, have=True, near_order=True, near_last=False, still=0, since=3) == (True, False, False, 0, 8)
    # interval 0 = measure but never sort
    assert order(policy, near_last=True, interval=0, now=0)[2] is False
