"""How many lanes are still walking in a bounce round?  Kernel body on the CPU (tests/host_harness, scripts/r04/path_start_sim.py's event logs), pose and spp from the command
line: per wave iteration of every bounce round the number of lanes whose ray is not finished, as cumulative shares.  usage: bounce_tail_lanes.py [pose] [spp]"""
import sys, ctypes as C
sys.path.insert(0, '/root/repo')
import numpy as np
exec(open('/root/repo/scripts/r04/path_start_sim.py').read().split("rows = {0:")[0])
rows = {0: (60, 110, 150, 190, 230, 262), 1: (20, 80, 140, 200, 250), 2: (30, 90, 150, 210, 260)}[pose]
tiles = [(tx, ty) for ty in rows for tx in (25, 100, 175, 250, 325, 400, 470)]
live = [t for t in tiles if beam_t0(t[0], t[1], 8) < 1e38]
hist = np.zeros(65); tot_iters = 0; rounds = 0
prim_iters = 0
for tx, ty in live:
    lanes = tile_events(tx, ty, 8, 0)
    # group by sample: per lane list of (kind, events)
    per = {}
    for i, l in enumerate(lanes):
        s = -1
        for k, q in l:
            if k == 0: s += 1
            per.setdefault((s, k), []).append(int(((q & 3) != 3).sum()))
    for (s, k), lens in per.items():
        if k == 2:
            lens = np.sort(np.array(lens))[::-1]
            m = lens[0]; rounds += 1; tot_iters += m
            for it in range(m):
                hist[int((lens > it).sum())] += 1
        if k == 0:
            prim_iters += max(lens)
c = np.cumsum(hist)
print(f"pose {pose}: {rounds} bounce rounds, {tot_iters / rounds:.1f} wave iterations each; primary rounds' iterations per bounce round {prim_iters / rounds:.1f}")
for thr in (1, 2, 4, 8, 16, 32):
    print(f"  iterations with <= {thr} lanes active: {c[thr] / tot_iters * 100:.1f} %")
