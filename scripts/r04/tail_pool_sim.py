import sys, ctypes as C
sys.path.insert(0, '/root/repo')
import numpy as np
exec(open('/root/repo/scripts/r04/path_start_sim.py').read().split("rows = {0:")[0])
rows = {0: (60, 110, 150, 190, 230, 262), 1: (20, 80, 140, 200, 250), 2: (30, 90, 150, 210, 260)}[pose]
tiles = [(tx, ty) for ty in rows for tx in (25, 100, 175, 250, 325, 400, 470)]
live = [t for t in tiles if beam_t0(t[0], t[1], 8) < 1e38]
res = {}
base_tot = 0; other_tot = 0
per_tile = []
for tx, ty in live:
    lanes = tile_events(tx, ty, 8, 0)
    per = {}
    for i, l in enumerate(lanes):
        s = -1
        for k, q in l:
            if k == 0: s += 1
            per.setdefault((s, k), []).append(int(((q & 3) != 3).sum()))
    per_tile.append(per)
for K in (16, 24, 32, 48, 10**6):
    for T in (32, 64):
        for R in (6,):
            total = 0; tails = 0; tail_it = 0
            for per in per_tile:
                pool = []
                for (s, k) in sorted(per):
                    lens = np.array(per[(s, k)])
                    if k != 2 or K >= 10**6:
                        total += lens.max()
                        continue
                    total += min(lens.max(), K)
                    pool += [int(x - K + R) for x in lens if x > K]
                    while len(pool) >= T:
                        batch, pool = pool[:64], pool[64:]
                        total += max(batch); tails += 1; tail_it += max(batch)
                if pool:
                    total += max(pool); tails += 1; tail_it += max(pool)
            res[(K, T)] = (total, tails, tail_it)
base = res[(10**6, 32)][0]
print(f"pose {pose}, {spp} spp, {len(per_tile)} live wave tiles: wave iterations (all kinds) {base} as now")
for (K, T), (tot, tails, tail_it) in res.items():
    if K < 10**6: print(f"  bounce rounds capped at {K}, tail round from {T} parked rays (restart = 6 iterations): {tot / base:.3f}x, {tails} tail rounds of {tail_it / max(tails, 1):.0f} iterations")
print("tail rounds capped too (unfinished rays parked again):")
for K in (16, 24, 32):
    for K2 in (24, 32, 48):
        T, R = 64, 6
        total = 0; tails = 0
        for per in per_tile:
            pool = []
            def drain(final):
                global total, tails, pool
                while len(pool) >= T or (final and pool):
                    batch, pool = pool[:64], pool[64:]
                    total += min(max(batch), K2 + R); tails += 1
                    again = [x - K2 for x in batch if x > K2 + R]
                    if final and not pool and all(x <= K2 + R for x in batch): pass
                    pool += [x + R for x in again]
            for (s, k) in sorted(per):
                lens = np.array(per[(s, k)])
                if k != 2:
                    total += lens.max(); continue
                total += min(lens.max(), K)
                pool += [int(x - K + R) for x in lens if x > K]
                drain(False)
            drain(True)
        print(f"  cap {K}, tail cap {K2}: {total / base:.3f}x, {tails} tail rounds")
print("cut by lanes still walking (<= N, after at least M trips) instead of by trips; rounds over parked rays the same way:")
for N in (8, 16, 24, 32):
    for M in (8, 16):
        T, R = 64, 6
        total = 0; tails = 0
        def cut_round(lens):
            lens = np.sort(np.array(lens))[::-1]
            # trips until at most N lanes are left (the (N+1)-th longest ray's length), at least M, at most the longest
            k = lens[N] if len(lens) > N else 0
            trips = min(max(k, M), lens[0])
            return trips, [int(x - trips + R) for x in lens if x > trips]
        for per in per_tile:
            pool = []
            def drain(final):
                global total, tails, pool
                while len(pool) >= T or (final and pool):
                    batch, pool = pool[:64], pool[64:]
                    if final and len(batch) <= N: total += max(batch); tails += 1; continue
                    trips, left = cut_round(batch)
                    total += trips; tails += 1
                    pool += left
            for (s, k) in sorted(per):
                lens = per[(s, k)]
                if k != 2:
                    total += max(lens); continue
                trips, left = cut_round(lens)
                total += trips; pool += left
                drain(False)
            drain(True)
        print(f"  N {N}, M {M}: {total / base:.3f}x, {tails} rounds over parked rays")
