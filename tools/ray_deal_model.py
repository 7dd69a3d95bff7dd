"""Model behind DESIGN.md 4.2 (5): what the raycast's workgroup -> tile-pair deal does to the cost sums per compute unit and
per SIMD, under the dispatch rule observed with tools/ray_diag.py (workgroup i -> compute unit i mod 256; its two waves ->
one SIMD pair, the pairs alternating per round).  Input: per-tile costs (trips + 5 * march batches of the slowest ray) of one
640x480 -> 512^3 launch, profiles/r02c_tile_costs_sdf512.npy.  Prints min / median / max of the sums for the image order,
for the snake deal of cost-sorted pairs that is built in, and for a round-wise LPT deal, with the finish time a linear fit
of the measured SIMD finish times predicts (22 us + 0.058 us per cost unit of the most loaded SIMD).   CPU only."""
import os
import sys
import numpy as np

path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r02c_tile_costs_sdf512.npy")
cost = np.load(path).astype(np.float64)
n, n_cus = len(cost), 256
pc = cost[0::2] + cost[1::2]
npairs = len(pc)
rounds = (npairs + n_cus - 1) // n_cus


def sums(assign):
    s = np.zeros((n_cus, 4))
    for wg, p in enumerate(assign):
        if p < 0:
            continue
        cu, rnd = wg % n_cus, wg // n_cus
        s[cu, (rnd % 2) * 2 + 0] += cost[2 * p]
        s[cu, (rnd % 2) * 2 + 1] += cost[2 * p + 1]
    return s


def report(name, assign):
    s = sums(assign)
    cu = s.sum(1)
    print(f"{name:28s} compute-unit sums {cu.min():5.0f} / {np.median(cu):5.0f} / {cu.max():5.0f}   SIMD sums {s.min():4.0f} / {np.median(s):4.0f} / {s.max():4.0f}"
          f"   predicted launch {22 + 0.058 * s.max():.1f} us")


report("image order", list(range(npairs)))
order = np.argsort(-pc, kind="stable")
snake = [-1] * (rounds * n_cus)
for wg in range(rounds * n_cus):
    rnd, cu = wg // n_cus, wg % n_cus
    pos = rnd * n_cus + (n_cus - 1 - cu if rnd & 1 else cu)
    if pos < npairs:
        snake[wg] = order[pos]
report("snake deal (built in)", snake)
load = np.zeros(n_cus)
lpt = [-1] * (rounds * n_cus)
for rnd in range(rounds):
    chunk = order[rnd * n_cus:(rnd + 1) * n_cus]
    rank = np.argsort(load, kind="stable")
    for k, p in enumerate(chunk):
        lpt[rnd * n_cus + rank[k]] = p
        load[rank[k]] += pc[p]
report("round-wise LPT", lpt)
