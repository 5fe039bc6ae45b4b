"""A queueing model of k_trace<0, 0>'s wave-level step scheduling: 64 lanes, each holding R rays; a step is either "expand an
interior record" or "test a triangle" for every lane that has a ray in that state; costs in vector instructions from the ISA (122 / 200 per
iteration, 174 per refill).  With R = 1 it reproduces the measured lane use (42.8 / 20.3 lanes per interior / leaf step; measured 42.6 / 20.6,
profiles/r03w_trace_step_statistics.txt).  R = 2 (two rays per lane, state picked by selects: +15 / +20 instructions per step) predicts
-14 % vector instructions per ray at best (leaf weighting 8).  python tools/experiments/sim_lane_use.py"""
import numpy as np, sys
rng = np.random.default_rng(1)
def make_ray():
    # sequence of step kinds: 0 interior, 1 leaf-test; ~50 interior lane-steps, ~3 leaf visits of ~2.2 tests
    seq = []
    nleaf = max(1, rng.poisson(3.0))
    for _ in range(nleaf):
        seq += [0] * max(1, int(rng.geometric(1 / (50.3 / (nleaf + 0.0))) ))
        seq += [1] * max(1, int(rng.geometric(1 / 2.2)))
    seq += [0] * int(rng.geometric(1 / 3.0))
    return seq
def simulate(R, triW, refill, nrays=20000, cost_int=122, cost_leaf=200, sel_int=0, sel_leaf=0):
    L = 64
    slots = [[None] * R for _ in range(L)]  # each: [seq, pos]
    pool = nrays
    valu = 0; steps = [0, 0]; lanes = [0, 0]
    def fill():
        nonlocal pool
        for l in range(L):
            for r in range(R):
                if slots[l][r] is None and pool > 0:
                    slots[l][r] = [make_ray(), 0]; pool -= 1
    fill()
    while True:
        # count idle slots
        idle = sum(1 for l in range(L) for r in range(R) if slots[l][r] is None)
        if pool > 0 and idle >= refill:
            fill(); valu += 174; continue
        elig = [[0, 0] for _ in range(L)]
        for l in range(L):
            for r in range(R):
                s = slots[l][r]
                if s is not None: elig[l][s[0][s[1]]] = 1
        nInt = sum(e[0] for e in elig); nTri = sum(e[1] for e in elig)
        if nInt == 0 and nTri == 0:
            if pool == 0: break
            fill(); valu += 174; continue
        kind = 1 if (nTri > 0 and (nInt == 0 or nTri * 16 >= nInt * triW)) else 0
        valu += (cost_leaf + sel_leaf) if kind else (cost_int + sel_int)
        steps[kind] += 1
        for l in range(L):
            for r in range(R):
                s = slots[l][r]
                if s is not None and s[0][s[1]] == kind:
                    s[1] += 1; lanes[kind] += 1
                    if s[1] >= len(s[0]): slots[l][r] = None
                    break  # one ray per lane per step
    return valu / nrays, lanes[0] / max(1, steps[0]), lanes[1] / max(1, steps[1])
print("R=1", simulate(1, 8, 16))
for triW in (8, 16, 24, 32):
    print("R=2 triW", triW, simulate(2, triW, 16, sel_int=15, sel_leaf=20))
print("R=3 triW 24", simulate(3, 24, 16, sel_int=30, sel_leaf=40))
