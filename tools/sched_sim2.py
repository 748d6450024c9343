"""Room-structure variants of the shipped schedule, replayed over the oracle's path events (see sched_sim.py).
Block costs: profiles/r1/v5_surface_pass_in_shade/block_profile_c2.txt (share / wave executions per sample)."""
import os
import sys
from collections import defaultdict

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import sched_sim as S  # noqa: E402
import ctypes as C  # noqa: E402

COST = {"closest": 4.47, "background": 4.03, "finish": 1.56, "surface": 6.3, "frame": 0.76, "nee": 4.5, "anyhit": 2.6, "eval": 8.25,
        "common": 3.0, "D": 0.39, "C": 6.15, "S": 8.4, "rest": 5.1, "vote": 0.15}


class T2:
    def __init__(self):
        self.e = defaultdict(int); self.l = defaultdict(int)
    def run(self, b, n):
        if n > 0:
            self.e[b] += 1; self.l[b] += n
    def add(self, o):
        for k in o.e:
            self.e[k] += o.e[k]; self.l[k] += o.l[k]
    def time(self):
        return sum(COST[k] * self.e[k] for k in self.e)
    def useful(self):
        return sum(COST[k] * self.l[k] / 64.0 for k in self.e)


def sim_rooms(pixels, variant, thr=56, thr2=40):
    """variant 'two': TRACE | SHADE (shipped).  'three': TRACE | HEAD (surface, frame, nee, anyhit, eval) | LOBES.
    'evalroom': TRACE | SHADE without eval | EVAL+LOBES ..."""
    T = T2()
    n = len(pixels)
    si = [0] * n; bi = [0] * n
    st = ["F0"] * n   # F0: needs first camera ray (free)
    st = ["T"] * n
    def cur(l): return pixels[l][si[l]][bi[l]]
    def end_sample(l):
        si[l] += 1; bi[l] = 0
        st[l] = "F" if si[l] < len(pixels[l]) else "X"
        if st[l] == "X": pass
    while True:
        fin = [l for l in range(n) if st[l] == "F"]
        if fin:
            T.run("finish", len(fin))
            for l in fin: st[l] = "T"
        # lanes that just retired count as X
        tr = [l for l in range(n) if st[l] == "T"]
        if tr:
            T.run("closest", len(tr))
            miss = [l for l in tr if cur(l)[0] == "M"]
            T.run("background", len(miss))
            for l in tr:
                k = cur(l)[0]
                if k == "H": st[l] = "H"
                else: end_sample(l)
        T.run("vote", 64)
        go = [l for l in range(n) if st[l] in "TF"]
        h = [l for l in range(n) if st[l] == "H"]
        lo = [l for l in range(n) if st[l] == "L"]
        if not go and not h and not lo:
            break
        def head(lanes, with_eval=True):
            T.run("surface", len(lanes)); T.run("frame", len(lanes)); T.run("nee", len(lanes))
            T.run("anyhit", sum(1 for l in lanes if cur(l)[1] in "sv"))
            if with_eval: T.run("eval", sum(1 for l in lanes if cur(l)[1] == "v"))
            T.run("rest", len(lanes))
        def lobes(lanes):
            T.run("common", len(lanes))
            for a in "DCS": T.run(a, sum(1 for l in lanes if cur(l)[2] == a))
            for l in lanes:
                if cur(l)[3]: end_sample(l)
                else:
                    bi[l] += 1; st[l] = "T"
        if variant == "two":
            if len(h) >= thr or not go:
                head(h); lobes(h)
        elif variant == "three":
            # the fuller of the two waiting rooms fires when it reaches its threshold, or when nobody can trace
            if len(lo) >= thr2 or (not go and not h) or (not go and len(lo) >= len(h)):
                lobes(lo)
            elif len(h) >= thr2 or not go:
                head(h)
                for l in h: st[l] = "L"
        elif variant == "chain":
            # HEAD fires at thr; its lanes join the LOBES room, which fires at thr (or with nobody left to trace / shade)
            if len(h) >= thr or (not go and h):
                head(h)
                for l in h: st[l] = "L"
                lo = [l for l in range(n) if st[l] == "L"]
            if len(lo) >= thr or (not go and not [l for l in range(n) if st[l] == "H"]):
                lobes(lo)
    return T


if __name__ == "__main__":
    spp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    ntiles = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    oracle = S.oracle_lib.Oracle("liboracle.so")
    oracle.lib.oracle_sample_events.restype = C.c_int
    desc = oracle.scene_analytical()
    rng = np.random.default_rng(5)
    tiles = [(int(rng.integers(0, 1920 // 32)) * 32, int(rng.integers(0, 1080 // 32)) * 32) for _ in range(ntiles)]
    variants = [("two", 56, 0), ("two", 64, 0), ("three", 0, 24), ("three", 0, 40), ("three", 0, 56), ("chain", 56, 0), ("chain", 40, 0)]
    res = {v: T2() for v in variants}
    ns = 0
    for (c0, r0) in tiles:
        px = S.tile_events(oracle, desc, c0, r0, 32, 32, spp)
        grid = [[px[r * 32 + c] for c in range(32)] for r in range(32)]
        ns += 1024 * spp
        for wy in range(4):
            for wx in range(4):
                wave = [grid[wy * 8 + y][wx * 8 + x] for y in range(8) for x in range(8)]
                for v in variants:
                    res[v].add(sim_rooms(wave, v[0], v[1], v[2]))
    base = res[("two", 56, 0)].time()
    for v in variants:
        T = res[v]
        print("%-18s cost/sample %.1f  (%.3fx of shipped)  useful %.1f%%   execs/sample: closest %.2f head %.2f lobes %.2f" % (
            "%s thr=%d/%d" % v, 64 * T.time() / ns, base / T.time(), 100 * T.useful() / T.time(), 64 * T.e["closest"] / ns, 64 * T.e["surface"] / ns, 64 * T.e["common"] / ns))
