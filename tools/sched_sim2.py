"""Room-structure variants of the shipped schedule, replayed over the oracle's path events (see sched_sim.py).
Block costs: profiles/r1/v5_surface_pass_in_shade/block_profile_c2.txt (share / wave executions per sample)."""
import os
import sys
from collections import defaultdict

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import sched_sim as S  # noqa: E402
import ctypes as C  # noqa: E402

COST = {"closest": 4.47, "background": 4.03, "finish": 1.56, "surface": 6.3, "frame": 0.76, "nee": 4.5, "anyhit": 2.6, "eval": 5.95,
        "common": 3.0, "D": 0.39, "C": 6.15, "S": 8.4, "rest": 5.1, "vote": 0.15, "evalcc": 3.6}


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
            if with_eval:
                T.run("eval", sum(1 for l in lanes if cur(l)[1] == "v"))
                T.run("evalcc", sum(1 for l in lanes if cur(l)[1] == "v" and cur(l)[4]))
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
        elif variant == "coat":
            # two SHADE rooms by material class: surfaces with a clearcoat lobe wait apart, so the plain passes skip the
            # clearcoat code altogether; thr for the plain room, thr2 for the clearcoat room
            hp = [l for l in h if not cur(l)[4]]
            hc = [l for l in h if cur(l)[4]]
            def shade(lanes, coat):
                T.run("surface", len(lanes)); T.run("frame", len(lanes)); T.run("nee", len(lanes))
                T.run("anyhit", sum(1 for l in lanes if cur(l)[1] in "sv"))
                T.run("eval", sum(1 for l in lanes if cur(l)[1] == "v"))
                if coat: T.run("evalcc", sum(1 for l in lanes if cur(l)[1] == "v"))
                T.run("rest", len(lanes))
                lobes(lanes)
            if len(hc) >= thr2 or (not go and not hp and hc) or (not go and len(hc) > len(hp)):
                shade(hc, True)
            elif len(hp) >= thr or (not go and hp):
                shade(hp, False)
        elif variant == "chain":
            # HEAD fires at thr; its lanes join the LOBES room, which fires at thr (or with nobody left to trace / shade)
            if len(h) >= thr or (not go and h):
                head(h)
                for l in h: st[l] = "L"
                lo = [l for l in range(n) if st[l] == "L"]
            if len(lo) >= thr or (not go and not [l for l in range(n) if st[l] == "H"]):
                lobes(lo)
    return T


def sim_finish_room(pixels, thr=56, thr_f=40):
    """TRACE | SHADE | FINISH: a path that ends (miss: background still to be added; emitter; pdf <= 0; depth) waits in a third
    room; background + blend + the pixel's next camera path run when thr_f lanes wait there, or nobody can trace and it is the
    fuller room (round 3: the verdict's "vote-defer FINISH the way SHADE is deferred")."""
    T = T2()
    n = len(pixels)
    si = [0] * n; bi = [0] * n
    st = ["T"] * n
    def cur(l): return pixels[l][si[l]][bi[l]]
    while True:
        tr = [l for l in range(n) if st[l] == "T"]
        if tr:
            T.run("closest", len(tr))
            for l in tr:
                k = cur(l)[0]
                st[l] = "H" if k == "H" else ("B" if k == "M" else "F")
        T.run("vote", 64)
        go = [l for l in range(n) if st[l] == "T"]
        h = [l for l in range(n) if st[l] == "H"]
        f = [l for l in range(n) if st[l] in "BF"]
        if not go and not h and not f:
            break
        # after TRACE every live lane waits in one of the two rooms: SHADE at its threshold, else FINISH at its own, else the fuller
        run_shade = len(h) >= thr or (len(f) < thr_f and len(h) >= len(f))
        if run_shade and h:
            T.run("surface", len(h)); T.run("frame", len(h)); T.run("nee", len(h))
            T.run("anyhit", sum(1 for l in h if cur(l)[1] in "sv"))
            T.run("eval", sum(1 for l in h if cur(l)[1] == "v"))
            T.run("evalcc", sum(1 for l in h if cur(l)[1] == "v" and cur(l)[4]))
            T.run("rest", len(h)); T.run("common", len(h))
            for a in "DCS": T.run(a, sum(1 for l in h if cur(l)[2] == a))
            for l in h:
                if cur(l)[3]: st[l] = "F"
                else:
                    bi[l] += 1; st[l] = "T"
        else:
            T.run("background", sum(1 for l in f if st[l] == "B"))
            T.run("finish", len(f))
            for l in f:
                si[l] += 1; bi[l] = 0
                st[l] = "T" if si[l] < len(pixels[l]) else "X"
    return T


def sim_two_rooms_k(pixels, k, thr=56, swap_cost=0.6):
    """The shipped two rooms, but every lane owns k pixels (= k paths): one path in the working registers, the others
    parked (a parked path is its PathRegs + one dword: 18 VGPRs).  In a pass a lane takes part with any one of its paths
    that is in the room."""
    T = T2()
    n = len(pixels); L = n // k
    si = [0] * n; bi = [0] * n
    st = ["T"] * n
    def cur(q): return pixels[q][si[q]][bi[q]]
    def end_sample(q):
        si[q] += 1; bi[q] = 0
        st[q] = "F" if si[q] < len(pixels[q]) else "X"
    extra = 0.0
    def pick(l, states):
        for j in range(k):
            if st[l * k + j] in states: return l * k + j
        return None
    while True:
        # finish + trace: lanes with a path in F or T (F first: regenerate then trace in the same pass, like the kernel)
        chosen = [pick(l, "FT") for l in range(L)]
        tr = [q for q in chosen if q is not None]
        if tr:
            fin = [q for q in tr if st[q] == "F"]
            T.run("finish", len(fin))
            for q in fin: st[q] = "T"
            T.run("closest", len(tr))
            T.run("background", sum(1 for q in tr if cur(q)[0] == "M"))
            for q in tr:
                if cur(q)[0] == "H": st[q] = "H"
                else: end_sample(q)
            extra += swap_cost if k > 1 else 0.0
        T.run("vote", 64)
        go = sum(1 for l in range(L) if pick(l, "FT") is not None)
        hl = [pick(l, "H") for l in range(L)]
        h = [q for q in hl if q is not None]
        if not go and not h: break
        if len(h) >= thr or not go:
            T.run("surface", len(h)); T.run("frame", len(h)); T.run("nee", len(h))
            T.run("anyhit", sum(1 for q in h if cur(q)[1] in "sv"))
            T.run("eval", sum(1 for q in h if cur(q)[1] == "v"))
            T.run("rest", len(h)); T.run("common", len(h))
            for a in "DCS": T.run(a, sum(1 for q in h if cur(q)[2] == a))
            for q in h:
                if cur(q)[3]: end_sample(q)
                else:
                    bi[q] += 1; st[q] = "T"
            extra += swap_cost if k > 1 else 0.0
    return T, extra


if __name__ == "__main__":
    spp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    ntiles = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    oracle = S.oracle_lib.Oracle("liboracle.so")
    oracle.lib.oracle_sample_events.restype = C.c_int
    desc = oracle.scene_analytical()
    rng = np.random.default_rng(5)
    tiles = [(int(rng.integers(0, 1920 // 32)) * 32, int(rng.integers(0, 1080 // 32)) * 32) for _ in range(ntiles)]
    variants = [("two", 56, 0), ("two", 64, 0), ("coat", 56, 16), ("coat", 56, 32), ("coat", 48, 24), ("coat", 56, 48), ("three", 0, 24), ("three", 0, 40), ("three", 0, 56), ("chain", 56, 0), ("chain", 40, 0)]
    fvariants = [(56, 1), (56, 8), (56, 16), (56, 24), (56, 32), (56, 48), (56, 65), (48, 24), (40, 24), (48, 16), (64, 24), (64, 65), (40, 65)]
    fres = {v: T2() for v in fvariants}
    res = {v: T2() for v in variants}
    kres = {k_: [T2(), 0.0] for k_ in (1, 2, 4)}
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
                for v in fvariants:
                    fres[v].add(sim_finish_room(wave, v[0], v[1]))
        for k_ in kres:
            for wy in range(32 // (8 * k_)):
                for wx in range(4):
                    wave = [grid[wy * 8 * k_ + j * 8 + y][wx * 8 + x] for y in range(8) for x in range(8) for j in range(k_)]
                    T, extra = sim_two_rooms_k(wave, k_)
                    kres[k_][0].add(T); kres[k_][1] += extra
    base = res[("two", 56, 0)].time()
    for k_, (T, extra) in kres.items():
        t = T.time() + extra
        print("two rooms, %d paths per lane: cost/sample %.1f (%.3fx of shipped) useful %.1f%%  execs/sample: closest %.2f head %.2f" % (
            k_, 64 * t / ns, base / t, 100 * T.useful() / t, 64 * T.e["closest"] / ns, 64 * T.e["surface"] / ns))
    for v in fvariants:
        T = fres[v]
        print("finish room thr=%d/%d   cost/sample %.1f  (%.3fx of shipped)  useful %.1f%%   execs/sample: closest %.2f head %.2f finish %.2f" % (
            v[0], v[1], 64 * T.time() / ns, base / T.time(), 100 * T.useful() / T.time(), 64 * T.e["closest"] / ns, 64 * T.e["surface"] / ns, 64 * T.e["finish"] / ns))
    for v in variants:
        T = res[v]
        print("%-18s cost/sample %.1f  (%.3fx of shipped)  useful %.1f%%   execs/sample: closest %.2f head %.2f lobes %.2f" % (
            "%s thr=%d/%d" % v, 64 * T.time() / ns, base / T.time(), 100 * T.useful() / T.time(), 64 * T.e["closest"] / ns, 64 * T.e["surface"] / ns, 64 * T.e["common"] / ns))
