"""Replay the oracle's path-event traces through candidate wave schedules (CPU only, no GPU needed).

The megakernel's cost is dominated by lane divergence, i.e. by WHEN each block of the path runs and with how many
lanes.  This tool takes the real event sequences of BASELINE configs[1] (oracle_sample_events: miss / emitter /
hit, light facing / shadowed / evaluated, sampled lobe, pdf <= 0) for a set of tiles, and counts for each
schedule how many wave-level executions of each block it needs, weighted by the block costs measured on the GPU
with tools/block_profile.py.  Schedules:

  current   the shipped kernel: two rooms per wave (TRACE with miss/regeneration inline, SHADE with the lobes
            inline), SHADE fires at >= 56 waiting lanes
  pool      a workgroup-wide pool: paths live in LDS queues, one per stage; a wave takes up to 64 paths of the
            fullest queue, runs that stage, and pushes them to their next queues (pool = pixels per workgroup)

usage: python tools/sched_sim.py [spp] [tiles]
"""
import ctypes as C
import heapq
import os
import sys
from collections import defaultdict

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest  # noqa: E402
import oracle_lib  # noqa: E402

conftest.load_package()

# block costs: time share / wave executions per sample, from gpurun_out/block_profile.txt (c2, 32 spp)
COST = {"closest": 25.2 / 3.177, "background": 7.9 / 1.786, "finalize": 7.6 / 2.838, "finish": 4.5 / 3.109,
        "head": (1.0 + 6.8 + 7.0) / 1.578, "anyhit": 3.7 / 1.568, "eval": 11.7 / 1.478,
        "D": 3.8 / 1.400, "C": 13.6 / 0.992, "S": 12.3 / 1.548, "tail": 0.7 / 1.575}


def parse(events):
    """bytes -> list of samples; a sample is a list of bounces ('M',) | ('E',) | ('H', nee, lobe, ended)"""
    samples, cur, i = [], [], 0
    ev = events.decode("ascii")
    n = len(ev)
    while i < n:
        c = ev[i]
        if c == ".":
            samples.append(cur)
            cur = []
            i += 1
        elif c in "ME":
            cur.append((c,))
            i += 1
        else:
            assert c == "H", c
            coat = ev[i + 1] == "k"
            if coat:
                i += 1
            nee, lobe = ev[i + 1], ev[i + 2]
            i += 3
            ended = False
            if ev[i] == "x":
                ended = True
                i += 1
            if ev[i] == ".":
                ended = True                      # depth exhausted
            cur.append(("H", nee, lobe, ended, coat))
    return samples


def tile_events(oracle, desc, col0, row0, cw, ch, spp, W=1920, H=1080):
    cap = cw * ch * spp * 24
    buf = (C.c_uint8 * cap)()
    n = oracle.lib.oracle_sample_events(C.byref(desc), col0, row0, cw, ch, C.c_uint64(0), spp, W, H, C.c_uint64(1), buf, C.c_uint64(cap))
    assert n > 0, n
    samples = parse(bytes(buf[:n]))
    assert len(samples) == cw * ch * spp
    return [samples[p * spp:(p + 1) * spp] for p in range(cw * ch)]       # per pixel


class Tally:
    def __init__(self):
        self.execs = defaultdict(int)
        self.lanes = defaultdict(int)

    def run(self, block, n):
        if n > 0:
            self.execs[block] += 1
            self.lanes[block] += n

    def add(self, other):
        for k in other.execs:
            self.execs[k] += other.execs[k]
            self.lanes[k] += other.lanes[k]

    def time(self, overhead_per_exec=None):
        t = sum(COST[k] * self.execs[k] for k in self.execs if k in COST)
        if overhead_per_exec:
            t += sum(v * self.execs[k] for k, v in overhead_per_exec.items())
        return t

    def useful(self):
        return sum(COST[k] * self.lanes[k] / 64.0 for k in self.execs if k in COST)


def sim_current(pixels, threshold=56):
    """64 pixels, one per lane; the shipped two-room schedule."""
    T = Tally()
    n = len(pixels)
    si = [0] * n                 # sample index
    bi = [0] * n                 # bounce index within the sample
    state = ["T"] * n            # T trace, S waiting for shade, X done
    while True:
        tr = [l for l in range(n) if state[l] == "T"]
        if tr:
            T.run("closest", len(tr))
            miss = [l for l in tr if pixels[l][si[l]][bi[l]][0] == "M"]
            hit = [l for l in tr if pixels[l][si[l]][bi[l]][0] != "M"]
            T.run("background", len(miss))
            T.run("finalize", len(hit))
            fin = miss + [l for l in hit if pixels[l][si[l]][bi[l]][0] == "E"]
            T.run("finish", len(fin))
            for l in hit:
                if pixels[l][si[l]][bi[l]][0] == "H":
                    state[l] = "S"
            for l in fin:
                si[l] += 1
                bi[l] = 0
                state[l] = "T" if si[l] < len(pixels[l]) else "X"
        sh = [l for l in range(n) if state[l] == "S"]
        tr = [l for l in range(n) if state[l] == "T"]
        if not sh and not tr:
            break
        if len(sh) >= threshold or not tr:
            T.run("head", len(sh))
            b = {l: pixels[l][si[l]][bi[l]] for l in sh}
            T.run("anyhit", sum(1 for l in sh if b[l][1] in "sv"))
            T.run("eval", sum(1 for l in sh if b[l][1] == "v"))
            for lobe in "DCS":
                T.run(lobe, sum(1 for l in sh if b[l][2] == lobe))
            T.run("tail", len(sh))
            fin = [l for l in sh if b[l][3]]
            T.run("finish", len(fin))
            for l in sh:
                if b[l][3]:
                    si[l] += 1
                    bi[l] = 0
                    state[l] = "T" if si[l] < len(pixels[l]) else "X"
                else:
                    bi[l] += 1
                    state[l] = "T"
    return T


# stages of the pooled schedule and the blocks a batch of that stage executes
def sim_pool(pixels, n_waves, stages="full", min_batch=1):
    """Workgroup pool: len(pixels) paths, n_waves waves of 64 lanes, one LDS queue per stage."""
    T = Tally()
    n = len(pixels)
    si = [0] * n
    bi = [0] * n
    queues = defaultdict(list)
    queues["T"] = list(range(n))
    live = n
    clock = 0.0
    # event-driven: each wave is busy until `t`; when free it takes the fullest queue
    waves = [(0.0, w) for w in range(n_waves)]
    heapq.heapify(waves)
    inflight = {}                # wave -> (stage, batch) to be retired at its finish time

    def stage_cost(stage, batch):
        b = [pixels[p][si[p]][bi[p]] for p in batch]
        k = len(batch)
        if stage == "T":
            T.run("closest", k)
            return COST["closest"]
        if stage == "M":                                      # miss: background + blend + next camera path
            T.run("background", k); T.run("finish", k)
            return COST["background"] + COST["finish"]
        if stage == "E":                                      # emitter exit
            T.run("finalize", k); T.run("finish", k)
            return COST["finalize"] + COST["finish"]
        if stage == "H":                                      # finalize + frame + light sample + shadow ray
            T.run("finalize", k); T.run("head", k)
            na = sum(1 for x in b if x[1] in "sv")
            T.run("anyhit", na)
            c = COST["finalize"] + COST["head"] + (COST["anyhit"] if na else 0.0)
            if stages != "full":                              # eval inline
                ne = sum(1 for x in b if x[1] == "v")
                T.run("eval", ne)
                c += COST["eval"] if ne else 0.0
            return c
        if stage == "V":
            T.run("eval", k)
            return COST["eval"]
        if stage in "DCS":
            T.run(stage, k); T.run("tail", k)
            nf = sum(1 for x in b if x[3])
            T.run("finish", nf)
            return COST[stage] + COST["tail"] + (COST["finish"] if nf else 0.0)
        raise AssertionError(stage)

    def retire(stage, batch):
        nonlocal live
        for p in batch:
            b = pixels[p][si[p]][bi[p]]
            if stage == "T":
                queues[b[0]].append(p)                        # M / E / H
            elif stage in "ME":
                si[p] += 1; bi[p] = 0
                if si[p] < len(pixels[p]): queues["T"].append(p)
                else: live -= 1
            elif stage == "H":
                if stages == "full" and b[1] == "v": queues["V"].append(p)
                else: queues[b[2]].append(p)
            elif stage == "V":
                queues[b[2]].append(p)
            else:
                if b[3]:
                    si[p] += 1; bi[p] = 0
                    if si[p] < len(pixels[p]): queues["T"].append(p)
                    else: live -= 1
                else:
                    bi[p] += 1
                    queues["T"].append(p)

    busy_time = 0.0
    idle_time = 0.0
    while live > 0 or inflight:
        t, w = heapq.heappop(waves)
        if w in inflight:
            retire(*inflight.pop(w))
        # pick the fullest queue
        best, bl = None, 0
        for s, q in queues.items():
            if len(q) > bl:
                best, bl = s, len(q)
        if best is None or (bl < min_batch and inflight):
            if not inflight and best is None:
                break
            # nothing (worth) taking: wait for the next wave to retire
            nxt = min(x[0] for x in waves) if waves else t
            nxt = max(nxt, t + 0.05)
            idle_time += nxt - t
            heapq.heappush(waves, (nxt, w))
            continue
        q = queues[best]
        batch = q[:64]
        del q[:64]
        c = stage_cost(best, batch) + POOL_OVERHEAD
        busy_time += c
        inflight[w] = (best, batch)
        heapq.heappush(waves, (t + c, w))
        clock = max(clock, t + c)
    return T, busy_time, idle_time


def sim_pairs(pixels, k=2, swap_cost=0.5, stages="full"):
    """Wave-private pool: 64 lanes, k pixels (= paths) per lane; one path of a lane is in registers, the others are parked
    in the lane's own LDS slots.  Each pass the wave runs the stage with the most lanes that hold a path in it."""
    T = Tally()
    n = len(pixels)
    L = n // k
    si = [0] * n
    bi = [0] * n
    st = ["T"] * n               # stage of each path, "X" done
    active = [l * k for l in range(L)]     # which of its paths lane l holds in registers
    busy = 0.0
    npass = 0
    nswap = 0
    order = "TMEHVDCS"
    while True:
        cand = {}
        for s_ in order:
            cand[s_] = sum(1 for l in range(L) if any(st[l * k + j] == s_ for j in range(k)))
        best = max(order, key=lambda s_: cand[s_])
        if cand[best] == 0:
            break
        batch = []
        swapped = False
        for l in range(L):
            if st[active[l]] != best:
                for j in range(k):
                    if st[l * k + j] == best:
                        active[l] = l * k + j
                        swapped = True
                        break
            if st[active[l]] == best:
                batch.append(active[l])
        npass += 1
        if swapped:
            busy += swap_cost
            nswap += 1
        kk = len(batch)
        b = [pixels[p][si[p]][bi[p]] for p in batch]
        c = 0.15                                               # the vote
        if best == "T":
            T.run("closest", kk); c += COST["closest"]
        elif best == "M":
            T.run("background", kk); T.run("finish", kk); c += COST["background"] + COST["finish"]
        elif best == "E":
            T.run("finalize", kk); T.run("finish", kk); c += COST["finalize"] + COST["finish"]
        elif best == "H":
            T.run("finalize", kk); T.run("head", kk)
            na = sum(1 for x in b if x[1] in "sv")
            T.run("anyhit", na)
            c += COST["finalize"] + COST["head"] + (COST["anyhit"] if na else 0.0)
            if stages != "full":
                ne = sum(1 for x in b if x[1] == "v")
                T.run("eval", ne); c += COST["eval"] if ne else 0.0
        elif best == "V":
            T.run("eval", kk); c += COST["eval"]
        else:
            T.run(best, kk); T.run("tail", kk)
            nf = sum(1 for x in b if x[3])
            T.run("finish", nf)
            c += COST[best] + COST["tail"] + (COST["finish"] if nf else 0.0)
        busy += c
        for p, x in zip(batch, b):
            if best == "T":
                st[p] = x[0]
            elif best in "ME":
                si[p] += 1; bi[p] = 0
                st[p] = "T" if si[p] < len(pixels[p]) else "X"
            elif best == "H":
                st[p] = "V" if (stages == "full" and x[1] == "v") else x[2]
            elif best == "V":
                st[p] = x[2]
            else:
                if x[3]:
                    si[p] += 1; bi[p] = 0
                    st[p] = "T" if si[p] < len(pixels[p]) else "X"
                else:
                    bi[p] += 1
                    st[p] = "T"
    return T, busy, npass, nswap


# name -> (lanes per workgroup, pixels (= paths) per workgroup, "full" = eval as its own stage | "fused", smallest batch a wave
# takes while other batches are in flight)
CONFIGS = {
    "pool 256 lanes / 256 paths": (256, 256, "full", 1),
    "pool 256 lanes / 256 paths, batches >= 32": (256, 256, "full", 32),
    "pool 256 lanes / 512 paths, batches >= 32": (256, 512, "full", 32),
    "pool 512 lanes / 512 paths, batches >= 32": (512, 512, "full", 32),
    "pool 512 lanes / 1024 paths, batches >= 48": (512, 1024, "full", 48),
    "pool 512 lanes / 1024 paths, eval fused, batches >= 48": (512, 1024, "fused", 48),
}
PAIRS = {"pairs k=1 (rooms in registers only)": (1, "full"), "pairs k=2": (2, "full"), "pairs k=2, eval fused": (2, "fused"), "pairs k=4": (4, "full")}
POOL_OVERHEAD = 0.8     # per batch: context load/store, queue push/pop (~50 wave instructions; 1 unit ~ 60)

if __name__ == "__main__":
    spp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    ntiles = int(sys.argv[2]) if len(sys.argv) > 2 else 24
    oracle = oracle_lib.Oracle("liboracle.so")
    oracle.lib.oracle_sample_events.restype = C.c_int
    desc = oracle.scene_analytical()
    rng = np.random.default_rng(5)
    # 32x32-pixel tiles at random positions of the 1920x1080 frame
    tiles = [(int(rng.integers(0, 1920 // 32)) * 32, int(rng.integers(0, 1080 // 32)) * 32) for _ in range(ntiles)]
    cur = Tally()
    t_cur = 0.0
    res = defaultdict(lambda: [Tally(), 0.0, 0.0])
    pairs = defaultdict(lambda: [Tally(), 0.0, 0, 0])
    n_samples = 0
    for (c0, r0) in tiles:
        px = tile_events(oracle, desc, c0, r0, 32, 32, spp)                # 1024 pixels, row-major within the tile
        grid = [[px[r * 32 + c] for c in range(32)] for r in range(32)]
        n_samples += 1024 * spp
        # current: 16 waves of 8x8
        for wy in range(4):
            for wx in range(4):
                wave = [grid[wy * 8 + y][wx * 8 + x] for y in range(8) for x in range(8)]
                cur.add(sim_current(wave))
        # wave-private pairs: a wave owns an 8 x (8k) block, lane = column/row within 8x8, k pixels per lane stacked vertically
        for name, (k, stg) in PAIRS.items():
            for wy in range(32 // (8 * k)):
                for wx in range(4):
                    wave = [grid[wy * 8 * k + j * 8 + y][wx * 8 + x] for y in range(8) for x in range(8) for j in range(k)]
                    T, busy, npass, nswap = sim_pairs(wave, k, 0.5, stg)
                    pairs[name][0].add(T); pairs[name][1] += busy; pairs[name][2] += npass; pairs[name][3] += nswap
        # pooled: workgroups of `lanes` lanes owning `pool` pixels of the tile
        for name, (lanes, pool, stages, mb) in CONFIGS.items():
            for g in range(1024 // pool):
                T, busy, idle = sim_pool(px[g * pool:(g + 1) * pool], lanes // 64, stages, mb)
                res[name][0].add(T); res[name][1] += busy; res[name][2] += idle
    print("samples: %d in %d tiles" % (n_samples, ntiles))
    tc = cur.time()
    print("\ncurrent schedule: cost %.1f per sample (useful %.1f): lane utilisation %.1f%%" % (64 * tc / n_samples, 64 * cur.useful() / n_samples, 100 * cur.useful() / tc))
    for k in ["closest", "background", "finalize", "finish", "head", "anyhit", "eval", "D", "C", "S", "tail"]:
        print("   %-11s execs/sample %.3f lanes/64 %5.1f%%" % (k, 64 * cur.execs[k] / n_samples, 100.0 * cur.lanes[k] / (64.0 * max(1, cur.execs[k]))))
    for name, (T, busy, npass, nswap) in pairs.items():
        print("\n%s: busy %.1f per sample -> %.2fx the current schedule; useful/busy %.1f%%; passes/sample %.2f, with swaps %.2f" % (
            name, 64 * busy / n_samples, tc / busy, 100 * T.useful() / busy, 64 * npass / n_samples, 64 * nswap / n_samples))
        for k in ["closest", "background", "finalize", "finish", "head", "anyhit", "eval", "D", "C", "S", "tail"]:
            print("   %-11s execs/sample %.3f lanes/64 %5.1f%%" % (k, 64 * T.execs[k] / n_samples, 100.0 * T.lanes[k] / (64.0 * max(1, T.execs[k]))))
    for name, (T, busy, idle) in res.items():
        print("\n%s: busy %.1f per sample, idle %.1f -> %.2fx the current schedule (%.2fx counting idle waves); useful/busy %.1f%%" % (
            name, 64 * busy / n_samples, 64 * idle / n_samples, tc / busy, tc / (busy + idle), 100 * T.useful() / busy))
        for k in ["closest", "background", "finalize", "finish", "head", "anyhit", "eval", "D", "C", "S", "tail"]:
            print("   %-11s execs/sample %.3f lanes/64 %5.1f%%" % (k, 64 * T.execs[k] / n_samples, 100.0 * T.lanes[k] / (64.0 * max(1, T.execs[k]))))
