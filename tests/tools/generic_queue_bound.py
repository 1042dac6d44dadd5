#!/usr/bin/env python3
"""TEST INFRASTRUCTURE -- bound of what a chain queue gives BATCHES of generic chains (round-4 review, item 2), in wave
passes, from the C oracle's per-frame evaluation counts (a lane's passes for a frame = nfev - 1 trial evaluations, + 1
when scipy ends the solve with status 1: the gtol test sits at the top of the next pass -- scripts/bench_generic.py).

Workload: scripts/bench_generic.py's -- windows of T = 32 frames of the shipped 6000-frame recording (offsets 11 s mod 5968),
legs RF + LF with the generic chain, one lane per chain, leg-pure wavefronts of 64 consecutive sequences.

Schedules on a GPU with `slots` resident wavefronts (1024: one 256-VGPR wavefront per SIMD):
  static   what seqik_generic_kernel<.., grouped = 0> does: a wavefront owns 64 chains and lives as long as its slowest lane;
           the hardware dispatcher starts the next wavefront on a slot when one retires
  queue    `slots` persistent wavefronts; a lane that finishes its chain pulls the next chain of its leg from a per-leg
           counter (greedy list scheduling in submission order)
  floor    max(slowest chain, all lane passes / (64 x slots))
A wave pass is priced the same whatever the number of active lanes (the generic pass has few conditional blocks; measured
pass times: 7.5-10.5 us).  Prints JSON.

    python tests/tools/generic_queue_bound.py > profiles/r05_generic_queue_bound.json      (CPU, about a minute)
"""
import heapq
import json
import os
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from oracle import c_oracle  # noqa: E402

T, SLOTS = 32, 1024


def chain_passes():
    z = np.load(os.path.join(ROOT, "tests", "golden", "anipose_shipped.npz"))
    out = {}
    c_oracle.lib()
    for leg in ("RF", "LF"):
        pose, seg, b, seed9 = z[f"{leg}_pose"], z[f"{leg}_seg"], z[f"{leg}_bounds"], z[f"{leg}_seeds"][18:27]

        def one(o):
            r = c_oracle.generic_leg(pose[o:o + T], seg, b, seed9, want_fk=False)
            return int((r["nfev"] - 1 + (r["status"] == 1)).sum())
        with ThreadPoolExecutor(8) as ex:
            out[leg] = np.array(list(ex.map(one, range(6000 - T))), dtype=np.int64)
    return out


def static_makespan(passes_by_leg, S, slots):
    """waves of 64 consecutive sequences per leg, wave w of leg l dispatched in grid order (legs interleaved as the kernel's
    leg-pure grid: all waves of the first leg, then the next), each on the earliest free slot"""
    waves = []
    for p in passes_by_leg:
        pad = (-S) % 64
        q = np.concatenate([p[:S], np.zeros(pad, np.int64)])
        waves += list(q.reshape(-1, 64).max(1))
    free = [0] * min(slots, len(waves))
    heapq.heapify(free)
    end = 0
    for w in waves:
        t0 = heapq.heappop(free)
        heapq.heappush(free, t0 + int(w))
        end = max(end, t0 + int(w))
    return end, float(np.mean(waves)), int(np.max(waves))


def queue_makespan(passes_by_leg, S, slots):
    """slots wavefronts split over the legs in proportion to their work; every lane of a leg's wavefronts pulls chains of
    that leg in submission order"""
    work = [float(p[:S].sum()) for p in passes_by_leg]
    end = 0
    lanes_total = 0
    for p, w in zip(passes_by_leg, work):
        n_waves = max(1, int(round(slots * w / sum(work))))
        lanes = min(n_waves * 64, S)
        lanes_total += lanes
        free = [0] * lanes
        heapq.heapify(free)
        for c in p[:S]:
            t0 = heapq.heappop(free)
            heapq.heappush(free, t0 + int(c))
        end = max(end, max(free))
    return end


def main():
    per_offset = chain_passes()
    rows = []
    for S in (4096, 32768, 65536, 131072, 262144, 524288):
        offs = (np.arange(S) * 11) % (6000 - T)
        legs = [per_offset["RF"][offs], per_offset["LF"][offs]]
        n_chains = 2 * S
        st, mean_wave, max_wave = static_makespan(legs, S, SLOTS)
        qu = queue_makespan(legs, S, SLOTS)
        total = int(sum(int(l.sum()) for l in legs))
        slowest = int(max(int(l.max()) for l in legs))
        floor = max(slowest, -(-total // (64 * SLOTS)))
        rows.append({"sequences": S, "chains": n_chains, "chains_per_lane_slot": n_chains / (64.0 * SLOTS),
                     "mean_lane_passes": total / n_chains, "mean_wave_passes_static": mean_wave, "slowest_wave_passes_static": max_wave,
                     "slowest_chain_passes": slowest, "makespan_passes": {"static": st, "queue": qu, "floor": floor},
                     "queue_speedup_over_static": st / qu})
    print(json.dumps({"workload": f"generic chain, RF + LF, windows of {T} frames of the shipped recording (scripts/bench_generic.py)",
                      "resident_wavefronts": SLOTS, "unit": "wave passes (one trial evaluation per lane and pass)",
                      "rows": rows,
                      "distinct_chains_evaluated_by_the_oracle": int(2 * (6000 - T))}, indent=1))


if __name__ == "__main__":
    main()
