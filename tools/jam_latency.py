#!/usr/bin/env python3
"""Diagnostic: latency of the event loop when a step runs into the 50-iteration cap.

Every arena starts with the puck inside player A1 (the C2 start that 4.5 % of the arenas have), so step 0
resolves 49 events per arena.  With few arenas (one wavefront per SIMD at most) the kernel time of that
step divided by 49 is the latency of one carried-out event + re-scan.  Needs a GPU; build the stamps
library (`make SUFFIX=_stamps STAMPS=1`) and set RGS_LIB_SUFFIX=_stamps for the per-phase cycle split.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "robo-game-sim_amd"))
import robo_game_py as rgp  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    rng = np.random.default_rng(5)
    x0 = np.stack([-2.5 + rng.uniform(-0.15, 0.15, B), 5.0 / 2 - 5.0 / (2 * n) + rng.uniform(-0.15, 0.15, B),
                   rng.uniform(-2, 2, B), rng.uniform(-2, 2, B)])
    ctl = np.ascontiguousarray(rng.uniform(-2, 2, (1, 2 * n, 2, B)))
    out = {}
    for name, variant in (("wave", 1), ("hybrid", 2), ("hybrid2", 3)):
        sim = rgp.BatchedGameSim(B, n)
        sim.set_variant(variant)
        buf = sim.alloc(ctl.size)
        buf.upload(ctl)
        res = []
        for rep in range(3):
            sim.reset(0.01, 1000000, x0)
            st = sim.stamps(True, clear=True)
            sim.sync()
            sim.timer_start()
            sim.step_device(buf.ptr, 0, 1)
            ms = sim.timer_stop()
            c = sim.counters()
            res.append(ms)
        st = sim.stamps(True)
        ev = max(1, c["resolved"]) / B
        out[name] = {"ms": [round(m, 4) for m in res], "events_per_arena": ev, "cap_hits": c["cap_hits"],
                     "us_per_event": round(min(res) * 1e3 / ev, 3), "stamps": {k: v for k, v in st.items() if v}}
        buf.free()
        sim.close()
    print(json.dumps({"arenas": B, "n": n, "results": out}, indent=1))


if __name__ == "__main__":
    main()
