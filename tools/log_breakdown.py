#!/usr/bin/env python3
"""Diagnostic: where the time of a logged Monte-Carlo ensemble goes (2048 runs x 201 steps, library log writer): handle,
reset, file creation, control tables, uploads, stepping + logging, close.  Run on the GPU box."""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "robo-game-sim_amd"))
import numpy as np
import robo_game_py as rgp
from robo_game_py.montecarlo import mindless_policy
B, N = 2048, 200
for rep in range(3):
    with tempfile.TemporaryDirectory() as d:
        T = {}
        def tick(name, t0): T[name] = T.get(name, 0.0) + time.perf_counter() - t0
        t0 = time.perf_counter(); sim = rgp.BatchedGameSim(B, 2); tick("create", t0)
        t0 = time.perf_counter(); sim.reset(0.05, 4, np.array([-1.0, 0, 0, 0]), 0.0, False, "", 1, 0.1, 0.5, 1.0, 0.5); tick("reset", t0)
        t0 = time.perf_counter(); sim.log_open(list(range(B)), [os.path.join(d, "mc_run_%d.log" % (r + 1)) for r in range(B)]); tick("log_open", t0)
        stride = 4 * 2 * B
        buf = sim.alloc(64 * stride)
        state = sim.state(); seeds = np.arange(B); t = 0.0
        for k0 in range(0, N, 64):
            k1 = min(N, k0 + 64)
            t0 = time.perf_counter()
            table = np.empty((k1 - k0, 4, 2, B))
            for k in range(k0, k1):
                table[k - k0] = mindless_policy(state, k, t, seeds); t += 0.05
            tick("table", t0)
            t0 = time.perf_counter(); buf.upload(table); tick("upload", t0)
            t0 = time.perf_counter(); sim.step_device_log(buf, stride, k1 - k0); tick("step_device_log", t0)
        t0 = time.perf_counter(); sim.log_close(); tick("log_close", t0)
        t0 = time.perf_counter(); sim.close(); tick("close", t0)
        print(rep, " ".join("%s %.1f ms" % (k, v * 1e3) for k, v in T.items()), "total %.1f ms" % (sum(T.values()) * 1e3), flush=True)
