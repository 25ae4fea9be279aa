#!/usr/bin/env python3
"""Diagnostic: arena-steps/s of the noise > 0 paths.
usage: python tools/noise_rate.py [arenas=4096] [steps=20] [n_per_team=2] [mode=exact|device] [dt=0.05]
exact: host-generated draw tapes, one launch per step, <= 4096 arenas; device: generator on the GPU, one fused launch."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "robo-game-sim_amd"))
import robo_game_py as rgp  # noqa: E402
from robo_game_py import workloads as w  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
n = int(sys.argv[3]) if len(sys.argv) > 3 else 2
mode = sys.argv[4] if len(sys.argv) > 4 else "exact"
dt = float(sys.argv[5]) if len(sys.argv) > 5 else 0.05
x0, amp, ph = w.c2_inputs(B, n)
sim = rgp.BatchedGameSim(B, n)
sim.set_noise_mode(mode)
sim.reset(dt, 1000000, x0, 0.1, False, "", 1, 0.1, 0.5, 1.0, 0.5)
if mode == "device":
    import torch
    table = w.device_circle_table(torch, torch.device("cuda", 0), amp, ph, w.step_times(K + 2, dt))
    stride = 2 * n * 2 * B
    sim.step_device(table.data_ptr(), stride, 2); sim.sync()
    t = time.perf_counter()
    sim.step_device(table.data_ptr() + 8 * 2 * stride, stride, K); sim.sync()
else:
    table = w.circle_controls(amp, ph, w.step_times(K + 2, dt))
    sim.step(table[0]); sim.step(table[1])
    t = time.perf_counter()
    for k in range(K):
        sim.step(table[2 + k])
    sim.sync()
el = time.perf_counter() - t
print("noise path (%s): %d arenas x %d steps (%dv%d, dt %g): %.3f ms per step, %.3g arena-steps/s, events resolved %d"
      % (mode, B, K, n, n, dt, el / K * 1e3, B * K / el, sim.counters()["resolved"]))
sim.close()
