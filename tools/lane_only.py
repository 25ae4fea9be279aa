#!/usr/bin/env python3
"""Diagnostic: what a 20-step fused launch of 65 536 2v2 arenas costs when NOTHING is ever flagged (pucks at rest away from
everybody, players commanded in small circles): the lane pass alone, per variant.  Compare with the driver's window
(bench.py --steps 20 --warmup 5).  usage: python tools/lane_only.py [arenas=65536] [steps=20] [dt=0.01]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "robo-game-sim_amd"))
import numpy as np
import torch
import robo_game_py as rgp
from robo_game_py import workloads as w

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dt = float(sys.argv[3]) if len(sys.argv) > 3 else 0.01
n = 2
x0 = np.zeros((4, B)); x0[1] = 2.0                               # puck at rest at (0, 2): 2 m from the nearest player's start
_, amp, ph = w.c2_inputs(B, n)
table = w.device_circle_table(torch, torch.device("cuda", 0), amp * 0.05, ph, w.step_times(5 + K, dt))
stride = 2 * n * 2 * B
for variant, name in ((5, "asynchronous pool"), (4, "pool"), (2, "one-lane hybrid")):
    sim = rgp.BatchedGameSim(B, n)
    sim.set_variant(variant)
    best = 1e9
    for rep in range(3):
        sim.reset(dt, 1000000, x0)
        sim.step_device(table.data_ptr(), stride, 5); sim.sync()
        sim.timer_start()
        sim.step_device(table.data_ptr() + 8 * 5 * stride, stride, K)
        best = min(best, sim.timer_stop())
    c = sim.counters()
    print("%-18s %d arenas x %d steps, nothing flagged (flagged %d, resolved %d): %.4f ms per launch, %.2f us per step, %.4g arena-steps/s"
          % (name, B, K, c["flagged"], c["resolved"], best, best * 1e3 / K, B * K / (best * 1e-3)), flush=True)
    sim.close()
