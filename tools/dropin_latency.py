#!/usr/bin/env python3
"""Diagnostic: wall time per run() of the one-arena drop-in GameSim (py/run_sim.py's loop shape)."""
import os, sys, time, math
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "robo-game-sim_amd"))
from robo_game_py import GameSim
sim = GameSim()
sim.reset(0.01, 4, np.array([-1.0, 0, 0, 0]), 0.0, False, "")
sim.run(np.zeros(2), np.zeros(2), np.zeros(2), np.zeros(2))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
t = 0.0
t0 = time.perf_counter()
for k in range(N):
    u = np.array([math.cos(t), math.sin(t)])
    sim.run(u, u, u, u)
    t += 0.01
el = time.perf_counter() - t0
print("drop-in GameSim: %d steps, %.1f us per run()" % (N, el / N * 1e6))
