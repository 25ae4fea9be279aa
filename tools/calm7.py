#!/usr/bin/env python3
"""Diagnostic: a hybrid variant at n v n on an input where no arena is ever flagged (lane pass only)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "robo-game-sim_amd"))
import robo_game_py as rgp
n, variant, B, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
amp = float(sys.argv[5]) if len(sys.argv) > 5 else 0.01
P = 2 * n
x0 = np.zeros((4, B))
table = np.full((K, P, 2, B), amp)
sim = rgp.BatchedGameSim(B, n)
sim.set_variant(variant)
sim.reset(0.01, 1000000, x0)
st = sim.run_table(table, chunk_steps=K)
print("n", n, "variant", variant, "B", B, "K", K, "amp", amp, "counters", sim.counters(), "finite", bool(np.isfinite(st).all()))
sim.close()
