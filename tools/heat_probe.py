#!/usr/bin/env python3
"""How well do an arena's event-pass steps in one launch predict the next launch?  (Input to the dealing rule of
rgs_hip.hip: rebalance.)  Runs the driver's shape: W warm-up steps, then K steps, dealing off, and prints the joint
distribution of per-arena heat in the two windows and the per-workgroup sums."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "robo-game-sim_amd"))
import torch
import robo_game_py as rgp
from robo_game_py import workloads as w
W, K = int(sys.argv[1]) if len(sys.argv) > 1 else 5, int(sys.argv[2]) if len(sys.argv) > 2 else 20
n, B, dt = 2, 65536, 0.01
x0, amp, ph = w.c2_inputs(B, n)
dev = torch.device("cuda", 0)
table = w.device_circle_table(torch, dev, amp, ph, w.step_times(W + K, dt))
stride = 2 * n * 2 * B
sim = rgp.BatchedGameSim(B, n)
sim.set_variant(5); sim.set_rebalance(True)
sim.reset(dt, 1000000, x0)
sim.step_device(table.data_ptr(), stride, W); sim.sync()
h1 = sim.heat().astype(np.int64)
sim.set_rebalance(False)                       # keeps counting, never deals
sim.set_rebalance(True)
sim.step_device(table.data_ptr() + 8 * W * stride, stride, 7); sim.sync()   # < 8 steps: no deal, heat accumulates
h2 = sim.heat().astype(np.int64) - h1
print("window 1 (%d steps): arenas with heat>=1: %d, >=3: %d, ==W: %d" % (W, (h1 >= 1).sum(), (h1 >= 3).sum(), (h1 >= W).sum()))
print("window 2 (7 steps): total heat %d, arenas>=1: %d" % (h2.sum(), (h2 >= 1).sum()))
for lo in range(0, W + 1):
    sel = h1 == lo
    print("  h1 == %d: %6d arenas, mean h2 %.2f, share of window-2 heat %.3f" % (lo, sel.sum(), h2[sel].mean() if sel.any() else 0, h2[sel].sum() / max(1, h2.sum())))
wg1, wg2 = h1.reshape(-1, 256).sum(1), h2.reshape(-1, 256).sum(1)
print("per-workgroup heat window 2: mean %.1f max %d min %d; corr with window 1: %.3f" % (wg2.mean(), wg2.max(), wg2.min(), np.corrcoef(wg1, wg2)[0, 1]))
sim.close()
