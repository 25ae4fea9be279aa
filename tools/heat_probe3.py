#!/usr/bin/env python3
"""Can the state at the deal predict how long a hot arena stays hot?  (puck-player overlap depth vs event-pass steps in
the next 20 steps, driver shape)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "robo-game-sim_amd"))
import torch
import robo_game_py as rgp
from robo_game_py import workloads as w
W, K = 5, 20
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
st = sim.state()
sim.step_device(table.data_ptr() + 8 * W * stride, stride, 7); sim.sync()      # < 8 steps: no deal, heat accumulates
sim.step_device(table.data_ptr() + 8 * (W + 7) * stride, stride, 7); sim.sync()
sim.step_device(table.data_ptr() + 8 * (W + 14) * stride, stride, 6); sim.sync()
h2 = sim.heat().astype(np.int64) - h1
hot = h1 >= W
pk = st[2:4]
d = np.full(B, 9.0)
for e in range(2 * n):
    pl = st[6 + 4 * e:8 + 4 * e]
    d = np.minimum(d, np.sqrt(((pk - pl) ** 2).sum(0)))
depth = 0.375 - d                                  # > 0: puck overlaps a player
vrel = np.sqrt((st[4:6] ** 2).sum(0))
print("hot arenas %d; of them overlapping (depth > 0): %d" % (hot.sum(), (hot & (depth > 0)).sum()))
for lo, hi in ((-9, 0), (0, 0.02), (0.02, 0.05), (0.05, 0.1), (0.1, 0.2), (0.2, 0.4)):
    sel = hot & (depth > lo) & (depth <= hi)
    if sel.any():
        print("  depth (%.2f, %.2f]: %5d arenas, next-20 heat mean %.1f std %.1f" % (lo, hi, sel.sum(), h2[sel].mean(), h2[sel].std()))
print("corr(depth, h2 | hot) = %.3f" % np.corrcoef(depth[hot], h2[hot])[0, 1])
# how much better would a weighted deal be?  per-WG sum of h2 under: identity, count-levelled (simulated greedy), weight-levelled by depth classes
def spread(assign):
    s = np.zeros(256)
    np.add.at(s, assign, h2)
    return s.mean(), s.max(), s.std()
home = np.arange(B) // 256
print("identity: mean %.1f max %.0f std %.1f" % spread(home))
def level(weights_class):
    assign = home.copy()
    for cls in np.unique(weights_class[weights_class >= 0]):
        idx = np.where(weights_class == cls)[0]
        cnt = np.bincount(home[idx], minlength=256)
        q = len(idx) // 256
        give = [list(idx[home[idx] == wg][q + 1:]) for wg in range(256)]
        pool = [a for g in give for a in g]
        for wg in range(256):
            while cnt[wg] < q and pool:
                a = pool.pop(); cnt[home[a]] -= 1; assign[a] = wg; cnt[wg] += 1
    return assign
cls_count = np.where(hot, 0, -1)
print("count-levelled: mean %.1f max %.0f std %.1f" % spread(level(cls_count)))
cls_depth = np.where(hot, np.digitize(depth, [0.0, 0.05, 0.1, 0.2]), -1)
print("depth-class-levelled: mean %.1f max %.0f std %.1f" % spread(level(cls_depth)))
oracle_cls = np.where(hot, np.digitize(h2, [5, 10, 15, 19]), -1)
print("(oracle: levelled by the true next-20 heat classes): mean %.1f max %.0f std %.1f" % spread(level(oracle_cls)))
sim.close()
