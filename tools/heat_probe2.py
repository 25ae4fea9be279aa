#!/usr/bin/env python3
"""After a deal: per-workgroup event-pass work (heat of the dealt launch) under the deal vs under the identity."""
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
sim.set_variant(5)
sim.reset(dt, 1000000, x0)
sim.step_device(table.data_ptr(), stride, W); sim.sync()
h1 = sim.heat().astype(np.int64)
sim.step_device(table.data_ptr() + 8 * W * stride, stride, K); sim.sync()
nsw, perm = sim.rebalance_info()
h2 = sim.heat().astype(np.int64)
print("dealt", nsw, "hot in window 1 (==W):", (h1 >= W).sum(), "window-2 heat total", h2.sum())
ident = h2.reshape(-1, 256).sum(1)
dealt = h2[perm].reshape(-1, 256).sum(1)
hot = h1 >= W
print("per-WG window-2 heat, identity: mean %.1f max %d min %d std %.1f" % (ident.mean(), ident.max(), ident.min(), ident.std()))
print("per-WG window-2 heat, dealt   : mean %.1f max %d min %d std %.1f" % (dealt.mean(), dealt.max(), dealt.min(), dealt.std()))
print("per-WG hot count identity: max %d min %d; dealt: max %d min %d" % (hot.reshape(-1, 256).sum(1).max(), hot.reshape(-1, 256).sum(1).min(), hot[perm].reshape(-1, 256).sum(1).max(), hot[perm].reshape(-1, 256).sum(1).min()))
print("window-2 heat by class: hot arenas %.3f of total; mean per hot arena %.2f; cold arenas with heat: %d" % (h2[hot].sum() / h2.sum(), h2[hot].mean(), (h2[~hot] > 0).sum()))
cold_wg = (h2 * (~hot))[perm].reshape(-1, 256).sum(1)
print("per-WG heat from COLD arenas (dealt): mean %.1f max %d std %.1f" % (cold_wg.mean(), cold_wg.max(), cold_wg.std()))
sim.close()
