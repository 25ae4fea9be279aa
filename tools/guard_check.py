#!/usr/bin/env python3
"""Debug run with guard zones around every device buffer (RGS_DEBUG_GUARD_BYTES=4096 python tools/guard_check.py): exercises the
round's new buffers -- dealing (perm / heat / lists), device noise generator (engine state, left-over normals), fused
log records, the log writer's buffers, trajectory snapshots -- on ragged batches and checks that no guard zone was written."""
import os, sys
assert os.environ.get("RGS_DEBUG_GUARD_BYTES"), "set RGS_DEBUG_GUARD_BYTES"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "robo-game-sim_amd"))
import numpy as np
import robo_game_py as rgp
from robo_game_py import workloads as w

for n, B in ((2, 8 * 256 + 37), (1, 777), (3, 1300), (4, 515), (7, 130)):
    x0, amp, ph = w.c2_inputs(B, n, seed=5 + n)
    K = 40
    table = w.circle_controls(amp * 2.0, ph, w.step_times(K, 0.05))
    for variant in (0, 1, 2, 3, 4, 5):
        sim = rgp.BatchedGameSim(B, n)
        sim.set_variant(variant)
        sim.reset(0.05, 3, x0)
        sim.run_table(table[:6], chunk_steps=6)
        sim.run_table(table[6:26], chunk_steps=20)          # a deal happens here for the pooled variants
        sim.run_table_traj(table[26:34], 4)
        sim.run_table_recorded(table[34:], chunk_steps=3)
        sim.check_guards()
        sim.close()
    sim = rgp.BatchedGameSim(B, n)
    sim.set_noise_mode("device")
    sim.reset(0.05, 3, x0, 0.2, False, "", 9)
    sim.run_table(table, chunk_steps=17)
    sim.run_table_recorded(table[:5], chunk_steps=5)
    import tempfile
    with tempfile.TemporaryDirectory() as d:                 # the library's log writer: records, per-file runs, ids, counts
        ids = [B - 1, 0, B // 2]
        sim.log_open(ids, [os.path.join(d, "a%d.log" % a) for a in ids])
        sim.run_table_logged(table[5:25], chunk_steps=7)
        sim.run_table_logged(table[25:], chunk_steps=64)     # a longer chunk: the buffers are re-allocated
        sim.check_guards()
        sim.log_close()
    sim.check_guards()
    sim.close()
    print("n=%d B=%d: guard zones intact" % (n, B), flush=True)
