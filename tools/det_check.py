#!/usr/bin/env python3
"""Diagnostic: repeat one golden case on one variant and report which repetitions match (determinism check)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "robo-game-sim_amd"))
import robo_game_py as rgp
name, variant, reps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
n, table = int(g["n"]), g["table"]
K, P, _, B = table.shape
tp, tl, mp, mk = [float(v) for v in g["phys"]]
res = []
for r in range(reps):
    sim = rgp.BatchedGameSim(B, n)
    sim.set_variant(variant)
    sim.reset(float(g["dt"]), int(g["win"]), g["x0"], 0.0, False, "", 0, tp, tl, mp, mk)
    st = sim.run_table(table, chunk_steps=int(g["rec_every"]))
    sim.check_guards()
    dbg = sim.stamps(False)
    bad = np.argwhere(st != g["state"])
    res.append((len(bad), sorted(set(bad[:, 1].tolist())), [hex(v & (2**64 - 1)) for v in list(dbg.values())[:4]]))
    sim.close()
print(name, "variant", variant, "lib", os.environ.get("RGS_LIB_SUFFIX", ""), res)
