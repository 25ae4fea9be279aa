#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the CPU oracle (oracle/gamesim_oracle.cpp).

PROVENANCE: these vectors are produced by this repository's own restatement of the reference
algorithm, NOT by the reference binary (which cannot be built here: it needs Eigen3).  They pin
the oracle against silent drift and give the GPU tests a second, frozen target.  The only vectors
that come from the compiled reference are the survey-recorded samples in survey_kats.json.

    python tests/golden/make_golden.py        # rewrites the .npz files next to this script
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "robo-game-sim_amd"))

from oracle import oracle as orc                      # noqa: E402
from robo_game_py import workloads as w               # noqa: E402


def scenario(name, n, B, K, dt, win, table, x0, rec_every, **phys):
    ref = orc.batch_run(n, x0, table, K, dt=dt, winning_score=win, rec_every=rec_every, n_threads=8,
                        want_damage=True, **phys)
    st = ref["stats"]
    np.savez_compressed(os.path.join(HERE, name + ".npz"), n=n, dt=dt, win=win, x0=x0,
                        table=table.astype(np.float64), rec_every=rec_every, state=ref["state"],
                        traj=ref["traj"], damage=ref["damage"],
                        stats=np.array([st["resolved"], st["cap_hits"], st["skipped_after_goal"], st["base_overflow"]]),
                        phys=np.array([phys.get("tau_puck", 0.1), phys.get("tau_player", 0.5),
                                       phys.get("player_mass", 1.0), phys.get("puck_mass", 0.5)]))
    print("%-28s arenas %4d steps %4d resolved %6d cap_hits %4d" % (name, B, K, st["resolved"], st["cap_hits"]))


def main():
    # C1-shaped: one 2v2 arena, mindless (cos t, sin t) controls, dt = 0.01 (py/run_sim.py:62-65)
    K = 2000
    scenario("c1_mindless_2v2", 2, 1, K, 0.01, 4, w.mindless_controls(4, 1, w.step_times(K, 0.01)),
             np.array([[-1.0], [0.0], [0.0], [0.0]]), 50)
    # C2-shaped: circles, dt = 0.01
    x0, amp, ph = w.c2_inputs(24, 2)                   # tables are stored (host cos/sin may differ by an ulp elsewhere)
    scenario("c2_circles_2v2", 2, 24, 300, 0.01, 1000000, w.circle_controls(amp * 1.5, ph, w.step_times(300, 0.01)), x0, 50)
    # collision-rich open loop at the reference's default dt
    x0, _, _ = w.c2_inputs(48, 2, seed=99)
    scenario("random_hold_2v2", 2, 48, 300, 0.05, 3, w.random_hold_controls(4, 48, 300, 10.0, hold=20, seed=7), x0, 50)
    # C3-shaped 7v7 and C5-shaped 4v4
    x0, amp, ph = w.c2_inputs(8, 7)
    scenario("c3_circles_7v7", 7, 8, 150, 0.01, 1000000, w.circle_controls(amp * 1.5, ph, w.step_times(150, 0.01)), x0, 50)
    x0, _, _ = w.c2_inputs(24, 4, seed=5)
    scenario("random_hold_4v4", 4, 24, 200, 0.05, 3, w.random_hold_controls(8, 24, 200, 10.0, hold=10, seed=3), x0, 50)
    # non-default physics (tau_puck = 1.0 is the "bounce_kick" setting of py/run_sim.py:19)
    x0, _, _ = w.c2_inputs(32, 2, seed=11)
    scenario("bounce_kick_params_2v2", 2, 32, 200, 0.02, 2, w.random_hold_controls(4, 32, 200, 8.0, hold=10, seed=2), x0,
             50, tau_puck=1.0, tau_player=0.3, player_mass=2.0, puck_mass=0.25)


def unit_scenarios():
    """six single-arena situations side by side, recorded after EVERY step."""
    K, dt, B = 80, 0.05, 6
    x0 = np.zeros((4, B))
    tab = np.zeros((K, 4, 2, B))
    # 0: free flight (no events): A1 commanded diagonally, puck coasting
    tab[:, 0, :, 0] = [1.0, -0.5]; x0[:, 0] = [0.0, 0.0, 0.8, 0.3]
    # 1: A1 into the top wall and held there (repeated wall events, overlap pushes)
    tab[:, 0, :, 1] = [0.0, 6.0]
    # 2: A1 head-on into a resting puck (player -> puck impulse, no damage)
    tab[:, 0, :, 2] = [4.0, 0.0]; x0[:, 2] = [-1.5, 1.25, 0.0, 0.0]
    # 3: A1 and B1 driven head-on (player <-> player, damage both ways)
    tab[:, 0, :, 3] = [6.0, 0.0]; tab[:, 2, :, 3] = [-6.0, 0.0]
    # 4: puck shot into the right goal mouth, then a second goal (scores, puck zeroing, later steps)
    x0[:, 4] = [3.0, 0.1, 9.0, 0.0]
    # 5: the puck squeezed between A1 and the left wall above the goal mouth (sustained contact: steps
    #    that end on the 50-iteration cap, include/GameSim.h:7)
    tab[:, 0, :, 5] = [-6.0, 0.0]; x0[:, 5] = [-4.6, 1.25, 0.0, 0.0]
    scenario("unit_scenarios_2v2", 2, B, K, dt, 2, tab, x0, 1)


if __name__ == "__main__":
    unit_scenarios()
    main()
