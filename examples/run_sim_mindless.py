#!/usr/bin/env python3
"""py/run_sim.py of the reference with its commented-out "mindless" controls (run_sim.py:62-65)
in place of the pydrake teams: one 2v2 game on the GPU through the drop-in GameSim.

    PYTHONPATH=robo-game-sim_amd python examples/run_sim_mindless.py [T] [dt]
"""
import sys
from math import cos, sin

import numpy as np

from robo_game_py import GameSim, SimStateView


class GameParams:                                   # py/run_sim.py:13-36
    def __init__(self, T=20.0, dt=0.05):
        self.T = T
        self.dt = dt
        self.winning_score = 4
        self.x0_ball = np.array([-1, 0., 0., 0.])
        self.log = True
        self.logname = "minimal_game.log"
        self.w_stdev = 0.


def main():
    params = GameParams(*(float(a) for a in sys.argv[1:3]))
    sim = GameSim()
    sim.reset(params.dt, params.winning_score, params.x0_ball, params.w_stdev, params.log, params.logname)   # run_sim.py:43
    sim_state = SimStateView(sim.run(np.zeros(2), np.zeros(2), np.zeros(2), np.zeros(2)))                    # :44
    t = 0.0
    while t < params.T:                             # :56-75
        vel = np.array([cos(t), sin(t)])
        sim_state = SimStateView(sim.run(vel, vel, vel, vel))
        t += params.dt
    print(sim_state.state)
    print("undecided:", sim.undecided(), " log:", params.logname)


if __name__ == "__main__":
    main()
