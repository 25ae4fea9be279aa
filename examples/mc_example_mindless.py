#!/usr/bin/env python3
"""py/mc_example_1.py with mindless teams: the reference's sequential Monte-Carlo loop and the batched
one, both on the GPU stepper, writing configuration.txt + mc_run_<i>.log like py/MonteCarlo.py.

    PYTHONPATH=robo-game-sim_amd python examples/mc_example_mindless.py
"""
import math

import numpy as np

from robo_game_py.montecarlo import monte_carlo, monte_carlo_batched


class GameParams:                                   # py/mc_example_1.py:9-36
    def __init__(self, prefix, num_runs):
        self.log_prefix = prefix
        self.num_runs = num_runs
        self.noise_stdev = 0.0
        self.T = 10.0
        self.dt = 0.05
        self.tau_player = 0.5
        self.tau_puck = 0.1
        self.player_mass = 1.0
        self.puck_mass = 0.5
        self.winning_score = 4
        self.x0_ball = np.array([-1, 0., 0., 0.])


class MindlessTeam:
    def __init__(self, dt):
        self.t, self.dt = 0.0, dt

    def run(self, sim_state):                       # same shape as ClassicalTeam.run: -> (vel1, vel2)
        v = np.array([math.cos(self.t), math.sin(self.t)])
        self.t += self.dt / 2                       # called once per team per step
        return v, v


if __name__ == "__main__":
    p = GameParams("mindless_vs_mindless", 4)
    monte_carlo(p, MindlessTeam(p.dt).run, MindlessTeam(p.dt).run)
    big = GameParams("mindless_vs_mindless_batched", 4096)
    out = monte_carlo_batched(big, logged_runs=[1, 2])
    print("batched: %d runs, %d steps each, scores A/B: %g / %g" % (big.num_runs, out["steps"], out["scores"][0].sum(),
                                                                    out["scores"][1].sum()))
