#!/usr/bin/env python3
"""A noisy Monte-Carlo ensemble at scale through the harness (SURVEY 8f-1 + 8f-3): mc_example_1 parameters with
noise_stdev = 0.1, 65 536 runs (run i seeded i, py/MonteCarlo.py:29) x (1 + 200) steps in ONE batch -- no 4096-run chunks:
the device generator is exact -- timed, and 64 runs spread over the ensemble replayed on the oracle and compared bit for bit.
usage: python tools/mc_noise_scale.py [runs=65536]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "robo-game-sim_amd"))
import numpy as np
from robo_game_py.montecarlo import monte_carlo_batched, mindless_policy, steps_per_run
from oracle import oracle as orc


class P:                                         # py/mc_example_1.py:9-36 with noise
    num_runs, T, dt, winning_score = int(sys.argv[1]) if len(sys.argv) > 1 else 65536, 10.0, 0.05, 4
    x0_ball = np.array([-1.0, 0.0, 0.0, 0.0])
    noise_stdev, tau_puck, tau_player, player_mass, puck_mass = 0.1, 0.1, 0.5, 1.0, 0.5
    log_prefix = "/tmp/mc_noise_scale"


monte_carlo_batched(P, mindless_policy, write_logs=False)                 # warm
t0 = time.perf_counter()
out = monte_carlo_batched(P, mindless_policy, write_logs=False)
el = time.perf_counter() - t0
N = steps_per_run(P)
print("%d noisy runs x %d steps in one batch: %.3f s, %.4g arena-steps/s (harness included); decided early: %d"
      % (P.num_runs, out["steps"], el, P.num_runs * out["steps"] / el, int((~out["undecided"]).sum())), flush=True)
bad = 0
picks = np.linspace(1, P.num_runs, 64).astype(int)
for r in picks:
    s = orc.OracleGameSim()
    s.reset(P.dt, P.winning_score, P.x0_ball, P.noise_stdev, False, "", int(r), P.tau_puck, P.tau_player, P.player_mass, P.puck_mass)
    st = s.run(np.zeros(2), np.zeros(2), np.zeros(2), np.zeros(2))
    t = 0.0
    for k in range(N):
        u = mindless_policy(st.reshape(22, 1), k, t, np.array([r]))
        st = s.run(np.ascontiguousarray(u[:, :, 0]))
        t += P.dt
    bad += not np.array_equal(out["state"][:, r - 1], st)
print("runs replayed on the oracle: %d, differing: %d" % (len(picks), bad))
sys.exit(1 if bad else 0)
