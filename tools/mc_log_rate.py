#!/usr/bin/env python3
"""Logged Monte-Carlo ensemble rate: mc_example_1 parameters, 2048 runs x (1 + 200) steps, every run's
29-double log written.  chunk_steps = 1 is round 2's shape (one stepping launch + one record download per step),
64 = records written in-kernel by fused launches.  Prints arena-steps/s for both and checks the files are identical."""
import hashlib, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "robo-game-sim_amd"))
import numpy as np
from robo_game_py.montecarlo import monte_carlo_batched, mindless_policy


class P:                                         # py/mc_example_1.py:9-36
    num_runs, T, dt, winning_score = 2048, 10.0, 0.05, 4
    x0_ball = np.array([-1.0, 0.0, 0.0, 0.0])
    noise_stdev, tau_puck, tau_player, player_mass, puck_mass = 0.0, 0.1, 0.5, 1.0, 0.5
    log_prefix = ""


def digest(d):
    h = hashlib.sha256()
    for i in range(1, P.num_runs + 1):
        h.update(open(os.path.join(d, "mc_run_%d.log" % i), "rb").read())
    return h.hexdigest()


out = {}
for chunk in (1, 64):
    with tempfile.TemporaryDirectory() as d:
        P.log_prefix = d
        monte_carlo_batched(P, mindless_policy, write_logs=True, chunk_steps=chunk)      # warm (kernels, page cache)
        t0 = time.perf_counter()
        r = monte_carlo_batched(P, mindless_policy, write_logs=True, chunk_steps=chunk)
        el = time.perf_counter() - t0
        out[chunk] = (P.num_runs * r["steps"] / el, digest(d))
        print("chunk_steps %3d: %.3f s, %.4g logged arena-steps/s" % (chunk, el, out[chunk][0]), flush=True)
print("identical logs:", out[1][1] == out[64][1], " speed-up: %.1fx" % (out[64][0] / out[1][0]))
