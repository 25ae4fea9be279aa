#!/usr/bin/env python3
"""Logged Monte-Carlo ensemble rate: mc_example_1 parameters, 2048 runs x (1 + 200) steps, every run's
29-double log written, three ways: round 2's shape (one stepping launch + one record download per step, files written
from Python), fused launches with in-kernel records (files still from Python), and the library's own writer
(rgs_log_open / rgs_step_device_log).  Prints arena-steps/s for each and checks the files are identical."""
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


def python_writer(P, chunk):
    """Rounds 2 / 3a: records downloaded as (K, R, B) arrays, the files written from Python (chunk 1: one stepping launch and
    one download per step, round 2's shape; 64: fused launches with in-kernel records)."""
    import robo_game_py as rgp
    from robo_game_py.montecarlo import steps_per_run, write_configuration
    B = P.num_runs
    sim = rgp.BatchedGameSim(B, 2)
    write_configuration(P)
    sim.reset(P.dt, P.winning_score, np.asarray(P.x0_ball, dtype=np.float64), 0.0, False, "", 1, P.tau_puck, P.tau_player,
              P.player_mass, P.puck_mass)
    files = [open(os.path.join(P.log_prefix, "mc_run_%d.log" % r), "wb") for r in range(1, B + 1)]
    R = sim.record_size()

    def run(table):
        rec = sim.run_table_recorded(table, chunk_steps=chunk)
        per_run = np.ascontiguousarray(rec.transpose(2, 0, 1))
        ran = per_run[:, :, R - 1] != 0.0
        for q, f in enumerate(files):
            f.write(per_run[q, ran[q], :R - 1].tobytes())

    run(np.zeros((1, 4, 2, B)))
    N, t, state, seeds = steps_per_run(P), 0.0, sim.state(), np.arange(1, B + 1)
    for k0 in range(0, N, chunk):
        k1 = min(N, k0 + chunk)
        table = np.empty((k1 - k0, 4, 2, B))
        for k in range(k0, k1):
            table[k - k0] = mindless_policy(state, k, t, seeds)
            t += P.dt
        run(table)
    for f in files:
        f.close()
    sim.close()
    return dict(steps=N + 1)


out = {}
cases = [("round 2 shape: launch + download per step, files written from Python", lambda: python_writer(P, 1)),
         ("fused launches, in-kernel records, files written from Python", lambda: python_writer(P, 64)),
         ("fused launches, files written by the library (rgs_step_device_log)",
          lambda: monte_carlo_batched(P, mindless_policy, write_logs=True, chunk_steps=64))]
for name, fn in cases:
    with tempfile.TemporaryDirectory() as d:
        P.log_prefix = d
        fn()                                                                              # warm (kernels, page cache)
        t0 = time.perf_counter()
        r = fn()
        el = time.perf_counter() - t0
        out[name] = (P.num_runs * r["steps"] / el, digest(d))
        print("%-75s %.3f s, %.4g logged arena-steps/s" % (name + ":", el, out[name][0]), flush=True)
v = list(out.values())
print("identical logs:", v[0][1] == v[1][1] == v[2][1], " speed-up over round 2's shape: %.1fx (Python writer) %.1fx (library writer)"
      % (v[1][0] / v[0][0], v[2][0] / v[0][0]))
