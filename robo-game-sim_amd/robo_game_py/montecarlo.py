"""Monte-Carlo harness on the HIP stepper (reference: py/MonteCarlo.py:8-49).

Two entry points:

monte_carlo(mc_params, team_A_run_method, team_B_run_method, cbf_run_method=None)
    Same signature, same call order and same files as the reference: one game after another, the
    (stateful) team objects are called exactly as py/MonteCarlo.py:36-45 calls them, every step runs
    on the GPU through the drop-in GameSim.  `py/mc_example_1.py`-style parameter bags work unchanged.

monte_carlo_batched(mc_params, policy, ...)
    All `num_runs` games advance together as one batch of arenas.  The policy must be vectorised
    over runs (it receives the (S, B) state array); the reference's per-run seed (`seed=i`,
    py/MonteCarlo.py:29) is available to it as `seeds`.  Writes the same configuration.txt and, for
    the runs asked for, the same binary logs.

Both write `configuration.txt` with the 13 scalars of py/MonteCarlo.py:14-21 and `mc_run_<i>.log`
files of 29 native-endian doubles per executed step (matlab/interpret_mc_results.m:32-48 reads them).
"""
import math
import os

import numpy as np

from . import BatchedGameSim, GameSim, SimStateView, log_record


def write_configuration(mc_params):
    """py/MonteCarlo.py:11-21: directory + configuration.txt, one `str(param)` per line."""
    if not os.path.exists(mc_params.log_prefix):
        os.makedirs(mc_params.log_prefix)
    x0 = mc_params.x0_ball
    values = [mc_params.num_runs, mc_params.T, mc_params.dt, mc_params.winning_score, x0[0], x0[1], x0[2], x0[3],
              mc_params.noise_stdev, mc_params.tau_puck, mc_params.tau_player, mc_params.player_mass,
              mc_params.puck_mass]
    with open(os.path.join(mc_params.log_prefix, "configuration.txt"), "w") as f:
        for v in values:
            f.write(str(v) + "\n")


def steps_per_run(mc_params):
    """N = ceil(T/dt) controller steps (py/MonteCarlo.py:34) after the one zero-input step (:33)."""
    return int(math.ceil(mc_params.T / mc_params.dt))


_StateView = SimStateView        # the public, N-generic look-alike of py/src/SimState.py


def monte_carlo(mc_params, team_A_run_method, team_B_run_method, cbf_run_method=None, state_class=None,
                sim=None, verbose=True):
    """Drop-in for py/MonteCarlo.py:8-49 (sequential games, reference call order).

    state_class: what the team methods receive (default: a SimState look-alike; pass the reference's
    own `src.SimState.SimState` when running inside its tree).  Returns the list of final states.
    """
    state_class = state_class or _StateView
    sim = sim or GameSim()
    write_configuration(mc_params)
    finals = []
    for i in range(1, mc_params.num_runs + 1):
        if verbose:
            print("MC RUN # {} / {}:".format(i, mc_params.num_runs))
        logname = os.path.join(mc_params.log_prefix, "mc_run_{}.log".format(i))
        sim.reset(mc_params.dt, mc_params.winning_score, mc_params.x0_ball, mc_params.noise_stdev, True, logname, i,
                  mc_params.tau_puck, mc_params.tau_player, mc_params.player_mass, mc_params.puck_mass)
        sim_state = state_class(sim.run(np.zeros(2), np.zeros(2), np.zeros(2), np.zeros(2)))
        for _ in range(steps_per_run(mc_params)):
            velA1, velA2 = team_A_run_method(sim_state)
            velB1, velB2 = team_B_run_method(sim_state)
            if cbf_run_method is not None:
                velA1, velA2, velB1, velB2 = cbf_run_method([velA1, velA2, velB1, velB2], sim_state)
            sim_state = state_class(sim.run(velA1, velA2, velB1, velB2))
        finals.append(np.array(sim_state.state, copy=True))
    sim.reset(mc_params.dt, mc_params.winning_score, mc_params.x0_ball)     # closes the last log
    if verbose:
        print("DONE!")
    return finals


def mindless_policy(state, k, t, seeds):
    """py/run_sim.py:62-65: every player is commanded (cos t, sin t)."""
    P = (state.shape[0] - 6) // 4
    u = np.empty((P, 2, state.shape[1]))
    u[:, 0, :] = math.cos(t)
    u[:, 1, :] = math.sin(t)
    return u


# A policy whose commands do not depend on the state's VALUES (only on k, t, seeds and the array shapes) may say so:
# monte_carlo_batched then asks it for a whole chunk of steps at once and runs the chunk as ONE fused launch, log
# records included (the stepping kernels write them in-kernel), instead of one launch and one download per step.
mindless_policy.open_loop = True


NOISE_MAX_ARENAS = 4096      # include/rgs_hip.h: noise != 0 from host-generated draw tapes: <= 4096 arenas per handle


def monte_carlo_batched(mc_params, policy=mindless_policy, n_per_team=2, device=0, write_logs=True, logged_runs=None,
                        sim=None, _first_run=1, _write_config=True, chunk_steps=64):
    """All runs of the ensemble at once.

    policy(state (S, B), k, t, seeds (B,)) -> controls (P, 2, B); called once per step for the whole batch
    (k = step index, t = k*dt accumulated like py/run_sim.py:75).  logged_runs: iterable of 1-based run
    numbers whose mc_run_<i>.log to write (default: all when write_logs).
    A policy marked `open_loop = True` (its commands ignore the state's values: mindless_policy) is evaluated for
    chunk_steps steps ahead and each chunk runs as one fused launch, log records included; any other policy is closed
    loop: one step, one state download per step, like the reference (py/MonteCarlo.py:36-45).
    With noise_stdev != 0 the draws come from the device generator where that is exact (BatchedGameSim.noise_is_exact():
    any ensemble size, fused launches); otherwise from host-generated tapes, and an ensemble of more than 4096 runs is
    stepped in chunks of 4096 (one handle each, the tapes' limit); run i keeps seed i either way.
    Returns dict(state (S, B), scores (2, B), damage (P, P, B), undecided (B,), steps).
    """
    B = int(mc_params.num_runs)
    if float(mc_params.noise_stdev) != 0.0 and B > NOISE_MAX_ARENAS and sim is None and not BatchedGameSim.noise_is_exact():
        import copy
        if _write_config:
            write_configuration(mc_params)
        parts = []
        for lo in range(0, B, NOISE_MAX_ARENAS):
            sub = copy.copy(mc_params)
            sub.num_runs = min(NOISE_MAX_ARENAS, B - lo)
            parts.append(monte_carlo_batched(sub, policy, n_per_team, device, write_logs, logged_runs, None,
                                             _first_run=_first_run + lo, _write_config=False, chunk_steps=chunk_steps))
        return dict(state=np.concatenate([q["state"] for q in parts], axis=1),
                    scores=np.concatenate([q["scores"] for q in parts], axis=1),
                    damage=np.concatenate([q["damage"] for q in parts], axis=2),
                    undecided=np.concatenate([q["undecided"] for q in parts]), steps=parts[0]["steps"])
    own = sim is None
    sim = sim or BatchedGameSim(B, n_per_team, device)
    P = sim.players
    seeds = np.arange(_first_run, _first_run + B)                          # py/MonteCarlo.py:23,29
    if _write_config:
        write_configuration(mc_params)
    # run i of the reference uses seed = i (py/MonteCarlo.py:29); arena a draws from seed + a, so seed = first run
    sim.reset(mc_params.dt, mc_params.winning_score, np.asarray(mc_params.x0_ball, dtype=np.float64),
              float(mc_params.noise_stdev), False, "", int(_first_run), mc_params.tau_puck, mc_params.tau_player,
              mc_params.player_mass, mc_params.puck_mass)
    N = steps_per_run(mc_params)
    mine = range(_first_run, _first_run + B)
    runs = list(mine) if logged_runs is None else [int(r) for r in logged_runs if int(r) in mine]
    # the log files are written by the library (rgs_log_open / rgs_step_device_log): per-file gather on the device, the
    # writes on host threads; a decided game logs nothing (cpp:121-123)
    logging = bool(write_logs and runs)
    if logging:
        sim.log_open([r - _first_run for r in runs],
                     [os.path.join(mc_params.log_prefix, "mc_run_{}.log".format(r)) for r in runs])

    def run(table):                                                         # table (K, P, 2, B): K steps, logged if asked for
        if logging:
            sim.run_table_logged(table, chunk_steps=max(1, chunk_steps))    # fused launches
        elif len(table) == 1:
            sim.step(table[0])
        else:
            sim.run_table(table, chunk_steps=max(1, chunk_steps))

    run(np.zeros((1, P, 2, B)))                                             # py/MonteCarlo.py:33
    t = 0.0
    if getattr(policy, "open_loop", False):
        state = sim.state()                                                 # shapes only; an open-loop policy ignores the values
        for k0 in range(0, N, max(1, chunk_steps)):
            k1 = min(N, k0 + max(1, chunk_steps))
            table = np.empty((k1 - k0, P, 2, B))
            for k in range(k0, k1):
                table[k - k0] = policy(state, k, t, seeds)
                t += mc_params.dt
            run(table)
    else:
        for k in range(N):
            state = sim.state()
            run(np.ascontiguousarray(policy(state, k, t, seeds), dtype=np.float64)[None])
            t += mc_params.dt
    if logging:
        sim.log_close()
    state = sim.state()
    out = dict(state=state, scores=state[:2].copy(), damage=sim.damage(), undecided=sim.undecided(), steps=N + 1)
    if own:
        sim.close()
    return out


# ---------------------------------------------------------------------------------------------------
# Closed loop entirely on the device
# ---------------------------------------------------------------------------------------------------

def chase_policy_torch(state, k, t, speed=4.0):
    """Every player heads for the puck at `speed` (the contact-rich test policy), on the GPU.
    state: torch (S, B) float64 on the stepper's device -> controls (P, 2, B)."""
    import torch
    P = (state.shape[0] - 6) // 4
    players = state[6:6 + 4 * P].reshape(P, 4, -1)[:, :2]               # (P, 2, B)
    d = state[2:4].unsqueeze(0) - players
    nrm = torch.sqrt((d * d).sum(1, keepdim=True)).clamp_min(1e-300)
    return speed * d / nrm


def mindless_policy_torch(state, k, t):
    import torch
    P = (state.shape[0] - 6) // 4
    u = torch.empty((P, 2, state.shape[1]), dtype=torch.float64, device=state.device)
    u[:, 0] = math.cos(t)
    u[:, 1] = math.sin(t)
    return u


def run_closed_loop_on_device(sim, policy, n_steps, dt=None, t0=0.0):
    """state -> policy -> step, n_steps times, without leaving the GPU.

    sim: a BatchedGameSim already reset; policy(state, k, t) maps a torch (S, B) float64 tensor on the
    sim's device to a (P, 2, B) tensor.  Per step: one device-to-device copy of the state into the torch
    tensor, the policy's own kernels, one stepping launch reading the controls in place.  This is the
    C5-shaped loop (controller output = one (2,) vector per player per step, py/test_dmpc.py:43-47)
    with the controller on the device.  Returns the final state as a torch tensor.
    """
    import torch
    dev = torch.device("cuda", sim.device)
    dt = sim.params["dt"] if dt is None else dt
    state = torch.empty((sim.state_size, sim.n_arenas), dtype=torch.float64, device=dev)
    t = float(t0)
    for k in range(n_steps):
        sim.copy_state_to(state.data_ptr())
        sim.sync()                                   # the stepper's stream -> visible to torch's stream
        u = policy(state, k, t).contiguous()
        torch.cuda.current_stream(dev).synchronize() # torch's stream -> visible to the stepper's stream
        sim.step_device(u.data_ptr(), 0, 1)
        t += dt
    sim.copy_state_to(state.data_ptr())
    sim.sync()
    return state
