"""Synthetic, host-generated inputs for the batched stepper.

The reference's own drivers need pydrake policies (py/run_sim.py:47-69); its open-loop
"mindless" controls u = (cos t, sin t) (py/run_sim.py:62-65) are the model for these tables.
Everything is float64 numpy computed on the host: device sin/cos differ from the host's in
the last ulp and the dynamics amplify that, so parity runs always share one host-made table.
"""
import math

import numpy as np


def step_times(n_steps, dt, t0=0.0):
    """t_k accumulated the way the reference drivers do (`t += dt`, py/run_sim.py:75)."""
    t = np.empty(n_steps)
    acc = float(t0)
    for k in range(n_steps):
        t[k] = acc
        acc += dt
    return t


def c2_inputs(n_arenas, n_per_team=2, seed=20260):
    """Per-arena puck start and per-player amplitude / phase of BASELINE config C2/C3."""
    rng = np.random.default_rng(seed)
    P = 2 * n_per_team
    x0 = np.stack([rng.uniform(-3, 3, n_arenas), rng.uniform(-1.5, 1.5, n_arenas),
                   rng.uniform(-2, 2, n_arenas), rng.uniform(-2, 2, n_arenas)])
    amp = rng.uniform(0.5, 4.0, (P, n_arenas))
    phase = rng.uniform(0.0, 2.0 * math.pi, (P, n_arenas))
    return x0, amp, phase


def circle_controls(amp, phase, times):
    """u[k, p, :, a] = amp[p, a] * (cos(t_k + phase), sin(t_k + phase)); shape (K, P, 2, B)."""
    arg = times[:, None, None] + phase[None]
    return np.ascontiguousarray(np.stack([amp[None] * np.cos(arg), amp[None] * np.sin(arg)], axis=2))


def mindless_controls(n_players, n_arenas, times):
    """py/run_sim.py:62-65: every player is commanded (cos t, sin t)."""
    c, s = np.cos(times), np.sin(times)
    u = np.stack([c, s], axis=1)                      # (K, 2)
    return np.ascontiguousarray(np.broadcast_to(u[:, None, :, None], (len(times), n_players, 2, n_arenas)))


def random_hold_controls(n_players, n_arenas, n_steps, limit, hold=20, seed=7):
    """Piecewise-constant random inputs with |u_x|,|u_y| <= limit (input_limit, py/run_sim.py:23)."""
    rng = np.random.default_rng(seed)
    n_seg = (n_steps + hold - 1) // hold
    seg = rng.uniform(-limit, limit, (n_seg, n_players, 2, n_arenas))
    return np.ascontiguousarray(np.repeat(seg, hold, axis=0)[:n_steps])


def chase_controls(state, n_per_team, speed=4.0):
    """Closed loop: every player heads for the puck at `speed` (contact-rich play).  state (S, B)."""
    P = 2 * n_per_team
    B = state.shape[1]
    u = np.zeros((P, 2, B))
    for e in range(P):
        d = state[2:4] - state[6 + 4 * e:8 + 4 * e]
        nrm = np.sqrt(d[0] * d[0] + d[1] * d[1])
        nrm = np.where(nrm > 0, nrm, 1.0)
        u[e] = speed * d / nrm
    return u


def mc_inputs(seeds, n_per_team=2):
    """Per-run control generator of the Monte-Carlo configuration (BASELINE C4).

    py/MonteCarlo.py:23,29 hands run i the seed i; with noise = 0 that seed has no effect on the
    physics, so here it seeds the run's open-loop control generator instead: the run's per-player
    amplitude A ~ U(0.5, 4) and phase phi ~ U(0, 2 pi) come from np.random.default_rng(seed).
    Returns (amp, phase), each of shape (P, len(seeds)).
    """
    P = 2 * n_per_team
    seeds = np.asarray(seeds, dtype=np.int64)
    amp = np.empty((P, len(seeds)))
    phase = np.empty((P, len(seeds)))
    for k, s in enumerate(seeds):
        rng = np.random.default_rng(int(s))
        amp[:, k] = rng.uniform(0.5, 4.0, P)
        phase[:, k] = rng.uniform(0.0, 2.0 * math.pi, P)
    return amp, phase


def device_circle_table(torch, device, amp, phase, times, zero_first=False, chunk=64):
    """circle_controls evaluated on the GPU: (K, P, 2, B) float64 tensor resident in HBM.

    Device cos/sin differ from numpy's in the last ulp, so whoever re-computes a run driven by this
    table on the CPU must use the table's own values (bench.py downloads the sampled arenas'
    columns).  zero_first: step 0 commands nothing (py/MonteCarlo.py:33).
    """
    K = len(times)
    P, B = amp.shape
    table = torch.empty((K, P, 2, B), dtype=torch.float64, device=device)
    amp_d = torch.from_numpy(np.ascontiguousarray(amp)).to(device)
    ph_d = torch.from_numpy(np.ascontiguousarray(phase)).to(device)
    t_d = torch.from_numpy(np.ascontiguousarray(times)).to(device)
    for k0 in range(0, K, chunk):
        arg = t_d[k0:k0 + chunk, None, None] + ph_d[None]
        table[k0:k0 + chunk, :, 0, :] = amp_d[None] * torch.cos(arg)
        table[k0:k0 + chunk, :, 1, :] = amp_d[None] * torch.sin(arg)
    if zero_first and K > 0:
        table[0] = 0.0
    return table
