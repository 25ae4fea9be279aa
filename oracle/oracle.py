"""ctypes front-end for the CPU oracle (oracle/gamesim_oracle.cpp).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package (robo-game-sim_amd/) never
imports this module.

`OracleGameSim` mirrors the reference binding (src/RoboGamePy.cpp:11-25):
reset / undecided / run with the same argument meaning, plus n-per-team and a
few inspection helpers.  `batch_run` drives many arenas with the device-side
array layouts so the HIP stepper can be compared buffer-for-buffer.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libgamesim_oracle.so")
_lib = None

_dp = C.POINTER(C.c_double)
_llp = C.POINTER(C.c_longlong)


def build(force=False):
    """Compile the oracle with gcc (no GPU, no reference sources involved)."""
    src = os.path.join(_HERE, "gamesim_oracle.cpp")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libgamesim_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.rgso_create.restype = C.c_void_p
        L.rgso_create.argtypes = [C.c_int, C.c_int]
        L.rgso_destroy.argtypes = [C.c_void_p]
        L.rgso_state_size.argtypes = [C.c_void_p]
        L.rgso_reset.argtypes = [C.c_void_p, C.c_double, C.c_int, _dp, C.c_double, C.c_int, C.c_char_p,
                                 C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, _dp]
        L.rgso_undecided.argtypes = [C.c_void_p]
        L.rgso_run.argtypes = [C.c_void_p, _dp, _dp]
        L.rgso_get_damage.argtypes = [C.c_void_p, _dp]
        L.rgso_get_time.argtypes = [C.c_void_p]
        L.rgso_get_time.restype = C.c_double
        L.rgso_set_state.argtypes = [C.c_void_p, _dp, C.c_double]
        L.rgso_get_stats.argtypes = [C.c_void_p, _llp]
        L.rgso_batch_run.argtypes = [C.c_int, C.c_longlong, C.c_longlong, C.c_longlong, C.c_int,
                                     C.c_double, C.c_int, C.c_double, C.c_int, C.c_double, C.c_double,
                                     C.c_double, C.c_double, _dp, _dp, C.c_longlong, _dp, C.c_int, _dp,
                                     _dp, _llp, C.c_int, C.c_int]
        L.rgso_batch_run.restype = C.c_int
        L.rgso_rng_streams.argtypes = [C.c_longlong, C.c_longlong, _dp, C.c_longlong, _dp]
        L.rgso_rng_streams.restype = None
        _lib = L
    return _lib


def _ptr(a):
    return a.ctypes.data_as(_dp)


STAT_NAMES = ["steps", "resolved", "cap_hits", "base_overflow", "skipped_after_goal",
              "steps_with_events", "h0", "h1", "h2", "h3", "h4_7", "h8_15", "h16_31", "h32p"]


class OracleGameSim:
    """Single-arena oracle with the reference's GameSim call surface."""

    def __init__(self, n_per_team=2, draw_always=False):
        self._L = lib()
        self._h = self._L.rgso_create(int(n_per_team), int(bool(draw_always)))
        if not self._h:
            raise ValueError("n_per_team must be in 1..7")
        self.n_per_team = int(n_per_team)
        self.players = 2 * self.n_per_team
        self.state_size = self._L.rgso_state_size(self._h)

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.rgso_destroy(self._h)
            self._h = None

    def reset(self, dt=0.05, winning_score=3, x0_ball=None, noise=0.0, log=False,
              logname="~/gamelog.log", seed=0, tau_puck=0.1, tau_player=0.5,
              player_mass=1.0, puck_mass=0.5):
        x0 = np.zeros(4) if x0_ball is None else np.ascontiguousarray(x0_ball, dtype=np.float64)
        out = np.empty(self.state_size)
        self._L.rgso_reset(self._h, dt, int(winning_score), _ptr(x0), noise, int(bool(log)),
                           str(logname).encode(), int(seed), tau_puck, tau_player, player_mass,
                           puck_mass, _ptr(out))
        return out

    def undecided(self):
        return bool(self._L.rgso_undecided(self._h))

    def run(self, *vels):
        if len(vels) == 1:
            u = np.ascontiguousarray(vels[0], dtype=np.float64).reshape(-1)
        else:
            u = np.concatenate([np.asarray(v, dtype=np.float64).reshape(2) for v in vels])
        assert u.size == 2 * self.players
        out = np.empty(self.state_size)
        self._L.rgso_run(self._h, _ptr(u), _ptr(out))
        return out

    def damage(self):
        out = np.empty((self.players, self.players))
        self._L.rgso_get_damage(self._h, _ptr(out))
        return out

    def time(self):
        return self._L.rgso_get_time(self._h)

    def set_state(self, state, t=0.0):
        s = np.ascontiguousarray(state, dtype=np.float64)
        assert s.size == self.state_size
        self._L.rgso_set_state(self._h, _ptr(s), float(t))

    def stats(self):
        out = (C.c_longlong * 14)()
        self._L.rgso_get_stats(self._h, out)
        return dict(zip(STAT_NAMES, list(out)))


def batch_run(n_per_team, x0_ball, controls, n_steps, dt=0.05, winning_score=3, noise=0.0,
              seed_base=0, tau_puck=0.1, tau_player=0.5, player_mass=1.0, puck_mass=0.5,
              rec_every=0, n_threads=1, draw_always=False, a_begin=0, a_end=None,
              want_damage=False):
    """Run `n_steps` steps on every arena from a fresh reset.

    x0_ball  : [4, B]; controls: [n_steps, P, 2, B] or [P, 2, B] (held for all steps).
    Returns dict(state=[S,B], traj=[n_rec,S,B] or None, damage=[P*P,B] or None, stats=dict).
    """
    L = lib()
    P = 2 * n_per_team
    S = 2 + 4 * (P + 1)
    x0 = np.ascontiguousarray(x0_ball, dtype=np.float64)
    B = x0.shape[1]
    ctl = np.ascontiguousarray(controls, dtype=np.float64)
    if ctl.ndim == 3:
        assert ctl.shape == (P, 2, B)
        stride = 0
    else:
        assert ctl.shape == (n_steps, P, 2, B), ctl.shape
        stride = P * 2 * B
    if a_end is None:
        a_end = B
    state = np.zeros((S, B))
    traj = None
    if rec_every and rec_every > 0:
        traj = np.zeros((n_steps // rec_every, S, B))
    dmg = np.zeros((P * P, B)) if want_damage else None
    stats = (C.c_longlong * 14)()
    rc = L.rgso_batch_run(n_per_team, B, a_begin, a_end, n_steps, dt, winning_score, noise, seed_base,
                          tau_puck, tau_player, player_mass, puck_mass, _ptr(x0), _ptr(ctl), stride,
                          _ptr(state), int(rec_every or 0), _ptr(traj) if traj is not None else None,
                          _ptr(dmg) if dmg is not None else None, stats, int(n_threads),
                          int(bool(draw_always)))
    if rc != 0:
        raise RuntimeError("rgso_batch_run failed: %d" % rc)
    return dict(state=state, traj=traj, damage=dmg, stats=dict(zip(STAT_NAMES, list(stats))))


def rng_streams(seed, n_normals, n_canonicals):
    """The host's <random> streams for `seed`: (normals, canonical uniforms), as the reference's generator draws them."""
    normals, canon = np.empty(n_normals), np.empty(n_canonicals)
    lib().rgso_rng_streams(int(seed), n_normals, _ptr(normals), n_canonicals, _ptr(canon))
    return normals, canon
