"""robo_game_py -- MI355X-native drop-in for the reference's pybind11 module of the same name.

The reference builds `robo_game_py` from src/RoboGamePy.cpp:11-25 and its callers do
`from robo_game_py import GameSim` (py/run_sim.py:3, py/MonteCarlo.py:1).  Put this
directory's parent (`robo-game-sim_amd/`) on PYTHONPATH and the same import resolves here:

    GameSim          one arena; reset / undecided / run / repr exactly as the binding
                     (include/GameSim.h:21-27), every step executed by the HIP stepper.
    BatchedGameSim   many independent arenas resident on one GPU (SoA, arena index fastest).

All physics runs in librgs_hip.so (hand-written HIP for gfx950) through the C ABI of
include/rgs_hip.h.  There is no CPU fallback.
"""
import ctypes as C
import os
import struct

import numpy as np

from . import _lib
from ._lib import RgsError, VARIANT_APOOL, VARIANT_AUTO, VARIANT_HYBRID, VARIANT_HYBRID2, VARIANT_POOL, VARIANT_WAVE  # noqa: F401

__all__ = ["GameSim", "BatchedGameSim", "SimStateView", "RgsError", "reset_arguments", "state_size",
           "log_record"]

# C++ defaults of GameSim::reset, src/GameSim.cpp:21-25 (the binding itself declares none, so the
# reference needs all 11 positionals; py/run_sim.py:43 passes 6 -- both arities are accepted here).
_RESET_NAMES = ("dt", "winning_score", "x0_ball", "noise", "log", "logname", "seed", "tau_puck",
                "tau_player", "player_mass", "puck_mass")
_RESET_DEFAULTS = (0.05, 3, None, 0.0, False, "~/gamelog.log", 0, 0.1, 0.5, 1.0, 0.5)


def state_size(n_per_team=2):
    """SimState::SIZE (include/SimState.h:22): 2 scores + 4 doubles per disc."""
    return 2 + 4 * (2 * n_per_team + 1)


def reset_arguments(*args, **kwargs):
    """Normalise reset(...) arguments to the full 11-tuple of include/GameSim.h:21-24."""
    if len(args) > len(_RESET_NAMES):
        raise TypeError("reset() takes at most %d arguments (%d given)" % (len(_RESET_NAMES), len(args)))
    vals = dict(zip(_RESET_NAMES, _RESET_DEFAULTS))
    vals.update(zip(_RESET_NAMES, args))
    for k, v in kwargs.items():
        if k not in vals:
            raise TypeError("reset() got an unexpected keyword argument %r" % k)
        if k in _RESET_NAMES[:len(args)]:
            raise TypeError("reset() got multiple values for argument %r" % k)
        vals[k] = v
    out = dict(vals)
    out["dt"] = float(out["dt"])
    out["winning_score"] = int(out["winning_score"])
    out["noise"] = float(out["noise"])
    out["log"] = bool(out["log"])
    out["logname"] = str(out["logname"])
    out["seed"] = int(out["seed"])
    for k in ("tau_puck", "tau_player", "player_mass", "puck_mass"):
        out[k] = float(out[k])
    return out


def log_record(t, state, damage):
    """One record of the reference's binary log (src/GameSim.cpp:161-170, include/utils.h:64-76).

    3 + 2 + 2P + P*P native-endian doubles (29 at 2v2): t, TAS, TBS, puck xy, each player's xy,
    then the reference's `damage.row(k)` blocks.  logVectors writes rows*cols CONTIGUOUS doubles
    starting at &damage(k, 0) of a column-major matrix, so block k holds flat elements k..k+P-1
    of the column-major array, not row k; that byte layout is reproduced as is.
    """
    state = np.asarray(state, dtype=np.float64)
    damage = np.asarray(damage, dtype=np.float64)
    P = damage.shape[0]
    rec = [float(t), state[0], state[1], state[2], state[3]]
    for e in range(P):
        rec += [state[6 + 4 * e], state[7 + 4 * e]]
    flat = damage.flatten(order="F")
    for k in range(P):
        rec += list(flat[k:k + P])
    return struct.pack("=%dd" % len(rec), *rec)


class SimStateView:
    """N-generic counterpart of py/src/SimState.py:4-43: index constants over one arena's state vector plus
    the reference's accessors (get_player_pos / vel / state, get_puck_pos / vel / state, transpose), each
    returning a copy like the reference's deepcopy.  At n = 2 the constants are the reference's
    (TAS 0, TBS 1, PK 2, A1 6, A2 10, B1 14, B2 18); for other team sizes A<k> / B<k> exist for k = 1..n.
    Policies written against the reference class (ClassicalTeam.run & co.) take this one unchanged.
    """

    def __init__(self, state, n_per_team=None):
        self.state = np.array(state, dtype=np.float64, copy=True)         # copy.deepcopy(state), SimState.py:13
        size = self.state.shape[0]
        n = (size - 6) // 8 if n_per_team is None else int(n_per_team)
        if size != 2 + 4 * (2 * n + 1):
            raise TypeError("a %dv%d state has %d entries, got %d" % (n, n, 2 + 4 * (2 * n + 1), size))
        self.n = n
        self.TAS, self.TBS, self.PK = 0, 1, 2
        for k in range(1, n + 1):
            setattr(self, "A%d" % k, self.offset("A", k))
            setattr(self, "B%d" % k, self.offset("B", k))

    def offset(self, team, player_id):
        if team not in ("A", "B") or not 1 <= int(player_id) <= self.n:
            raise ValueError("team = 'A' or 'B', player_id = 1..%d (got %r, %r)" % (self.n, team, player_id))
        return 6 + 4 * ({"A": 0, "B": self.n}[team] + int(player_id) - 1)

    def get_player_state(self, team, player_id):
        o = self.offset(team, player_id)
        return self.state[o:o + 4].copy()

    def get_player_pos(self, team, player_id):
        return self.get_player_state(team, player_id)[0:2]

    def get_player_vel(self, team, player_id):
        return self.get_player_state(team, player_id)[2:4]

    def get_puck_state(self):
        return self.state[2:6].copy()

    def get_puck_pos(self):
        return self.state[2:4].copy()

    def get_puck_vel(self):
        return self.state[4:6].copy()

    def transpose(self):
        return self.state.transpose().copy()


class _DeviceBuffer:
    """Doubles resident in HBM, owned by a BatchedGameSim."""

    def __init__(self, sim, n):
        self._sim = sim
        self.n = int(n)
        p = C.c_void_p()
        _lib.check(_lib.lib().rgs_device_alloc(sim._h, self.n, C.byref(p)))
        self.ptr = p.value

    def upload(self, host, offset=0):
        a = np.ascontiguousarray(host, dtype=np.float64)
        assert offset + a.size <= self.n
        _lib.check(_lib.lib().rgs_device_upload(self._sim._h, C.c_void_p(self.ptr + 8 * offset), _lib.dptr(a), a.size))

    def download(self, n=None, offset=0):
        n = self.n - offset if n is None else int(n)
        out = np.empty(n)
        _lib.check(_lib.lib().rgs_device_download(self._sim._h, _lib.dptr(out), C.c_void_p(self.ptr + 8 * offset), n))
        return out

    def free(self):
        if self.ptr:
            _lib.check(_lib.lib().rgs_device_free(self._sim._h, C.c_void_p(self.ptr)))
            self.ptr = None


class BatchedGameSim:
    """`n_arenas` independent games on one MI355X.

    Arrays follow the device layout, arena index fastest:
    state (S, B), controls (P, 2, B), x0_ball (4,) or (4, B), damage (P, P, B), time (B,).
    """

    def __init__(self, n_arenas, n_per_team=2, device=0):
        self._L = _lib.lib()
        self._h = None
        h = C.c_void_p()
        _lib.check(self._L.rgs_create(int(n_arenas), int(n_per_team), int(device), C.byref(h)))
        self._h = h
        self.n_arenas = int(n_arenas)
        self.n_per_team = int(n_per_team)
        self.players = 2 * self.n_per_team
        self.state_size = state_size(self.n_per_team)
        self.device = int(device)
        self.params = reset_arguments()
        self._pinned = []

    def close(self):
        if getattr(self, "_h", None):
            self._L.rgs_destroy(self._h)
            self._h = None
            for p in getattr(self, "_pinned", []):
                self._L.rgs_host_free(C.c_void_p(p))
            self._pinned = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- reference surface, batched -------------------------------------------------------
    def reset(self, *args, **kwargs):
        """GameSim::reset for every arena; x0_ball may be (4,) or (4, B).  Returns state (S, B)."""
        a = reset_arguments(*args, **kwargs)
        x0 = np.zeros(4) if a["x0_ball"] is None else np.ascontiguousarray(a["x0_ball"], dtype=np.float64)
        if x0.shape == (4,):
            batched = 0
        elif x0.shape == (4, self.n_arenas):
            batched = 1
        else:
            raise TypeError("x0_ball must have shape (4,) or (4, n_arenas), got %r" % (x0.shape,))
        _lib.check(self._L.rgs_reset(self._h, a["dt"], a["winning_score"], _lib.dptr(x0), batched, a["noise"],
                                     a["seed"], a["tau_puck"], a["tau_player"], a["player_mass"], a["puck_mass"]))
        self.params = a
        return self.state()

    def undecided(self):
        out = np.empty(self.n_arenas, dtype=np.uint8)
        _lib.check(self._L.rgs_undecided(self._h, out.ctypes.data_as(C.POINTER(C.c_uint8))))
        return out.astype(bool)

    def run(self, controls):
        """GameSim::run for every arena with host controls (P, 2, B); returns state (S, B)."""
        self.step(controls)
        return self.state()

    # -- batched extras -------------------------------------------------------------------
    def step(self, controls):
        c = np.ascontiguousarray(controls, dtype=np.float64)
        if c.shape != (self.players, 2, self.n_arenas):
            raise TypeError("controls must have shape (%d, 2, %d), got %r" % (self.players, self.n_arenas, c.shape))
        _lib.check(self._L.rgs_step(self._h, _lib.dptr(c)))

    def step_device(self, ctl_ptr, step_stride, n_steps):
        """Asynchronous: n_steps steps reading controls from device memory (ints are raw pointers)."""
        ptr = ctl_ptr.ptr if isinstance(ctl_ptr, _DeviceBuffer) else int(ctl_ptr)
        _lib.check(self._L.rgs_step_device(self._h, C.c_void_p(ptr), int(step_stride), int(n_steps)))

    def run_table(self, table, chunk_steps=256):
        """Step through a host control table (K, P, 2, B), uploaded in chunks; returns state (S, B)."""
        t = np.ascontiguousarray(table, dtype=np.float64)
        K = t.shape[0]
        if t.shape[1:] != (self.players, 2, self.n_arenas):
            raise TypeError("table must have shape (K, %d, 2, %d)" % (self.players, self.n_arenas))
        stride = self.players * 2 * self.n_arenas
        buf = self.alloc(min(K, chunk_steps) * stride)
        try:
            for k0 in range(0, K, chunk_steps):
                k1 = min(K, k0 + chunk_steps)
                buf.upload(t[k0:k1])
                self.step_device(buf, stride, k1 - k0)
                self.sync()
        finally:
            buf.free()
        return self.state()

    def run_table_traj(self, table, rec_every):
        """Step through a host control table (K, P, 2, B) in ONE fused launch and return the state after
        every rec_every-th step: array (K // rec_every, S, B)."""
        t = np.ascontiguousarray(table, dtype=np.float64)
        K = t.shape[0]
        if t.shape[1:] != (self.players, 2, self.n_arenas):
            raise TypeError("table must have shape (K, %d, 2, %d)" % (self.players, self.n_arenas))
        n_rec = K // int(rec_every)
        stride = self.players * 2 * self.n_arenas
        cbuf = self.alloc(K * stride)
        tbuf = self.alloc(max(1, n_rec) * self.state_size * self.n_arenas)
        try:
            cbuf.upload(t)
            _lib.check(self._L.rgs_step_device_traj(self._h, C.c_void_p(cbuf.ptr), stride, K, int(rec_every),
                                                    C.c_void_p(tbuf.ptr)))
            out = tbuf.download(n_rec * self.state_size * self.n_arenas).reshape(n_rec, self.state_size, self.n_arenas)
        finally:
            cbuf.free()
            tbuf.free()
        return out

    def record_size(self):
        return int(self._L.rgs_record_size(self.n_per_team))

    def run_table_recorded(self, table, chunk_steps=64):
        """Step through a host control table (K, P, 2, B) and return the reference's log record of every
        step for every arena: array (K, R, B), R = record_size(); row R-1 is 1.0 where the step executed."""
        t = np.ascontiguousarray(table, dtype=np.float64)
        K = t.shape[0]
        if t.shape[1:] != (self.players, 2, self.n_arenas):
            raise TypeError("table must have shape (K, %d, 2, %d)" % (self.players, self.n_arenas))
        R, B = self.record_size(), self.n_arenas
        stride = self.players * 2 * B
        out = np.empty((K, R, B))
        ch = max(1, min(K, chunk_steps))
        cbuf, rbuf = self.alloc(ch * stride), self.alloc(ch * R * B)
        try:
            for k0 in range(0, K, ch):
                k1 = min(K, k0 + ch)
                cbuf.upload(t[k0:k1])
                _lib.check(self._L.rgs_step_device_record(self._h, C.c_void_p(cbuf.ptr), stride, k1 - k0, C.c_void_p(rbuf.ptr)))
                out[k0:k1] = rbuf.download((k1 - k0) * R * B).reshape(k1 - k0, R, B)
        finally:
            cbuf.free()
            rbuf.free()
        return out

    def log_open(self, arena_ids, paths):
        """Opens (creates / truncates) one log file per listed arena, written by the library from then on: every
        step_device_log / run_table_logged appends the executed steps' records (29 doubles at 2v2), byte for byte the
        reference's mc_run_<i>.log (include/utils.h:55-75, src/GameSim.cpp:160-170)."""
        ids = np.ascontiguousarray(arena_ids, dtype=np.int32)
        if ids.ndim != 1 or len(paths) != ids.size or ids.size == 0:
            raise TypeError("arena_ids and paths must be equally long, non-empty lists")
        arr = (C.c_char_p * ids.size)(*[os.fsencode(p) for p in paths])
        _lib.check(self._L.rgs_log_open(self._h, int(ids.size), ids.ctypes.data_as(C.POINTER(C.c_int)), arr))

    def step_device_log(self, controls_dev, step_stride, n_steps):
        """step_device + log records appended to the files of log_open (one fused launch, native writer)."""
        ptr = controls_dev.ptr if isinstance(controls_dev, _DeviceBuffer) else int(controls_dev)
        _lib.check(self._L.rgs_step_device_log(self._h, C.c_void_p(ptr), int(step_stride), int(n_steps)))

    def run_table_logged(self, table, chunk_steps=64):
        """Step through a host control table (K, P, 2, B), uploaded in chunks, logging to the files of log_open."""
        t = np.ascontiguousarray(table, dtype=np.float64)
        K = t.shape[0]
        if t.shape[1:] != (self.players, 2, self.n_arenas):
            raise TypeError("table must have shape (K, %d, 2, %d)" % (self.players, self.n_arenas))
        stride = self.players * 2 * self.n_arenas
        ch = max(1, min(K, chunk_steps))
        buf = self.alloc(ch * stride)
        try:
            for k0 in range(0, K, ch):
                k1 = min(K, k0 + ch)
                buf.upload(t[k0:k1])
                self.step_device_log(buf, stride, k1 - k0)
        finally:
            buf.free()

    def log_close(self):
        _lib.check(self._L.rgs_log_close(self._h))

    def pinned(self, shape):
        """A float64 numpy array over page-locked host memory (freed with the sim)."""
        n = int(np.prod(shape))
        p = C.c_void_p()
        _lib.check(self._L.rgs_host_alloc(n, C.byref(p)))
        self._pinned.append(p.value)
        buf = (C.c_double * n).from_address(p.value)
        return np.frombuffer(buf, dtype=np.float64).reshape(shape)

    def step_roundtrip(self, controls, state_out):
        """controls (P, 2, B) -> one step -> state_out (S, B); both should come from pinned()."""
        assert controls.shape == (self.players, 2, self.n_arenas) and controls.flags["C_CONTIGUOUS"]
        assert state_out.shape == (self.state_size, self.n_arenas) and state_out.flags["C_CONTIGUOUS"]
        _lib.check(self._L.rgs_step_roundtrip(self._h, _lib.dptr(controls), _lib.dptr(state_out)))
        return state_out

    def sync(self):
        _lib.check(self._L.rgs_sync(self._h))

    def state(self):
        out = np.empty((self.state_size, self.n_arenas))
        _lib.check(self._L.rgs_get_state(self._h, _lib.dptr(out)))
        return out

    def damage(self):
        out = np.empty((self.players, self.players, self.n_arenas))
        _lib.check(self._L.rgs_get_damage(self._h, _lib.dptr(out)))
        return out

    def time(self):
        out = np.empty(self.n_arenas)
        _lib.check(self._L.rgs_get_time(self._h, _lib.dptr(out)))
        return out

    def set_state(self, state, time=None, damage=None):
        """Restore a checkpoint: state (S, B), time (B,) or scalar, damage (P, P, B) as returned by damage()
        (zeroed when omitted).  Event counters and, with noise, the generators are not restored."""
        s = np.ascontiguousarray(state, dtype=np.float64)
        if s.shape != (self.state_size, self.n_arenas):
            raise TypeError("state must have shape (%d, %d)" % (self.state_size, self.n_arenas))
        t = np.zeros(self.n_arenas) if time is None else np.ascontiguousarray(
            np.broadcast_to(np.asarray(time, dtype=np.float64), (self.n_arenas,)))
        _lib.check(self._L.rgs_set_state(self._h, _lib.dptr(s), _lib.dptr(t)))
        if damage is not None:
            d = np.ascontiguousarray(damage, dtype=np.float64)
            if d.size != self.players * self.players * self.n_arenas:
                raise TypeError("damage must have %d x %d x %d entries" % (self.players, self.players, self.n_arenas))
            _lib.check(self._L.rgs_set_damage(self._h, _lib.dptr(d)))

    def counters(self):
        out = (C.c_int64 * 9)()
        if hasattr(self._L, "rgs_get_counters_ex"):
            _lib.check(self._L.rgs_get_counters_ex(self._h, out, 9))
        else:                                                  # an older A/B build (RGS_LIB_SUFFIX)
            _lib.check(self._L.rgs_get_counters(self._h, out))
            out[8], out[7] = out[7], 0
        names = ["resolved", "cap_hits", "base_overflow", "skipped_after_goal", "event_steps", "flagged"]
        d = dict(zip(names, list(out)[:6]))
        d["auto_prefers_wave"] = int(out[8])
        return d

    def stamps(self, enable=True, clear=False):
        """Diagnostic: enable/disable in-kernel phase stamps (optionally clearing them) and return the totals so far."""
        out = (C.c_int64 * 17)()
        _lib.check(self._L.rgs_get_stamps(self._h, int(bool(enable)) | (2 if clear else 0), out, 17))
        return dict(zip(["lane_cycles", "wave_cycles", "rounds", "total_cycles", "g0_fill", "g0_scan0", "g0_events",
                         "g0_commit", "g0_arenas", "max_total_cycles", "max_wave_cycles", "max_rounds",
                         "max_event_loop_cycles", "ev_head", "ev_carry", "ev_scan", "ev_count"], list(out)))

    def wave_log(self):
        """Diagnostic (stamps library, stamps on): (n_waves, 8) int64 records of the last asynchronous-pool launch; the
        first call only arms the log."""
        nw = 4 * ((self.n_arenas + 255) // 256)
        out = np.zeros((nw, 8), dtype=np.int64)
        _lib.check(self._L.rgs_debug_wave_log(self._h, out.ctypes.data_as(C.POINTER(C.c_int64)), nw))
        return out

    def check_guards(self):
        """Debug: raises if a device buffer's guard zone was overwritten (needs RGS_DEBUG_GUARD_BYTES in the environment)."""
        _lib.check(self._L.rgs_debug_check_guards(self._h))

    def set_variant(self, variant):
        _lib.check(self._L.rgs_set_variant(self._h, int(variant)))

    def set_noise_mode(self, mode):
        """How noise != 0 is drawn, from the next reset() on: "host" (the reference's <random> on the host: draw tapes,
        <= 4096 arenas, one launch per step) or "device" (the same generator on the GPU: any batch size, fused launches).
        Both are bit-identical to the reference where noise_is_exact() (the library was built with the host's log());
        "device" is then the default.  Otherwise "device" is statistical and "host" the default.  ("exact" = "host".)"""
        m = {"host": 0, "exact": 0, "device": 1, 0: 0, 1: 1}[mode]
        _lib.check(self._L.rgs_set_noise_mode(self._h, m))

    @staticmethod
    def noise_is_exact():
        """True if the device noise generator reproduces the reference's draws bit for bit (rgs_noise_is_exact)."""
        return bool(_lib.lib().rgs_noise_is_exact())

    def set_rebalance(self, enable):
        """Pooled variants: deal arenas in lasting contact evenly over the workgroups between fused launches (default on)."""
        _lib.check(self._L.rgs_set_rebalance(self._h, int(bool(enable))))

    def rebalance_info(self):
        """(hot arenas moved by the last deal, perm (B,) int32: the arena each lane slot steps)."""
        n = C.c_int(0)
        perm = np.empty(self.n_arenas, dtype=np.int32)
        _lib.check(self._L.rgs_get_rebalance_info(self._h, C.byref(n), perm.ctypes.data_as(C.POINTER(C.c_int))))
        return int(n.value), perm

    def heat(self):
        """(B,) uint32: steps of the pooled launches since the last deal in which the arena went through the event pass."""
        out = np.zeros(self.n_arenas, dtype=np.uint32)
        _lib.check(self._L.rgs_get_heat(self._h, out.ctypes.data_as(C.POINTER(C.c_uint32))))
        return out

    def set_fast_division(self, level):
        """0/False: IEEE division everywhere; 1: reciprocal with two correction steps; 2/True: also the
        host-certified one-step form (default).  Same results at every level."""
        level = 2 if level is True else int(level)
        _lib.check(self._L.rgs_set_fast_division(self._h, level))

    def set_sparse_watch(self, enable):
        _lib.check(self._L.rgs_set_sparse_watch(self._h, int(enable)))

    def alloc(self, n_doubles):
        return _DeviceBuffer(self, n_doubles)

    def state_device_ptr(self):
        p = C.c_void_p()
        _lib.check(self._L.rgs_state_device_ptr(self._h, C.byref(p)))
        return p.value

    def copy_state_to(self, dst_dev_ptr):
        _lib.check(self._L.rgs_copy_state_to_device(self._h, C.c_void_p(int(dst_dev_ptr))))

    def timer_start(self):
        _lib.check(self._L.rgs_timer_start(self._h))

    def timer_stop(self):
        ms = C.c_float()
        _lib.check(self._L.rgs_timer_stop(self._h, C.byref(ms)))
        return float(ms.value)


class GameSim:
    """Drop-in for the reference's `robo_game_py.GameSim` (src/RoboGamePy.cpp:15-24).

    reset(dt, winning_score, x0_ball, noise, log, logname, seed, tau_puck, tau_player,
    player_mass, puck_mass) -> ndarray(S,); undecided() -> bool; run(vel...) -> ndarray(S,).
    6 to 11 positional arguments are accepted with the C++ defaults of src/GameSim.cpp:21-25.
    `n_per_team` (default 2, the reference's only size) is an extension; run() then takes 2n
    velocity arguments in disc order A1..An, B1..Bn, or one array of shape (2n, 2).

    Every run() is one rgs_step_roundtrip: controls up, one step on the GPU, state down (pinned
    buffers, one synchronisation).  undecided() and the time are derived on the host from what
    came back, exactly as the reference derives them (src/GameSim.cpp:57-60, 158).
    """

    def __init__(self, n_per_team=2, device=0):
        self._b = BatchedGameSim(1, n_per_team, device)
        self.n_per_team = int(n_per_team)
        self._log = None
        self._ctl = self._b.pinned((self._b.players, 2, 1))
        self._st = self._b.pinned((self._b.state_size, 1))
        self._ctl[...] = 0.0
        self._win = 3
        self._dt = 0.05
        self._t = 0.0
        self._state = self._b.state()[:, 0].copy()

    def __repr__(self):
        return "GameSim Object"                      # src/RoboGamePy.cpp:20-23

    def _close_log(self):
        if self._log is not None:
            self._log.close()
            self._log = None

    def __del__(self):
        try:
            self._close_log()
        except Exception:
            pass

    def reset(self, *args, **kwargs):
        a = reset_arguments(*args, **kwargs)
        x0 = np.zeros(4) if a["x0_ball"] is None else np.asarray(a["x0_ball"], dtype=np.float64).reshape(-1)
        if x0.shape != (4,):
            raise TypeError("x0_ball must hold 4 doubles")
        a["x0_ball"] = x0
        state = self._b.reset(**a)
        self._win, self._dt, self._t = a["winning_score"], a["dt"], 0.0
        self._state = state[:, 0].copy()
        self._close_log()                            # src/GameSim.cpp:48-53
        if a["log"]:
            try:
                self._log = open(a["logname"], "wb")
            except OSError:
                self._log = None                     # std::ofstream fails silently (include/utils.h:55-60)
        return self._state.copy()

    def undecided(self):
        return bool(self._state[0] < self._win and self._state[1] < self._win)   # src/GameSim.cpp:57-60

    def run(self, *vels):
        P = self._b.players
        if len(vels) == 1 and np.asarray(vels[0]).size == 2 * P:
            u = np.asarray(vels[0], dtype=np.float64).reshape(P, 2)
        elif len(vels) == P:
            u = np.stack([np.asarray(v, dtype=np.float64).reshape(2) for v in vels])
        else:
            raise TypeError("run() takes %d velocity arguments of shape (2,) (%d given)" % (P, len(vels)))
        live = self.undecided()
        self._ctl[:, :, 0] = u
        self._b.step_roundtrip(self._ctl, self._st)
        self._state = self._st[:, 0].copy()
        if live:
            self._t += self._dt                      # t_ += dt_, src/GameSim.cpp:158
            if self._log is not None:                # a decided game logs nothing, src/GameSim.cpp:121-123
                self._log.write(log_record(self._t, self._state, self._b.damage()[:, :, 0]))
                self._log.flush()
        return self._state.copy()

    # extras (not part of the reference binding)
    def damage(self):
        return self._b.damage()[:, :, 0].copy()

    def time(self):
        return float(self._b.time()[0])

    def set_state(self, state, t=0.0):
        self._b.set_state(np.asarray(state, dtype=np.float64).reshape(-1, 1), t)
        self._state = np.asarray(state, dtype=np.float64).reshape(-1).copy()
        self._t = float(t)
