"""Properties of the CPU oracle that do not need the reference: an independent numpy RK4, the
semantics spelled out in SURVEY.md Appendix A, and the claims the GPU design leans on."""
import os
import struct

import numpy as np

Z = np.zeros(2)


def rk4_free(x, u, tau, coef, h):
    """Independent restatement of src/GameSim.cpp:70-88 / 98-116 for one disc (4-vector)."""
    def f(s):
        return np.array([s[2], s[3], (u[0] - coef * s[2]) / tau, (u[1] - coef * s[3]) / tau])
    k1 = f(x)
    k2 = f(x + k1 * (h / 2.0))
    k3 = f(x + k2 * (h / 2.0))
    k4 = f(x + k3 * h)
    return x + (k1 + k2 * 2.0 + k3 * 2.0 + k4) * (h / 6.0)


def test_event_free_step_is_fifty_rk4_substeps(oracle):
    s = oracle.OracleGameSim()
    x0 = np.array([0.3, -0.2, 1.5, -0.7])
    st = s.reset(0.05, 3, x0)
    u = [np.array([1.0, -2.0]), np.array([0.5, 0.25]), np.array([-1.0, 0.0]), np.array([0.0, 3.0])]
    got = s.run(*u)
    h = 0.05 / 50
    pk = x0.copy()
    pl = [st[6 + 4 * e:10 + 4 * e].copy() for e in range(4)]
    for _ in range(50):
        pk = rk4_free(pk, Z, 0.1, 0.2, h)
        pl = [rk4_free(pl[e], u[e], 0.5, 1.0, h) for e in range(4)]
    assert np.array_equal(got[2:6], pk)
    for e in range(4):
        assert np.array_equal(got[6 + 4 * e:10 + 4 * e], pl[e])
    assert s.time() == 0.05 and s.stats()["resolved"] == 0


def test_dead_rng_draws_do_not_change_values(oracle):
    """The reference draws two normals per f_player call and multiplies them by w_stdev_ = 0
    (src/GameSim.cpp:66-67).  Skipping the draws must not change any value."""
    from robo_game_py import workloads as w
    n, B, K = 2, 64, 300
    x0, _, _ = w.c2_inputs(B, n, seed=4)
    table = w.random_hold_controls(4, B, K, 10.0, hold=15, seed=9)
    a = oracle.batch_run(n, x0, table, K, dt=0.05, winning_score=100, n_threads=4, draw_always=False)
    b = oracle.batch_run(n, x0, table, K, dt=0.05, winning_score=100, n_threads=4, draw_always=True)
    assert a["stats"]["resolved"] > 500
    assert np.array_equal(a["state"], b["state"])
    assert a["stats"] == b["stats"]


def test_noise_is_seeded_and_reproducible(oracle):
    def run(seed):
        s = oracle.OracleGameSim()
        s.reset(0.05, 3, np.zeros(4), 0.3, False, "", seed, 0.1, 0.5, 1.0, 0.5)
        for _ in range(5):
            st = s.run(Z + 1.0, Z, Z, Z)
        return st
    assert np.array_equal(run(7), run(7))
    assert not np.array_equal(run(7), run(8))


def test_wall_reflection_and_goal_side(oracle):
    s = oracle.OracleGameSim()
    s.reset(0.05, 3, np.array([-4.5, 0.1, -6.0, 0.0]))       # towards the LEFT goal mouth
    s.run(Z, Z, Z, Z)
    st = s.run(Z, Z, Z, Z)
    assert (st[0], st[1]) == (0.0, 1.0)                      # WALL_LEFT scores for team B (cpp:324)
    s.reset(0.05, 3, np.array([0.0, 2.2, 0.0, 6.0]))         # puck into the top wall
    st = s.run(Z, Z, Z, Z)
    assert st[5] < 0 and st[3] + 0.175 <= 2.5 + 1e-12        # vy flipped, kept inside
    assert st[0] == 0 and st[1] == 0


def test_player_collision_accumulates_symmetric_damage(oracle):
    s = oracle.OracleGameSim()
    s.reset(0.05, 3, np.array([0.0, 2.0, 0.0, 0.0]))
    # A1 (-2.5, 1.25) and B1 (2.5, 1.25) driven head-on
    for _ in range(80):
        s.run(np.array([6.0, 0.0]), Z, np.array([-6.0, 0.0]), Z)
    d = s.damage()
    assert d[0, 2] > 0 and d[0, 2] == d[2, 0]                # equal masses: m_i*dv_sq both ways (cpp:279-281)
    assert d[1].sum() == 0 and d[3].sum() == 0


def test_log_file_has_29_doubles_per_executed_step(oracle, tmp_path):
    path = str(tmp_path / "game.log")
    s = oracle.OracleGameSim()
    s.reset(0.05, 1, np.array([4.5, 0.2, 5.0, 0.0]), 0.0, True, path)
    states = [s.run(Z, Z, Z, Z) for _ in range(4)]           # decided after step 2: steps 3, 4 log nothing
    s.reset(0.05, 1, np.zeros(4))                            # closes the file (cpp:52-53)
    raw = open(path, "rb").read()
    assert len(raw) == 2 * 29 * 8
    rec = struct.unpack("=58d", raw)
    assert rec[0] == 0.05 and rec[29] == 0.05 + 0.05
    assert rec[30:32] == (1.0, 0.0)                          # TAS, TBS after the goal
    assert rec[3:5] == tuple(states[0][2:4])                 # puck xy
    assert rec[5:7] == tuple(states[0][6:8])                 # A1 xy


def test_n_generic_layout_reduces_to_reference_at_two(oracle):
    for n in (1, 3, 7):
        s = oracle.OracleGameSim(n)
        st = s.reset(0.05, 3, np.array([1.0, 2.0, 3.0, 4.0]))
        assert st.shape == (2 + 4 * (2 * n + 1),)
        ys = st[7:7 + 4 * n:4]
        assert np.all(np.diff(ys) < 0) and abs(ys[0] + ys[-1]) < 1e-15      # rows symmetric about y = 0
        assert np.all(st[6:6 + 4 * n:4] == -2.5) and np.all(st[6 + 4 * n:6 + 8 * n:4] == 2.5)
        if n > 1:
            assert np.min(np.abs(np.diff(ys))) > 0.4                       # no spawn overlap (2R)
