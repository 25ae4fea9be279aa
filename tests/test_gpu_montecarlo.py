"""Callers either side of the hot path: Monte-Carlo harness and the binary log."""
import math
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class Params:                                            # py/mc_example_1.py:9-36, noise-free
    def __init__(self, prefix, num_runs=3, T=4.0, dt=0.05):
        self.log_prefix = prefix
        self.num_runs = num_runs
        self.noise_stdev = 0.0
        self.T = T
        self.dt = dt
        self.tau_player = 0.5
        self.tau_puck = 0.1
        self.player_mass = 1.0
        self.puck_mass = 0.5
        self.winning_score = 2
        self.x0_ball = np.array([-1.0, 0.0, 0.0, 0.0])


def chase(team, sign):
    """Stateless 'run method' in the shape of ClassicalTeam.run: SimState -> (vel1, vel2)."""
    def run(sim_state):
        pk = sim_state.get_puck_pos()
        out = []
        for pid in (1, 2):
            d = pk - sim_state.get_player_pos(team, pid) + np.array([0.3 * sign, 0.05 * pid])
            out.append(4.0 * d / max(1e-9, math.hypot(d[0], d[1])))
        return out[0], out[1]
    return run


def oracle_game(oracle, params, logname, team_a, team_b, seed):
    from robo_game_py.montecarlo import _StateView, steps_per_run
    s = oracle.OracleGameSim()
    s.reset(params.dt, params.winning_score, params.x0_ball, 0.0, True, logname, seed, params.tau_puck,
            params.tau_player, params.player_mass, params.puck_mass)
    st = _StateView(s.run(np.zeros(2), np.zeros(2), np.zeros(2), np.zeros(2)))
    for _ in range(steps_per_run(params)):
        a1, a2 = team_a(st)
        b1, b2 = team_b(st)
        st = _StateView(s.run(a1, a2, b1, b2))
    s.reset(params.dt, params.winning_score, params.x0_ball)
    return st.state


def test_sequential_monte_carlo_writes_reference_files(rgp, oracle, tmp_path):
    from robo_game_py.montecarlo import monte_carlo
    p = Params(str(tmp_path / "gpu_mc"))
    finals = monte_carlo(p, chase("A", -1), chase("B", 1), verbose=False)
    cfg = open(os.path.join(p.log_prefix, "configuration.txt")).read().split("\n")
    assert cfg[:4] == ["3", "4.0", "0.05", "2"] and cfg[4:8] == ["-1.0", "0.0", "0.0", "0.0"] and len(cfg) == 14
    for i in range(1, p.num_runs + 1):
        ref_log = str(tmp_path / ("oracle_%d.log" % i))
        ref_final = oracle_game(oracle, p, ref_log, chase("A", -1), chase("B", 1), i)
        got = open(os.path.join(p.log_prefix, "mc_run_%d.log" % i), "rb").read()
        want = open(ref_log, "rb").read()
        assert len(want) % (29 * 8) == 0 and len(want) > 29 * 8 * 10
        assert got == want, "run %d: log bytes differ" % i
        assert np.array_equal(finals[i - 1], ref_final)


def test_batched_monte_carlo_equals_sequential(rgp, oracle, tmp_path):
    from robo_game_py.montecarlo import monte_carlo_batched
    p = Params(str(tmp_path / "gpu_mc_batched"), num_runs=48)
    p.winning_score = 1                                  # so that some games end before T

    def policy(state, k, t, seeds):                      # vectorised chase with a per-run (seeded) offset
        P, B = 4, state.shape[1]
        u = np.empty((P, 2, B))
        for e in range(P):
            d = state[2:4] - state[6 + 4 * e:8 + 4 * e]
            d = d + np.stack([0.01 * seeds * (1 if e < 2 else -1), 0.02 * (e + 1) * np.ones(B)])
            u[e] = 4.0 * d / np.maximum(1e-9, np.sqrt(d[0] * d[0] + d[1] * d[1]))
        return u

    out = monte_carlo_batched(p, policy, logged_runs=[1, 7, 48])
    assert out["steps"] == 81 and out["state"].shape == (22, 48)
    assert sorted(f for f in os.listdir(p.log_prefix)) == ["configuration.txt", "mc_run_1.log", "mc_run_48.log", "mc_run_7.log"]
    # the same games, one at a time, on the oracle with the same policy restricted to one run
    for r in (1, 7, 48):
        s = oracle.OracleGameSim()
        ref_log = str(tmp_path / ("o_%d.log" % r))
        s.reset(p.dt, p.winning_score, p.x0_ball, 0.0, True, ref_log, r, p.tau_puck, p.tau_player, p.player_mass, p.puck_mass)
        st = s.run(np.zeros(2), np.zeros(2), np.zeros(2), np.zeros(2))
        for k in range(80):
            u = policy(st.reshape(22, 1), k, 0.0, np.array([r]))
            st = s.run(np.ascontiguousarray(u[:, :, 0]))
        s.reset(p.dt, p.winning_score, p.x0_ball)
        assert np.array_equal(out["state"][:, r - 1], st)
        assert open(os.path.join(p.log_prefix, "mc_run_%d.log" % r), "rb").read() == open(ref_log, "rb").read()
    assert out["scores"].sum() > 0 and (~out["undecided"]).any()       # some games were decided early


def test_noisy_batched_monte_carlo_writes_the_reference_files(rgp, oracle, tmp_path):
    """mc_example-style ensemble WITH noise (py/MonteCarlo.py:28-30 passes noise_stdev and seed = run number): all runs in
    one batch, open-loop policy in fused chunks, files from the library's writer -- byte-identical to the oracle playing
    each run alone with seed i (device generator where it is exact, host tapes otherwise)."""
    from robo_game_py.montecarlo import monte_carlo_batched, mindless_policy, steps_per_run
    p = Params(str(tmp_path / "noisy_mc"), num_runs=40, T=3.0)
    p.noise_stdev = 0.3
    p.winning_score = 1
    p.x0_ball = np.array([3.5, 0.2, 6.0, 0.0])                     # shot towards the right goal: some runs end early
    out = monte_carlo_batched(p, mindless_policy, chunk_steps=16)
    N = steps_per_run(p)
    ended = 0
    for r in range(1, p.num_runs + 1):
        s = oracle.OracleGameSim()
        ref_log = str(tmp_path / ("o_%d.log" % r))
        s.reset(p.dt, p.winning_score, p.x0_ball, p.noise_stdev, True, ref_log, r, p.tau_puck, p.tau_player, p.player_mass, p.puck_mass)
        st = s.run(np.zeros(2), np.zeros(2), np.zeros(2), np.zeros(2))
        t = 0.0
        for k in range(N):
            u = mindless_policy(st.reshape(22, 1), k, t, np.array([r]))
            st = s.run(np.ascontiguousarray(u[:, :, 0]))
            t += p.dt
        s.reset(p.dt, p.winning_score, p.x0_ball)
        want = open(ref_log, "rb").read()
        assert open(os.path.join(p.log_prefix, "mc_run_%d.log" % r), "rb").read() == want, "run %d" % r
        assert np.array_equal(out["state"][:, r - 1], st)
        ended += len(want) < 29 * 8 * (N + 1)
    assert 0 < ended                                               # games that were decided stop logging


@pytest.mark.parametrize("variant,chunk", [(0, 64), (1, 64), (2, 64), (3, 64), (4, 64), (5, 64), (5, 7), (2, 1)])
def test_recorded_steps_match_oracle_log_records(rgp, oracle, tmp_path, variant, chunk):
    """run_table_recorded returns exactly the reference's per-step record for every arena (4v4 too).  The records are
    written inside the fused stepping launch by every kernel variant; chunk = steps per launch (7: the "executed" row of a
    launch's first step is judged against the clocks the previous launch left; 1: the old one-launch-per-step shape)."""
    from robo_game_py import workloads as w
    for n in (2, 4):
        B, K, checked = 300, 60, 8
        x0, _, _ = w.c2_inputs(B, n, seed=8)
        table = w.random_hold_controls(2 * n, B, K, 10.0, hold=10, seed=n)
        sim = rgp.BatchedGameSim(B, n)
        sim.set_variant(variant)
        sim.reset(0.05, 1, x0)
        rec = sim.run_table_recorded(table, chunk_steps=chunk)
        R = sim.record_size()
        assert rec.shape == (K, R, B) and R == 3 + 2 + 4 * n + 4 * n * n + 1
        ref = oracle.batch_run(n, x0, table, K, dt=0.05, winning_score=1, n_threads=8)
        assert np.array_equal(sim.state(), ref["state"])
        decided_early = 0
        for a in list(range(checked)) + [B - 1]:
            path = str(tmp_path / ("n%d_a%d.log" % (n, a)))
            o = oracle.OracleGameSim(n)
            o.reset(0.05, 1, x0[:, a], 0.0, True, path)
            for k in range(K):
                o.run(np.ascontiguousarray(table[k, :, :, a]))
            o.reset(0.05, 1, x0[:, a])
            want = np.frombuffer(open(path, "rb").read(), dtype=np.float64).reshape(-1, R - 1)
            ran = rec[:, R - 1, a] != 0.0
            assert ran.sum() == len(want)
            assert np.array_equal(rec[ran, :R - 1, a], want)
            decided_early += len(want) < K
        # every arena: the executed row is exactly "the clock moved", and decided games stop logging for good
        t = rec[:, 0, :]
        moved = np.diff(np.concatenate([np.zeros((1, B)), t]), axis=0) != 0.0
        assert np.array_equal(moved, rec[:, R - 1, :] != 0.0)
        sim.close()


@pytest.mark.parametrize("n,chunk", [(2, 64), (2, 7), (4, 25)])
def test_library_written_log_files_equal_the_oracles(rgp, oracle, tmp_path, n, chunk):
    """rgs_log_open / rgs_step_device_log / rgs_log_close (the reference's Logger, include/utils.h:55-75, for a batch):
    the files of an unordered subset of arenas are byte-identical to the oracle's logs of the same games -- games that
    end early stop growing, a chunk boundary (7, 25) does not show, re-opening truncates, a bad path is an error."""
    from robo_game_py import workloads as w
    B, K = 300, 60
    x0, _, _ = w.c2_inputs(B, n, seed=8)
    table = w.random_hold_controls(2 * n, B, K, 10.0, hold=10, seed=n)
    ref = oracle.batch_run(n, x0, table, K, dt=0.05, winning_score=1, n_threads=8)
    decided = np.flatnonzero((ref["state"][0] >= 1) | (ref["state"][1] >= 1))
    assert 4 <= decided.size < B
    arenas = [299, 0, 17] + decided[:4].tolist()[::-1]
    arenas = sorted(set(arenas), key=arenas.index)
    paths = [str(tmp_path / ("lib_a%d.log" % a)) for a in arenas]
    sim = rgp.BatchedGameSim(B, n)
    with pytest.raises(rgp.RgsError):
        sim.log_open([0], [str(tmp_path / "no_such_dir" / "x.log")])
    with pytest.raises(rgp.RgsError):
        sim.log_open([B], [paths[0]])
    for attempt in range(2):                                      # the second pass re-opens (truncates) the same files
        sim.reset(0.05, 1, x0)
        sim.log_open(arenas, paths)
        sim.run_table_logged(table, chunk_steps=chunk)
        sim.log_close()
    assert np.array_equal(sim.state(), ref["state"])
    R = sim.record_size()
    shorter = 0
    for a, path in zip(arenas, paths):
        opath = str(tmp_path / ("oracle_a%d.log" % a))
        o = oracle.OracleGameSim(n)
        o.reset(0.05, 1, x0[:, a], 0.0, True, opath)
        for k in range(K):
            o.run(np.ascontiguousarray(table[k, :, :, a]))
        o.reset(0.05, 1, x0[:, a])
        want = open(opath, "rb").read()
        assert open(path, "rb").read() == want, "arena %d" % a
        assert len(want) % (8 * (R - 1)) == 0
        shorter += len(want) < 8 * (R - 1) * K
    assert shorter > 0                                            # at least one of these games ended before the table did
    sim.close()


def test_log_open_reports_the_descriptor_limit(rgp, tmp_path):
    """Every log file stays open until log_close: more files than the process may hold open is refused up front with a
    message that says so (below the hard limit the soft limit is raised instead)."""
    import resource
    soft, hard = resource.getrlimit(resource.RLIMIT_NOFILE)
    if hard == resource.RLIM_INFINITY or hard > 100000:
        pytest.skip("no practical descriptor limit here")
    B = hard + 10
    sim = rgp.BatchedGameSim(B, 2)
    with pytest.raises(rgp.RgsError) as e:
        sim.log_open(list(range(B)), [str(tmp_path / ("f%d.log" % a)) for a in range(B)])
    assert "open-file limit" in str(e.value)
    assert not any(tmp_path.iterdir())                            # nothing was created
    sim.close()


@pytest.mark.parametrize("mode", ["host", "device"])
def test_recorded_steps_with_noise(rgp, oracle, tmp_path, mode):
    """noise > 0 (host-tape wave kernel, one step per launch; device generator, fused launches): the in-kernel records
    still equal the reference's log."""
    from robo_game_py import workloads as w
    if mode == "device" and not rgp.BatchedGameSim.noise_is_exact():
        pytest.skip("built without the host's log tables: the device noise mode is statistical")
    n, B, K = 2, 12, 40
    x0, _, _ = w.c2_inputs(B, n, seed=18)
    table = w.random_hold_controls(2 * n, B, K, 8.0, hold=8, seed=5)
    sim = rgp.BatchedGameSim(B, n)
    sim.set_noise_mode(mode)
    sim.reset(0.05, 2, x0, 0.3, False, "", 11)
    rec = sim.run_table_recorded(table, chunk_steps=16)
    R = sim.record_size()
    for a in range(B):
        path = str(tmp_path / ("noise_a%d.log" % a))
        o = oracle.OracleGameSim(n)
        o.reset(0.05, 2, x0[:, a], 0.3, True, path, 11 + a)
        for k in range(K):
            o.run(np.ascontiguousarray(table[k, :, :, a]))
        o.reset(0.05, 2, x0[:, a])
        want = np.frombuffer(open(path, "rb").read(), dtype=np.float64).reshape(-1, R - 1)
        ran = rec[:, R - 1, a] != 0.0
        assert ran.sum() == len(want) and np.array_equal(rec[ran, :R - 1, a], want)
    sim.close()


def test_closed_loop_on_device_matches_host_loop(rgp, oracle):
    """The same chase policy evaluated by torch on the GPU (no PCIe per step) and step by step through
    the oracle with the controls torch produced: identical trajectories."""
    torch = pytest.importorskip("torch")
    from robo_game_py import workloads as w
    from robo_game_py.montecarlo import chase_policy_torch, run_closed_loop_on_device
    n, B, K, dt = 2, 64, 60, 0.05
    x0, _, _ = w.c2_inputs(B, n, seed=21)
    sim = rgp.BatchedGameSim(B, n)
    sim.reset(dt, 3, x0)
    used = []

    def policy(state, k, t):
        u = chase_policy_torch(state, k, t)
        used.append(u.cpu().numpy().copy())
        return u

    final = run_closed_loop_on_device(sim, policy, K).cpu().numpy()
    assert np.array_equal(final, sim.state())
    ref = oracle.batch_run(n, x0, np.stack(used), K, dt=dt, winning_score=3, n_threads=8)
    assert np.array_equal(final, ref["state"])
    assert ref["stats"]["resolved"] > 100
    sim.close()
