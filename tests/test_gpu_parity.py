"""HIP stepper vs. the CPU oracle on identical inputs (all through the C ABI via robo_game_py).

Bar (BASELINE.json north_star): positions / velocities within 1e-5 of the reference step for
step, scores bit-exact.  The dynamics are chaotic once discs collide, so in practice the only
way to stay inside 1e-5 is to be bit-identical; the tests therefore assert exact equality
(numeric ==, which lets -0.0 equal +0.0) and report max |delta| for the written tolerance.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-5


def assert_same(gpu, ref, what):
    gpu = np.asarray(gpu)
    ref = np.asarray(ref)
    if gpu.dtype == bool:
        gpu, ref = gpu.astype(np.int8), ref.astype(np.int8)
    assert gpu.shape == ref.shape
    err = float(np.max(np.abs(gpu - ref))) if gpu.size else 0.0
    assert err <= TOL, "%s: max |delta| = %g > %g" % (what, err, TOL)
    bad = np.argwhere(gpu != ref)
    assert bad.size == 0, "%s: %d values differ (max |delta| %g), first at %s" % (what, len(bad), err, bad[:4].tolist())


def run_both(rgp, oracle, n, x0, table, dt, win=1000000, rec_every=0, variant=None, **phys):
    K, P, _, B = table.shape
    sim = rgp.BatchedGameSim(B, n)
    if variant is not None:
        sim.set_variant(variant)
    sim.reset(dt, win, x0, 0.0, False, "", 0, phys.get("tau_puck", 0.1), phys.get("tau_player", 0.5),
              phys.get("player_mass", 1.0), phys.get("puck_mass", 0.5))
    traj = []
    if rec_every:
        stride = P * 2 * B
        buf = sim.alloc(K * stride)
        buf.upload(table)
        for k0 in range(0, K, rec_every):
            sim.step_device(buf.ptr + 8 * k0 * stride, stride, min(rec_every, K - k0))
            traj.append(sim.state())
        buf.free()
        state = traj[-1] if K % rec_every == 0 else sim.state()
    else:
        state = sim.run_table(table)
    ref = oracle.batch_run(n, x0, table, K, dt=dt, winning_score=win, rec_every=rec_every, n_threads=8,
                           want_damage=True, **phys)
    out = dict(state=state, traj=np.stack(traj) if traj else None, damage=sim.damage().reshape(P * P, B),
               time=sim.time(), counters=sim.counters(), undecided=sim.undecided())
    sim.close()
    return out, ref


def check(out, ref, what):
    assert_same(out["state"][:2], ref["state"][:2], what + " scores")
    assert_same(out["state"], ref["state"], what + " final state")
    assert_same(out["damage"], ref["damage"], what + " damage")
    if out["traj"] is not None and ref["traj"] is not None:
        n = min(len(out["traj"]), len(ref["traj"]))
        assert_same(out["traj"][:n], ref["traj"][:n], what + " trajectory")
    st = ref["stats"]
    c = out["counters"]
    assert c["resolved"] == st["resolved"], (c, st)
    assert c["cap_hits"] == st["cap_hits"], (c, st)
    assert c["skipped_after_goal"] == st["skipped_after_goal"], (c, st)
    assert c["base_overflow"] == st["base_overflow"], (c, st)


def test_device_division_and_sqrt_are_ieee(rgp):
    """The parity argument rests on correctly rounded fp64 '/' and sqrt on the device."""
    import ctypes as C
    rng = np.random.default_rng(1)
    n = 1 << 21
    a = np.concatenate([rng.uniform(0, 10, n // 2), np.exp(rng.uniform(-300, 300, n // 2))])
    b = np.concatenate([rng.uniform(-10, 10, n // 2), np.exp(rng.uniform(-300, 300, n // 2)) * rng.choice([-1, 1], n // 2)])
    a[:8] = [0.0, 1.0, 2.0, 0.5, 1e-310, 4.0, 9.0, 1e308]
    b[:8] = [3.0, 3.0, 0.1, 0.5, 7.0, 1e-310, 0.001, 1e-5]
    q = np.empty(n)
    s = np.empty(n)
    L = rgp._lib.lib()
    rgp._lib.check(L.rgs_selftest_math(0, rgp._lib.dptr(a), rgp._lib.dptr(b), n, rgp._lib.dptr(q), rgp._lib.dptr(s)))
    with np.errstate(all="ignore"):
        assert np.array_equal(q.view(np.uint64), (a / b).view(np.uint64))
        assert np.array_equal(s.view(np.uint64), np.sqrt(a).view(np.uint64))


def _near_midpoint_dividends(tau, count, rng):
    """Doubles a whose quotient a/tau lies as close to a rounding boundary as 53-bit inputs allow:
    a = round_to_double((q + ulp(q)/2) * tau), both neighbours, for random 53-bit q."""
    import math
    mt, et = math.frexp(tau)
    Bt = int(mt * (1 << 53))                          # tau = Bt * 2^(et-53)
    out = []
    for _ in range(count):
        M = int(rng.integers(1 << 52, 1 << 53))
        k = int(rng.integers(-40, 40))
        num = (2 * M + 1) * Bt                        # exact (q + ulp/2) * tau = num * 2^(k-1+et-53)
        sh = num.bit_length() - 53
        lo = num >> sh
        for cand in (lo, lo + 1):
            out.append(math.ldexp(float(cand), sh + k - 1 + et - 53))
            out.append(-out[-1])
    return np.array(out)


@pytest.mark.parametrize("tau", [0.1, 0.5, 1.0, 0.3, 0.7, 1.0 / 3.0, 1.9999999, 123.456, 3e-5])
def test_division_by_tau_rounds_like_ieee(rgp, tau):
    """div_tau<true> (reciprocal + two Markstein corrections) == IEEE '/' on the device == host '/'."""
    rng = np.random.default_rng(int(tau * 1e6) % 9973)
    n = 1 << 21
    a = np.concatenate([rng.uniform(-12, 12, n // 2), np.exp(rng.uniform(-500, 330, n // 2)) * rng.choice([-1, 1], n // 2),
                        _near_midpoint_dividends(tau, 20000, rng), [0.0, 1.0, -1.0, tau, 3 * tau, 1e-300 * tau]])
    a = np.ascontiguousarray(a)
    import ctypes as C
    L = rgp._lib.lib()
    ok, p2, one_ok, cnt = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    worst = np.zeros(65536)
    rgp._lib.check(L.rgs_division_info(tau, C.byref(ok), C.byref(p2), C.byref(one_ok), rgp._lib.dptr(worst), worst.size,
                                       C.byref(cnt)))
    assert ok.value == 1
    worst = worst[:cnt.value]                                # quotients within 2^-51 ulp of a midpoint
    a = np.ascontiguousarray(np.concatenate([a, worst, -worst, worst * 2.0 ** 37, worst * 2.0 ** -61]))
    fast, ieee, one = np.empty(a.size), np.empty(a.size), np.empty(a.size)
    rgp._lib.check(L.rgs_selftest_divtau(0, rgp._lib.dptr(a), a.size, tau, rgp._lib.dptr(fast), rgp._lib.dptr(ieee),
                                         rgp._lib.dptr(one)))
    host = a / tau
    assert np.array_equal(ieee.view(np.uint64), host.view(np.uint64))
    assert np.array_equal(fast, host), "%d of %d quotients differ" % (int((fast != host).sum()), a.size)
    if one_ok.value:                                         # host-certified: one correction step suffices
        assert cnt.value > 0 or p2.value
        assert np.array_equal(one, host), "one-step: %d of %d quotients differ" % (int((one != host).sum()), a.size)
    else:                                                    # not certified => some worst case really fails
        print("tau=%r not certified for one step (%d worst cases)" % (tau, cnt.value))


def test_fast_and_ieee_division_builds_agree(rgp, oracle):
    """The same run with the IEEE-division kernel forced gives the same bits."""
    from robo_game_py import workloads as w
    n, B, K = 2, 512, 200
    x0, amp, ph = w.c2_inputs(B, n, seed=21)
    table = w.circle_controls(amp * 2.5, ph, w.step_times(K, 0.05))
    res = []
    for fast in (2, 1, 0):
        sim = rgp.BatchedGameSim(B, n)
        sim.set_fast_division(fast)
        sim.reset(0.05, 1000, x0)
        res.append(sim.run_table(table))
        sim.close()
    assert_same(res[0], res[1], "one-step vs two-step division")
    assert_same(res[0], res[2], "fast vs ieee division")
    ref = oracle.batch_run(n, x0, table, K, dt=0.05, winning_score=1000, n_threads=8)
    assert_same(res[0], ref["state"], "fast division vs oracle")


@pytest.mark.parametrize("variant", [2, 3], ids=["hybrid", "hybrid2"])
@pytest.mark.parametrize("n,dt,tau_player", [(2, 0.01, 0.5), (2, 0.05, 0.5), (4, 0.05, 0.5), (2, 0.05, 0.005)])
def test_sparse_watch_equals_dense_watch(rgp, oracle, n, dt, tau_player, variant):
    """The lane pass' overlap watch runs every 5th sub-step with widened thresholds; evaluating it at
    every sub-step instead must give the same bits (tau_player = 0.005 forces the dense watch: dt_col/tau
    > 0.1), and fast players (|u| up to 10) must never slip through between checks."""
    from robo_game_py import workloads as w
    B, K = 512, 200
    x0, _, _ = w.c2_inputs(B, n, seed=41)
    table = w.random_hold_controls(2 * n, B, K, 10.0, hold=15, seed=13)
    res, cnt = [], []
    for sparse in (True, False):
        sim = rgp.BatchedGameSim(B, n)
        sim.set_variant(variant)
        sim.set_sparse_watch(sparse)
        sim.reset(dt, 1000, x0, 0.0, False, "", 0, 0.1, tau_player)
        res.append(sim.run_table(table))
        cnt.append(sim.counters())
        sim.close()
    assert_same(res[0], res[1], "sparse vs dense watch")
    assert cnt[0]["resolved"] == cnt[1]["resolved"] and cnt[0]["flagged"] >= cnt[1]["flagged"]
    ref = oracle.batch_run(n, x0, table, K, dt=dt, winning_score=1000, n_threads=8, tau_player=tau_player)
    assert_same(res[0], ref["state"], "sparse watch vs oracle")
    assert cnt[0]["resolved"] == ref["stats"]["resolved"] and ref["stats"]["resolved"] > 500


def test_tiny_velocities_take_the_exact_path(rgp, oracle):
    """Puck speeds decaying through 1e-200 .. denormal: outside div_tau<true>'s window, so the lane
    pass must hand those arenas to the IEEE wave pass; results stay bit-identical."""
    n, B, K = 2, 64, 30
    sim = rgp.BatchedGameSim(B, n)
    sim.reset(0.05, 1000, np.zeros(4))
    st = sim.state()
    mags = np.exp(np.linspace(np.log(1e-323), np.log(1e-150), B))
    st[4] = mags
    st[5] = -mags[::-1]
    st[8] = mags * 3.0                                # a player too (u = 0 below)
    sim.set_state(st, 0.0)
    ctl = np.zeros((4, 2, B))
    osims = []
    for a in range(B):
        o = oracle.OracleGameSim(n)
        o.reset(0.05, 1000, np.zeros(4))
        o.set_state(st[:, a], 0.0)
        osims.append(o)
    for k in range(K):
        got = sim.run(ctl)
        ref = np.stack([o.run(np.ascontiguousarray(ctl[:, :, a])) for a, o in enumerate(osims)], axis=1)
        assert np.array_equal(got, ref), "step %d" % k
    sim.close()


def test_reference_sample_values_through_dropin(rgp):
    """SURVEY.md 8c sample values of the compiled reference, via the GameSim drop-in."""
    Z = np.zeros(2)
    sim = rgp.GameSim()
    assert repr(sim) == "GameSim Object"
    sim.reset(0.05, 3, np.zeros(4), 0.0, False, "", 0, 0.1, 0.5, 1.0, 0.5)
    st = sim.run(np.array([1.0, 0.0]), Z, Z, Z)
    assert st.shape == (22,) and st.dtype == np.float64
    assert list(st[6:10]) == [float("-2.4975812909820143"), 1.25, float("0.095162581964028339"), 0.0]
    sim.reset(0.05, 3, np.array([4.5, 0.2, 5.0, 0.0]), 0.0, False, "")          # 6-arg arity (py/run_sim.py:43)
    st = sim.run(Z, Z, Z, Z)
    assert list(st[2:6]) == [float("4.7379064549100676"), 0.2, float("4.5241870901798578"), 0.0]
    st = sim.run(Z, Z, Z, Z)
    assert list(st[:6]) == [1.0, 0.0, 0.0, 0.0, 0.0, 0.0]
    sim.reset(0.05, 3, np.array([4.5, 1.0, 5.0, 0.0]), 0.0, False, "", 0, 0.1, 0.5, 1.0, 0.5)
    sim.run(Z, Z, Z, Z)
    st = sim.run(Z, Z, Z, Z)
    assert list(st[2:6]) == [float("4.6968234197802463"), 1.0, float("-4.093653765390016"), 0.0]
    sim.reset(0.05, 1, np.array([4.5, 0.2, 5.0, 0.0]), 0.0, False, "", 0, 0.1, 0.5, 1.0, 0.5)
    sim.run(Z, Z, Z, Z)
    st1 = sim.run(Z, Z, Z, Z)
    assert not sim.undecided()
    st2 = sim.run(np.array([2.0, 2.0]), Z, Z, Z)
    assert np.array_equal(st1, st2)


def test_c1_run_sim_loop_100_seconds(rgp, oracle):
    """BASELINE config C1: one 2v2 arena, 100 s at dt = 0.01, mindless controls, driven exactly like
    py/run_sim.py:43-75 (reset, one zero-input run, `while t < T: ...; t += dt` in float64) through the
    drop-in GameSim, against the oracle step by step."""
    from math import cos, sin
    args = (0.01, 4, np.array([-1.0, 0.0, 0.0, 0.0]), 0.0, False, "", 0, 0.1, 0.5, 1.0, 0.5)
    g, o = rgp.GameSim(), oracle.OracleGameSim()
    assert np.array_equal(g.reset(*args), o.reset(*args))
    z = np.zeros(2)
    assert np.array_equal(g.run(z, z, z, z), o.run(z, z, z, z))
    t, steps = 0.0, 0
    while t < 100.0:
        u = np.array([cos(t), sin(t)])
        sg, so = g.run(u, u, u, u), o.run(u, u, u, u)
        if steps % 97 == 0 or t + 0.01 >= 100.0:
            assert np.array_equal(sg, so), "step %d: max |delta| %g" % (steps, np.max(np.abs(sg - so)))
        t += 0.01
        steps += 1
    assert steps in (10000, 10001) and np.array_equal(sg, so)
    assert g.time() == o.time() and g.undecided() == o.undecided()


VARIANTS = [pytest.param(1, id="wave"), pytest.param(2, id="hybrid"), pytest.param(3, id="hybrid2"), pytest.param(4, id="pool"), pytest.param(5, id="apool")]


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("n", [2, 1, 3, 4, 5, 6, 7])
def test_c2_circles_match_oracle(rgp, oracle, n, variant):
    """BASELINE C2/C3-shaped inputs (dt = 0.01, circle controls), recorded every 50 steps."""
    from robo_game_py import workloads as w
    B, K = (1024, 400) if n == 2 else (256, 300)
    x0, amp, ph = w.c2_inputs(B, n)
    table = w.circle_controls(amp, ph, w.step_times(K, 0.01))
    out, ref = run_both(rgp, oracle, n, x0, table, 0.01, rec_every=50, variant=variant)
    check(out, ref, "c2 n=%d" % n)
    assert_same(out["time"], np.full(B, ref_time(K, 0.01)), "time")


def ref_time(K, dt):
    t = 0.0
    for _ in range(K):
        t += dt
    return t


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("n,limit,dt", [(2, 10.0, 0.05), (2, 3.0, 0.01), (4, 10.0, 0.05), (7, 10.0, 0.05)])
def test_random_hold_controls_match_oracle(rgp, oracle, n, limit, dt, variant):
    """Collision-rich open-loop play: random piecewise-constant inputs up to input_limit."""
    from robo_game_py import workloads as w
    B, K = 512, 400
    x0, _, _ = w.c2_inputs(B, n, seed=99)
    table = w.random_hold_controls(2 * n, B, K, limit, hold=20, seed=5 + n)
    out, ref = run_both(rgp, oracle, n, x0, table, dt, rec_every=100, variant=variant)
    check(out, ref, "random n=%d" % n)
    assert ref["stats"]["resolved"] > 1000          # the event pass really ran


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("n", [2, 4])
def test_chase_the_puck_closed_loop(rgp, oracle, n, variant):
    """Sustained contact (many steps end on the 50-iteration cap), scores change, games end."""
    from robo_game_py import workloads as w
    B, K, dt, win = 192, 250, 0.05, 2
    x0, _, _ = w.c2_inputs(B, n, seed=3)
    sim = rgp.BatchedGameSim(B, n)
    sim.set_variant(variant)
    state = sim.reset(dt, win, x0)
    osims = [oracle.OracleGameSim(n) for _ in range(B)]
    for a, o in enumerate(osims):
        o.reset(dt, win, x0[:, a])
    for k in range(K):
        u = w.chase_controls(state, n)
        state = sim.run(u)
        ost = np.stack([o.run(np.ascontiguousarray(u[:, :, a])) for a, o in enumerate(osims)], axis=1)
        assert_same(state, ost, "chase n=%d step %d" % (n, k))
    assert_same(sim.undecided(), np.array([o.undecided() for o in osims]), "undecided")
    assert_same(sim.damage(), np.stack([o.damage() for o in osims], axis=2), "damage")
    assert_same(sim.time(), np.array([o.time() for o in osims]), "time")
    caps = sum(o.stats()["cap_hits"] for o in osims)
    assert sim.counters()["cap_hits"] == caps and caps > 0
    assert state[:2].sum() > 0                       # somebody scored
    sim.close()


def test_non_default_physics_parameters(rgp, oracle):
    from robo_game_py import workloads as w
    B, K, n = 256, 200, 2
    x0, _, _ = w.c2_inputs(B, n, seed=11)
    table = w.random_hold_controls(4, B, K, 8.0, hold=10, seed=2)
    out, ref = run_both(rgp, oracle, n, x0, table, 0.02, win=2, tau_puck=1.0, tau_player=0.3, player_mass=2.0,
                        puck_mass=0.25)
    check(out, ref, "physics params")


def test_set_state_and_ragged_batch(rgp, oracle):
    """Batch sizes that do not fill a block / wave, and restart from an arbitrary state."""
    from robo_game_py import workloads as w
    n = 2
    for B in (1, 3, 65, 130):
        x0, amp, ph = w.c2_inputs(B, n, seed=B)
        table = w.circle_controls(amp * 2.0, ph, w.step_times(120, 0.05))
        out, ref = run_both(rgp, oracle, n, x0, table, 0.05)
        check(out, ref, "ragged B=%d" % B)
    # restart: run 60 steps, read back, set_state into a fresh sim, run 60 more == 120 straight
    B = 64
    x0, amp, ph = w.c2_inputs(B, n, seed=5)
    table = w.circle_controls(amp * 2.0, ph, w.step_times(120, 0.05))
    a = rgp.BatchedGameSim(B, n)
    a.reset(0.05, 1000, x0)
    mid = a.run_table(table[:60])
    tmid, dmid = a.time(), a.damage()
    full = a.run_table(table[60:])
    b = rgp.BatchedGameSim(B, n)
    b.reset(0.05, 1000, x0)
    b.set_state(mid, tmid, dmid)                               # checkpoint = state + time + damage
    assert_same(b.run_table(table[60:]), full, "restart")
    assert_same(b.damage(), a.damage(), "restart damage")
    assert a.damage().sum() > dmid.sum() > 0.0
    a.close()
    b.close()


def test_full_size_batch_is_shard_invariant(rgp):
    """BASELINE C2 size (65 536 arenas): arenas are independent, so any sub-range stepped on its
    own must reproduce its rows of the full batch bit for bit (size-independent property)."""
    from robo_game_py import workloads as w
    n, B, K = 2, 65536, 40
    x0, amp, ph = w.c2_inputs(B, n)
    t = w.step_times(K, 0.01)
    big = rgp.BatchedGameSim(B, n)
    big.reset(0.01, 1000, x0)
    stride = 4 * 2 * B
    buf = big.alloc(K * stride)
    for k0 in range(0, K, 8):
        buf.upload(w.circle_controls(amp, ph, t[k0:k0 + 8]), offset=k0 * stride)
    big.step_device(buf, stride, K)
    full = big.state()
    buf.free()
    big.close()
    for lo, hi in ((0, 1024), (30000, 30257), (65536 - 64, 65536)):
        sub = rgp.BatchedGameSim(hi - lo, n)
        sub.reset(0.01, 1000, x0[:, lo:hi])
        got = sub.run_table(w.circle_controls(amp[:, lo:hi], ph[:, lo:hi], t))
        assert_same(got, full[:, lo:hi], "shard [%d,%d)" % (lo, hi))
        sub.close()


@pytest.mark.parametrize("mode", ["host", "device"])
def test_noise_matches_reference_generator_single_game(rgp, oracle, mode):
    """noise > 0 through the drop-in GameSim: same libstdc++ engine/distribution stream, consumed in the
    reference's call order (fill, then every re-integration), incl. the w_dist_ that survives reset().  Both sources of
    the draws: the host's <random> (tapes) and the generator on the device (one step per launch here: draws generated
    ahead are carried from call to call and dropped by reset)."""
    if mode == "device" and not rgp.BatchedGameSim.noise_is_exact():
        pytest.skip("built without the host's log tables: the device noise mode is statistical")
    Z = np.zeros(2)
    g, o = rgp.GameSim(), oracle.OracleGameSim()
    g._b.set_noise_mode(mode)
    rng = np.random.default_rng(3)
    for seed, noise, x0 in ((7, 0.3, [0.5, 0.2, 3.0, 1.0]), (11, 2.0, [-2.0, 1.0, -4.0, 2.0]), (0, 0.05, [0, 0, 0, 0])):
        sg = g.reset(0.05, 3, np.array(x0, dtype=float), noise, False, "", seed, 0.1, 0.5, 1.0, 0.5)
        so = o.reset(0.05, 3, np.array(x0, dtype=float), noise, False, "", seed, 0.1, 0.5, 1.0, 0.5)
        assert np.array_equal(sg, so)
        for k in range(120):
            u = [rng.uniform(-8, 8, 2) for _ in range(4)]
            sg, so = g.run(*u), o.run(*u)
            assert np.array_equal(sg, so), "seed %d step %d: max |delta| %g" % (seed, k, np.max(np.abs(sg - so)))
        assert g.undecided() == o.undecided()
    assert o.stats()["resolved"] > 10                       # collisions happened, so re-integration draws were used


@pytest.mark.parametrize("mode", ["host", "device"])
@pytest.mark.parametrize("n", [2, 4])
def test_noise_batched_with_many_events(rgp, oracle, n, mode):
    """Batched noise: arena a uses seed + a; chase play forces steps with more events than the first,
    short draw tape holds, which exercises the redo-with-a-longer-tape path (host tapes) and 49-event steps whose
    re-integrations draw block after block (device generator)."""
    from robo_game_py import workloads as w
    if mode == "device" and not rgp.BatchedGameSim.noise_is_exact():
        pytest.skip("built without the host's log tables: the device noise mode is statistical")
    B, K, dt = 48, 80, 0.05
    x0, _, _ = w.c2_inputs(B, n, seed=9)
    sim = rgp.BatchedGameSim(B, n)
    sim.set_noise_mode(mode)
    state = sim.reset(dt, 3, x0, 0.4, False, "", 100, 0.1, 0.5, 1.0, 0.5)
    osims = [oracle.OracleGameSim(n) for _ in range(B)]
    for a, o in enumerate(osims):
        o.reset(dt, 3, x0[:, a], 0.4, False, "", 100 + a, 0.1, 0.5, 1.0, 0.5)
    for k in range(K):
        u = w.chase_controls(state, n)
        state = sim.run(u)
        ost = np.stack([o.run(np.ascontiguousarray(u[:, :, a])) for a, o in enumerate(osims)], axis=1)
        assert_same(state, ost, "noise chase n=%d step %d" % (n, k))
    assert_same(sim.damage(), np.stack([o.damage() for o in osims], axis=2), "damage")
    caps = sum(o.stats()["cap_hits"] for o in osims)
    assert sim.counters()["cap_hits"] == caps and caps > 0   # 49-event steps need the worst-case tape
    sim.close()


@pytest.mark.parametrize("seed", range(12))
def test_parameter_fuzz(rgp, oracle, seed):
    """Seeded random physics / timing parameters, team sizes and control styles, every variant, against
    the oracle: final state, damage, event counters.  Includes taus that are not powers of two (two-step
    or one-step reciprocal division), dt_col/tau > 0.1 (dense watch) and tiny winning scores (games that
    end mid-run and freeze)."""
    from robo_game_py import workloads as w
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.choice([1, 2, 2, 3, 4, 5, 6, 7]))
    B, K = 192, int(rng.integers(60, 160))
    dt = float(rng.choice([0.002, 0.01, 0.02, 0.05, 0.1]))
    phys = dict(tau_puck=float(rng.choice([0.1, 0.1, 0.05, 0.3, 1.0, 0.013])),
                tau_player=float(rng.choice([0.5, 0.5, 0.25, 0.4, 0.07, 1.0 / 3.0])),
                player_mass=float(rng.uniform(0.3, 3.0)), puck_mass=float(rng.uniform(0.1, 2.0)))
    win = int(rng.choice([1, 2, 3, 1000]))
    x0, amp, ph = w.c2_inputs(B, n, seed=seed)
    x0[2:] *= rng.uniform(0.5, 4.0)
    style = seed % 3
    if style == 0:
        table = w.random_hold_controls(2 * n, B, K, float(rng.uniform(2, 10)), hold=int(rng.integers(3, 30)), seed=seed)
    elif style == 1:
        table = w.circle_controls(amp * rng.uniform(0.5, 3.0), ph, w.step_times(K, dt))
    else:                                                     # everybody towards the puck's start: pile-ups
        s0 = oracle.OracleGameSim(n).reset(dt, win, np.zeros(4))
        table = np.zeros((K, 2 * n, 2, B))
        for e in range(2 * n):
            d = x0[:2] - s0[6 + 4 * e:8 + 4 * e, None]
            table[:, e] = (float(rng.uniform(2, 6)) * d / np.maximum(1e-9, np.sqrt((d * d).sum(0))))[None]
    ref = oracle.batch_run(n, x0, table, K, dt=dt, winning_score=win, n_threads=8, want_damage=True, **phys)
    for variant in (1, 2, 3, 4, 5):
        sim = rgp.BatchedGameSim(B, n)
        sim.set_variant(variant)
        sim.reset(dt, win, x0, 0.0, False, "", 0, phys["tau_puck"], phys["tau_player"], phys["player_mass"], phys["puck_mass"])
        got = sim.run_table(table, chunk_steps=int(rng.integers(1, K + 1)))
        what = "fuzz seed %d n=%d dt=%g variant %d %r" % (seed, n, dt, variant, phys)
        assert_same(got, ref["state"], what)
        assert_same(sim.damage().reshape(4 * n * n, B), ref["damage"], what + " damage")
        c = sim.counters()
        for k in ("resolved", "cap_hits", "skipped_after_goal", "base_overflow"):
            assert c[k] == ref["stats"][k], (what, k, c, ref["stats"])
        assert_same(sim.undecided(), (ref["state"][0] < win) & (ref["state"][1] < win), what + " undecided")
        sim.close()


@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4, 5])
def test_pathological_inputs_neither_hang_nor_diverge(rgp, oracle, variant):
    """NaN / inf / 1e200 controls and states, dt = 0, a tiny tau: every loop of the kernels is bounded
    and every grid index clamped, so the step returns, and what it returns equals the oracle's result
    value for value (NaN where the oracle has NaN)."""
    n, B, K = 2, 96, 12
    rng = np.random.default_rng(5)
    x0 = np.stack([rng.uniform(-3, 3, B), rng.uniform(-1.5, 1.5, B), rng.uniform(-2, 2, B), rng.uniform(-2, 2, B)])
    table = rng.uniform(-6, 6, (K, 4, 2, B))
    table[:, 0, 0, 0:8] = np.nan
    table[3:, 1, 1, 8:16] = np.inf
    table[:, 2, :, 16:24] = 1e200
    table[:, 3, 0, 24:32] = -1e308
    x0[2, 32:40] = 1e150                                     # absurd puck speed
    x0[0, 40:44] = np.nan
    for dt, phys in ((0.05, {}), (0.0, {}), (0.05, dict(tau_puck=1e-12, tau_player=1e-9)), (1e6, {})):
        ref = oracle.batch_run(n, x0, table, K, dt=dt, winning_score=3, n_threads=4, **phys)
        sim = rgp.BatchedGameSim(B, n)
        sim.set_variant(variant)
        sim.reset(dt, 3, x0, 0.0, False, "", 0, phys.get("tau_puck", 0.1), phys.get("tau_player", 0.5), 1.0, 0.5)
        got = sim.run_table(table)
        sim.close()
        with np.errstate(all="ignore"):
            same = (got == ref["state"]) | (np.isnan(got) & np.isnan(ref["state"]))
        assert same.all(), "dt=%g %r: %d values differ, arenas %s" % (dt, phys, int((~same).sum()),
                                                                      sorted(set(np.argwhere(~same)[:, 1].tolist()))[:10])


@pytest.mark.parametrize("variant", [1, 2, 3, 4, 5])
@pytest.mark.parametrize("n", [2, 4])
def test_in_kernel_trajectory_snapshots(rgp, oracle, n, variant):
    """State snapshots taken inside one fused launch equal the oracle's recorded trajectory."""
    from robo_game_py import workloads as w
    B, K, rec = 200, 120, 7
    x0, _, _ = w.c2_inputs(B, n, seed=12)
    table = w.random_hold_controls(2 * n, B, K, 8.0, hold=9, seed=4)
    sim = rgp.BatchedGameSim(B, n)
    sim.set_variant(variant)
    sim.reset(0.05, 3, x0)
    traj = sim.run_table_traj(table, rec)
    ref = oracle.batch_run(n, x0, table, K, dt=0.05, winning_score=3, rec_every=rec, n_threads=8)
    assert traj.shape == ref["traj"].shape == (K // rec, 2 + 4 * (2 * n + 1), B)
    assert_same(traj, ref["traj"], "in-kernel snapshots")
    assert_same(sim.state(), ref["state"], "final state")
    sim.close()


def test_auto_variant_switches_in_sustained_contact(rgp, oracle):
    """AUTO watches the event counters between launches: a jammed arena set (everybody pushed to the
    centre) makes it prefer the wave-per-arena kernel, calm play brings the hybrid back; the states are
    the same bits as the oracle's throughout."""
    n, B, dt = 2, 512, 0.05
    x0 = np.zeros((4, B))
    sim = rgp.BatchedGameSim(B, n)
    state = sim.reset(dt, 1000, x0)
    pull = np.zeros((4, 2, B))
    for e in range(4):
        d = -state[6 + 4 * e:8 + 4 * e]
        pull[e] = 4.0 * d / np.sqrt((d * d).sum(0))
    table = np.ascontiguousarray(np.broadcast_to(pull, (70, 4, 2, B)))
    prefers = []
    for _ in range(4):                                        # 4 launches of 70 steps: jam builds up
        state = sim.run_table(table, chunk_steps=70)
        prefers.append(sim.counters()["auto_prefers_wave"])
    assert prefers[-1] == 1, prefers
    ref = oracle.batch_run(n, x0, np.concatenate([table] * 4), 280, dt=dt, winning_score=1000, n_threads=8)
    assert_same(state, ref["state"], "jam")
    calm = np.zeros((10, 4, 2, B))
    spread = np.array([[-3.0, 1.5], [-3.0, -1.5], [3.0, 1.5], [3.0, -1.5]])
    for k in range(60):                                       # closed loop: walk everybody back out, then rest
        pos = state[6:22].reshape(4, 4, B)[:, :2]
        calm[:] = np.clip(1.0 * (spread[:, :, None] - pos), -1.5, 1.5)[None]
        state = sim.run_table(calm, chunk_steps=10)
    assert sim.counters()["auto_prefers_wave"] == 0
    sim.close()


def test_noise_is_limited_not_faked(rgp):
    sim = rgp.BatchedGameSim(5000, 2)
    sim.set_noise_mode("host")                                     # draw tapes from the host's <random>: <= 4096 arenas
    with pytest.raises(rgp.RgsError):
        sim.reset(0.05, 3, np.zeros(4), 0.1)
    sim.close()


def oracle_from_state(oracle, n, st, table, dt, win=3):
    """Per-arena oracle runs started from an arbitrary state (batch_run always starts from a reset)."""
    K, P, _, B = table.shape
    out = np.zeros_like(st)
    stats = {}
    for a in range(B):
        o = oracle.OracleGameSim(n)
        o.reset(dt, win, np.zeros(4))
        o.set_state(st[:, a].copy(), 0.0)
        for k in range(K):
            out[:, a] = o.run(*[table[k, q, :, a].copy() for q in range(P)])
        for key, val in o.stats().items():
            stats[key] = stats.get(key, 0) + val
    return out, stats


@pytest.mark.gpu
@pytest.mark.parametrize("variant", [1, 2, 3, 4, 5])
def test_division_guard_on_one_axis_only(rgp, oracle, variant):
    """A control far below the fast-division window on ONE axis trips the guard in one lane of the
    two-lane pass only; the arena must still be re-run from column 1 (an early wall bounce of another
    disc is otherwise found late).  Regression: the lanes exchanged the flag inside a short-circuited
    condition, i.e. with the sending lane masked off."""
    n, B, K, dt = 2, 64, 6, 0.02
    x0 = np.zeros((4, B))
    table = np.zeros((K, 4, 2, B))
    table[:, 0, 1, :] = 1e-250                               # A1's y command: outside the window, x is not
    table[:, 3, 1, :] = -5.0                                 # B2 is pushed into the bottom wall
    st = np.zeros((22, B))
    st[6:10] = np.array([[-2.5], [1.25], [0.0], [0.0]])
    st[10:14] = np.array([[-2.5], [-1.25], [0.0], [0.0]])
    st[14:18] = np.array([[2.5], [1.25], [0.0], [0.0]])
    st[18] = 2.5
    st[19] = -2.29 + 0.004 * np.arange(B) / B               # 1 .. 5 mm above the wall zone
    st[21] = -4.4
    sim = rgp.BatchedGameSim(B, n)
    sim.set_variant(variant)
    sim.reset(dt, 3, x0)
    sim.set_state(st)
    got = sim.run_table(table)
    c = sim.counters()
    sim.close()
    ref, stats = oracle_from_state(oracle, n, st, table, dt)
    assert c["resolved"] == stats["resolved"] and c["resolved"] >= B
    assert_same(got, ref, "one-axis guard")


@pytest.mark.gpu
@pytest.mark.parametrize("variant", [2, 3])
@pytest.mark.parametrize("dt", [0.01, 0.05])
def test_wall_watch_at_grazing_and_reversing_approaches(rgp, oracle, variant, dt):
    """The lane pass tests walls at column 50 only, plus an up-front bound for chains whose velocity
    reverses inside the step.  Discs that dip into a wall zone and come back out within one step
    (command opposite to the velocity), that stop just short of it, or that sit inside it moving away
    are the cases that rule has to get right."""
    n, B, K = 2, 1024, 3
    rng = np.random.default_rng(77)
    x0 = np.zeros((4, B))
    st = np.zeros((22, B))
    st[6:10] = np.array([[-2.5], [0.0], [0.0], [0.0]])
    st[10:14] = np.array([[-1.0], [0.0], [0.0], [0.0]])
    st[14:18] = np.array([[1.0], [0.0], [0.0], [0.0]])
    st[18:22] = np.array([[3.5], [0.0], [0.0], [0.0]])
    st[2] = 0.0; st[3] = 1.5                                 # puck parked away from everything
    v = rng.uniform(0.2, 6.0, B)                             # A1 flies at the top wall ...
    gap = rng.uniform(-0.002, 1.2, B) * v * dt               # ... from where it does / does not reach the zone
    st[7] = 2.3 - gap
    st[9] = v
    table = np.zeros((K, 4, 2, B))
    table[:, 0, 1, :] = -rng.uniform(0.0, 300.0, B)         # ... while commanded back down, some hard enough to reverse
    table[:, 0, 0, :] = rng.uniform(-3, 3, B)
    table[:, 1:, :, :] = rng.uniform(-1, 1, (K, 3, 2, B))
    sim = rgp.BatchedGameSim(B, n)
    sim.set_variant(variant)
    sim.reset(dt, 3, x0)
    sim.set_state(st)
    got = sim.run_table(table)
    c = sim.counters()
    sim.close()
    ref, stats = oracle_from_state(oracle, n, st, table, dt)
    assert c["resolved"] == stats["resolved"]
    assert 0.05 * B < stats["steps_with_events"] < 2.5 * B  # both outcomes are well represented
    assert_same(got, ref, "grazing / reversing wall approaches")


@pytest.mark.gpu
@pytest.mark.parametrize("variant", [2, 3])
@pytest.mark.parametrize("tau_player,dt", [(0.25, 0.01), (2.0, 0.03), (0.0625, 0.005), (0.5, 0.037)])
def test_power_of_two_tau_folded_into_step_constants(rgp, oracle, tau_player, dt, variant):
    """With tau_player = 2^j the lane pass never divides or multiplies by it: the factor lives in the RK4
    step constants (gsim_lane.hpp, rk4_chains).  Other powers of two than the default 0.5, and step
    sizes whose thirds and sixths are not round numbers, against the oracle's plain divisions."""
    from robo_game_py import workloads as w
    n, B, K = 2, 512, 150
    x0, amp, ph = w.c2_inputs(B, n, seed=31)
    table = w.circle_controls(amp * 1.5, ph, w.step_times(K, dt))
    out, ref = run_both(rgp, oracle, n, x0, table, dt, variant=variant, tau_player=tau_player)
    check(out, ref, "tau_player=%g dt=%g" % (tau_player, dt))


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 6, 7])
def test_every_kernel_instantiation_matches_oracle(rgp, oracle, n):
    """Every (team size, stepper variant, division mode) the library can launch, on one eventful workload.
    The large-team kernels sit at the register limit and one build of k_step_hybrid2<7> faulted on the device
    without any source-level cause: each instantiation is therefore exercised explicitly."""
    from robo_game_py import workloads as w
    B, K, dt = 160, 48, 0.02
    x0, amp, ph = w.c2_inputs(B, n, seed=100 + n)
    table = w.circle_controls(amp * 2.5, ph, w.step_times(K, dt))
    modes = [("pow2+onestep", 0.1, 0.5, 2), ("pow2+markstein", 0.1, 0.5, 1), ("markstein", 0.13, 0.37, 2), ("ieee", 0.1, 0.5, 0)]
    for name, tau_puck, tau_player, level in modes:
        ref = oracle.batch_run(n, x0, table, K, dt=dt, winning_score=1000, n_threads=8, want_damage=True,
                               tau_puck=tau_puck, tau_player=tau_player)
        assert ref["stats"]["resolved"] > 50
        for variant in (1, 2, 3, 4, 5):
            sim = rgp.BatchedGameSim(B, n)
            sim.set_variant(variant)
            sim.set_fast_division(level)
            sim.reset(dt, 1000, x0, 0.0, False, "", 0, tau_puck, tau_player)
            got = sim.run_table(table, chunk_steps=K)
            what = "n=%d variant=%d %s" % (n, variant, name)
            assert_same(got, ref["state"], what)
            assert_same(sim.damage().reshape(4 * n * n, B), ref["damage"], what + " damage")
            assert sim.counters()["resolved"] == ref["stats"]["resolved"], what
            sim.close()


@pytest.mark.parametrize("variant", [4, 5], ids=["pool", "apool"])
@pytest.mark.parametrize("B", [1, 3, 63, 257, 1000])
def test_pooled_kernels_on_ragged_batches(rgp, oracle, variant, B):
    """The pooled kernels own 256 arenas per workgroup: batches that leave wavefronts partly or wholly empty, a
    single arena, launches of one step and of many, and a launch that records trajectories (which switches the
    asynchronous pool's follow-up and straggler rules off) must all give the oracle's bits."""
    from robo_game_py import workloads as w
    n, K, dt = 2, 40, 0.05
    x0, amp, ph = w.c2_inputs(B, n, seed=500 + B)
    table = w.circle_controls(amp * 2.0, ph, w.step_times(K, dt))
    ref = oracle.batch_run(n, x0, table, K, dt=dt, winning_score=3, n_threads=8, rec_every=8)
    sim = rgp.BatchedGameSim(B, n)
    sim.set_variant(variant)
    sim.reset(dt, 3, x0)
    got = sim.run_table(table[:1], chunk_steps=1)              # one launch of one step
    got = sim.run_table(table[1:], chunk_steps=K)              # one launch of the other 39
    assert_same(got, ref["state"], "B=%d variant=%d fused" % (B, variant))
    assert sim.counters()["resolved"] == ref["stats"]["resolved"]
    sim.reset(dt, 3, x0)
    traj = sim.run_table_traj(table, 8)
    assert_same(traj, ref["traj"], "B=%d variant=%d in-kernel snapshots" % (B, variant))
    sim.close()


def test_pool_timeout_is_a_sticky_error_until_reset(rgp):
    """a pooled kernel's bounded wait running out leaves arenas short of their steps; every synchronising
    call must then fail (not hand out the state with RGS_OK) until the handle is reset."""
    B, n = 64, 2
    sim = rgp.BatchedGameSim(B, n)
    sim.reset(0.05, 3, np.zeros(4))
    sim.state()                                                    # healthy
    rgp._lib.check(rgp._lib.lib().rgs_debug_inject_pool_timeout(sim._h))
    for call in (sim.state, sim.time, sim.damage, sim.undecided, sim.sync, sim.counters,
                 lambda: sim.step(np.zeros((2 * n, 2, B)))):
        with pytest.raises(rgp.RgsError) as e:
            call()
        assert "timed out" in str(e.value)
    sim.reset(0.05, 3, np.zeros(4))                                # reset clears counters and the error
    sim.step(np.zeros((2 * n, 2, B)))
    assert sim.state().shape == (22, B) and sim.counters()["resolved"] == 0
    sim.close()


@pytest.mark.parametrize("variant", [4, 5], ids=["pool", "apool"])
def test_hot_arenas_are_dealt_evenly_and_results_do_not_change(rgp, oracle, variant):
    """Pooled kernels: arenas in lasting contact (here: pucks spawned inside a player, all of them in the first two
    workgroups) are dealt over the workgroups between fused launches.  The deal must be a permutation that evens out
    the hot arenas, and every value must stay the oracle's -- with the dealing on, off, and across several deals."""
    from robo_game_py import workloads as w
    n, dt, B = 2, 0.01, 8 * 256 + 37
    x0, amp, ph = w.c2_inputs(B, n, seed=77)
    x0 = x0 * np.array([0.3, 0.3, 0.05, 0.05])[:, None]            # calm pucks near the centre ...
    hot = np.arange(B) < 400                                       # ... except 400 spawned inside player A1, at rest
    x0[:, hot] = np.array([[-2.5 + 0.05], [1.25 + 0.02], [0.0], [0.0]])
    K = 6 + 12 + 12 + 3
    table = w.circle_controls(amp * 0.2, ph, w.step_times(K, dt))
    ref = oracle.batch_run(n, x0, table, K, dt=dt, winning_score=1000, n_threads=8, want_damage=True)
    stride = 2 * n * 2 * B
    finals = []
    for dealing in (True, False):
        sim = rgp.BatchedGameSim(B, n)
        sim.set_variant(variant)
        sim.set_rebalance(dealing)
        sim.reset(dt, 1000, x0)
        buf = sim.alloc(K * stride)
        buf.upload(table)
        moved = []
        k = 0
        for m in (6, 12, 12, 3):                                   # history, first deal, second deal, a short launch (no deal)
            sim.step_device(buf.ptr + 8 * k * stride, stride, m)
            k += m
            nsw, perm = sim.rebalance_info()
            moved.append(nsw)
            assert sorted(perm.tolist()) == list(range(B)), "the deal is not a permutation"
            assert np.array_equal(perm[8 * 256:], np.arange(8 * 256, B)), "the ragged last workgroup keeps its arenas"
            if dealing and nsw:
                per_wg = np.array([np.count_nonzero(hot[perm[q * 256:(q + 1) * 256]]) for q in range(8)])
                # 256 + 144 spawn-jammed arenas in workgroups 0 and 1 (nearly all of them hot by the library's count:
                # in the event pass at >= 7/8 of the steps), q = hot / 8 per workgroup: each of the two gives its
                # 64-arena maximum per deal, the 128 fill the open places of the following workgroups up to q
                assert nsw == 128 and per_wg.sum() == 400, (nsw, per_wg)
                assert per_wg[0] == 192 and per_wg[1] == 80 and 44 <= per_wg[2] == per_wg[3] <= 50, per_wg
                assert per_wg[4] == 128 - 2 * per_wg[2] and per_wg[5:].tolist() == [0, 0, 0], per_wg
        got = sim.state()
        assert_same(got, ref["state"], "variant %d dealing %s" % (variant, dealing))
        assert_same(sim.damage().reshape(4 * n * n, B), ref["damage"], "damage, dealing %s" % dealing)
        assert sim.counters()["resolved"] == ref["stats"]["resolved"]
        if dealing:
            assert moved[0] == 0 and moved[1] > 0, moved           # no history yet -> identity; then hot arenas move
        else:
            assert moved == [0, 0, 0, 0]
        finals.append(got)
        buf.free()
        sim.close()
    assert np.array_equal(finals[0], finals[1])


@pytest.mark.parametrize("seed", [0, 1, 7, 2147483647, 123456789, -5, 4294967297])
def test_device_noise_generator_reproduces_the_hosts_random(rgp, oracle, seed):
    """csrc/gsim_rng.hpp against the host's <random> (what the reference draws from): the canonical uniforms of one
    million engine outputs are bit-equal (integer engine + IEEE arithmetic), the normals -- same polar method, same
    order, the device's log() -- agree to 2 ulp."""
    import ctypes as C
    n, attempts = 4096, 250000                                     # 4 engine outputs per attempt
    normals, canon = np.empty(n), np.empty(2 * attempts)
    x_end, have = C.c_uint32(), C.c_int()
    rgp._lib.check(rgp._lib.lib().rgs_selftest_rng(0, seed, n, attempts, rgp._lib.dptr(normals), rgp._lib.dptr(canon),
                                                   C.byref(x_end), C.byref(have)))
    want_n, want_c = oracle.rng_streams(seed, n, 2 * attempts)
    assert np.array_equal(canon.view(np.uint64), want_c.view(np.uint64))
    ulp = np.abs(normals.view(np.int64) - want_n.view(np.int64))
    assert np.sign(normals).tolist() == np.sign(want_n).tolist() and ulp.max() <= 2, ulp.max()
    if rgp._lib.lib().rgs_noise_is_exact():                        # built with the host's log (csrc/gsim_hostlog.hpp):
        assert np.array_equal(normals.view(np.uint64), want_n.view(np.uint64))   # std::normal_distribution's draws bit for bit
    assert have.value >= n


@pytest.mark.parametrize("n,B,K", [(2, 6000, 60), (4, 1500, 40), (1, 1000, 80)])
def test_device_noise_mode_is_bit_identical_to_the_reference_draws(rgp, oracle, n, B, K):
    """With the host's log() restated on the device (rgs_noise_is_exact) the device generator IS the reference's
    std::default_random_engine + std::normal_distribution: noisy games -- collisions, goals, re-integrations that consume
    draws out of column order, launch boundaries with draws generated ahead, more arenas than the host-tape mode takes --
    end in the oracle's states, damage and event counts bit for bit."""
    from robo_game_py import workloads as w
    if not rgp._lib.lib().rgs_noise_is_exact():
        pytest.skip("built without the host's log tables: the device noise mode is statistical")
    dt, sigma = 0.05, 0.3
    x0, amp, ph = w.c2_inputs(B, n, seed=90 + n)
    x0 = x0 * np.array([0.4, 0.4, 0.5, 0.5])[:, None]
    x0[:, ::5] = np.array([4.0, 0.1, 8.0, 0.0])[:, None]           # every fifth puck is shot at the right goal
    x0[2:, 1::7] = np.array([1e-200, -3e-250])[:, None]            # pucks below the reciprocal division's window: IEEE '/' there
    table = w.circle_controls(amp * 1.5, ph, w.step_times(K, dt))
    ref = oracle.batch_run(n, x0, table, K, dt=dt, winning_score=2, noise=sigma, seed_base=17, n_threads=8, want_damage=True)
    sim = rgp.BatchedGameSim(B, n)
    sim.set_noise_mode("device")
    sim.reset(dt, 2, x0, sigma, False, "", 17)
    sim.run_table(table[:7], chunk_steps=7)
    got = sim.run_table(table[7:], chunk_steps=K - 7)
    c = sim.counters()
    assert_same(got, ref["state"], "device noise, n=%d" % n)
    P = 2 * n
    assert_same(sim.damage().reshape(P * P, B), ref["damage"], "damage")
    assert c["resolved"] == ref["stats"]["resolved"] and c["cap_hits"] == ref["stats"]["cap_hits"]
    assert ref["stats"]["resolved"] > B and (ref["state"][:2] > 0).any()      # collisions and goals happened
    sim.close()


def test_device_noise_mode_statistics_and_scale(rgp, oracle):
    """noise > 0 with the device generator (statistical parity): identical arenas that differ only in their seed.
    (a) Before anything collides the two modes agree to rounding (same uniforms, normals within 2 ulp).  (b) Over 4096
    noisy games the means and variances of the final state agree with the exact mode's ensemble inside 5-sigma bands.
    (c) No 4096-arena limit, fused launches; the exact mode still refuses more than 4096."""
    from robo_game_py import workloads as w
    n, dt, sigma = 2, 0.05, 0.4
    B, K = 4096, 30
    x0 = np.repeat(np.array([[1.0], [0.4], [-1.5], [0.8]]), B, axis=1)
    _, amp, ph = w.c2_inputs(1, n, seed=3)
    table = np.repeat(w.circle_controls(amp * 1.5, ph, w.step_times(K, dt)), B, axis=3)
    ref = oracle.batch_run(n, x0, table, K, dt=dt, winning_score=1000, noise=sigma, seed_base=100, n_threads=8)
    sim = rgp.BatchedGameSim(B, n)
    sim.set_noise_mode("device")
    sim.reset(dt, 1000, x0, sigma, False, "", 100)
    early = sim.run_table(table[:2], chunk_steps=2)
    ref2 = oracle.batch_run(n, x0, table[:2], 2, dt=dt, winning_score=1000, noise=sigma, seed_base=100, n_threads=8)
    assert np.max(np.abs(early - ref2["state"])) < 1e-12                 # (a) two collision-free steps
    sim.reset(dt, 1000, x0, sigma, False, "", 100)
    got = sim.run_table(table, chunk_steps=K)                             # one fused launch
    for r in range(2, got.shape[0]):                                      # (b) every position / velocity row
        g, o = got[r], ref["state"][r]
        se = np.sqrt(g.var() / B + o.var() / B)
        assert abs(g.mean() - o.mean()) <= 5 * se + 1e-12, (r, g.mean(), o.mean(), se)
        assert abs(g.var() - o.var()) <= 0.2 * max(g.var(), o.var()) + 1e-12, (r, g.var(), o.var())
    assert got[:, 0].tolist() != got[:, 1].tolist()                       # different seeds, different games
    sim.close()
    big = rgp.BatchedGameSim(20000, n)                                    # (c)
    big.set_noise_mode("host")
    with pytest.raises(rgp.RgsError):
        big.reset(dt, 1000, x0[:, :1].repeat(20000, axis=1), sigma, False, "", 5)
    big.set_noise_mode("device")
    big.reset(dt, 1000, x0[:, :1].repeat(20000, axis=1), sigma, False, "", 5)
    st = big.run_table(np.repeat(table[:8, :, :, :1], 20000, axis=3), chunk_steps=8)
    assert np.isfinite(st).all() and np.unique(st[6]).size > 19000        # A1's x: every arena its own noise (the puck draws none)
    big.close()


@pytest.mark.parametrize("n", [4, 6, 7])
def test_pile_ups_in_the_split_grid(rgp, oracle, n):
    """Two-lane hybrid with the split grid (4v4, 6v6, 7v7), everybody pulled to the centre: multi-disc pile-ups, where a
    re-integrated disc keeps meeting discs the lane pass had not flagged (their velocity rows are read from the global
    scratch).  State, damage and event counts as the oracle's."""
    from robo_game_py import workloads as w
    B, K, dt = 192, 90, 0.05
    x0, amp, ph = w.c2_inputs(B, n, seed=300 + n)
    x0 = x0 * np.array([0.3, 0.3, 0.5, 0.5])[:, None]
    sim = rgp.BatchedGameSim(B, n)
    sim.set_variant(3)
    st0 = sim.reset(dt, 1000, x0)
    P = 2 * n
    table = np.empty((K, P, 2, B))
    for e in range(P):                                             # constant pull towards the centre, strength from amp
        d = -st0[6 + 4 * e:8 + 4 * e]
        d = d / np.maximum(np.sqrt((d * d).sum(0)), 1e-9)
        table[:, e] = (amp[e] * 1.5 * d)[None]
    ref = oracle.batch_run(n, x0, table, K, dt=dt, winning_score=1000, n_threads=8, want_damage=True)
    got = sim.run_table(table, chunk_steps=K)
    c = sim.counters()
    assert_same(got, ref["state"], "split grid n=%d" % n)
    assert_same(sim.damage().reshape(P * P, B), ref["damage"], "damage")
    assert c["resolved"] == ref["stats"]["resolved"] and c["cap_hits"] == ref["stats"]["cap_hits"]
    assert ref["stats"]["resolved"] > 20 * B
    sim.close()
