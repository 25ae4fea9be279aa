"""Host-side mirror of the reference interface: argument handling, log record, layouts."""
import os
import struct
import sys

import numpy as np
import pytest


def test_reset_accepts_the_arities_the_reference_callers_use(rgp):
    # py/run_sim.py:43 passes 6 positionals, py/MonteCarlo.py:28-30 passes 11
    x0 = np.array([-1.0, 0, 0, 0])
    a6 = rgp.reset_arguments(0.05, 4, x0, 0.0, True, "minimal_game.log")
    assert a6["seed"] == 0 and a6["tau_puck"] == 0.1 and a6["tau_player"] == 0.5
    assert a6["player_mass"] == 1.0 and a6["puck_mass"] == 0.5 and a6["log"] is True
    a11 = rgp.reset_arguments(0.05, 4, x0, 0.0, True, "mc_run_1.log", 1, 0.1, 0.5, 1.0, 0.5)
    assert a11["seed"] == 1 and a11["logname"] == "mc_run_1.log"
    d = rgp.reset_arguments()                        # C++ defaults, src/GameSim.cpp:21-25
    assert (d["dt"], d["winning_score"], d["noise"], d["log"], d["logname"]) == (0.05, 3, 0.0, False, "~/gamelog.log")
    with pytest.raises(TypeError):
        rgp.reset_arguments(*range(12))
    with pytest.raises(TypeError):
        rgp.reset_arguments(0.05, dt=0.01)
    with pytest.raises(TypeError):
        rgp.reset_arguments(bogus=1)


def test_log_record_layout_matches_reference_bytes(rgp):
    # src/GameSim.cpp:161-170: t, TAS, TBS, puck xy, A1 xy, A2 xy, B1 xy, B2 xy, then four 4-double
    # blocks starting at flat column-major offsets 0,1,2,3 of the 4x4 damage matrix (utils.h:75).
    state = np.arange(22, dtype=np.float64) + 100
    dmg = np.arange(16, dtype=np.float64).reshape(4, 4)     # dmg[r, c] = 4r + c
    rec = rgp.log_record(1.25, state, dmg)
    vals = struct.unpack("=29d", rec)
    assert vals[:3] == (1.25, 100.0, 101.0)
    assert vals[3:5] == (102.0, 103.0)
    assert vals[5:13] == (106.0, 107.0, 110.0, 111.0, 114.0, 115.0, 118.0, 119.0)
    flat = dmg.flatten(order="F")
    assert vals[13:17] == tuple(flat[0:4])           # "row 0" = column 0
    assert vals[17:21] == tuple(flat[1:5])           # "row 1" = flat elements 1..4
    assert vals[25:29] == tuple(flat[3:7])


def test_state_view_offsets_match_reference_simstate(rgp):
    # py/src/SimState.py:6-12
    v = rgp.SimStateView(np.arange(22.0))
    assert (v.offset("A", 1), v.offset("A", 2), v.offset("B", 1), v.offset("B", 2)) == (6, 10, 14, 18)
    assert list(v.get_puck_state()) == [2, 3, 4, 5]
    assert list(v.get_player_state("B", 2)) == [18, 19, 20, 21]
    v7 = rgp.SimStateView(np.arange(62.0), 7)
    assert v7.offset("B", 1) == 6 + 4 * 7


def test_workloads_are_deterministic_and_shaped(rgp):
    from robo_game_py import workloads as w
    x0, amp, ph = w.c2_inputs(32, 2)
    x0b, ampb, phb = w.c2_inputs(32, 2)
    assert np.array_equal(x0, x0b) and np.array_equal(amp, ampb) and np.array_equal(ph, phb)
    assert x0.shape == (4, 32) and amp.shape == (4, 32)
    t = w.step_times(5, 0.01)
    assert t[0] == 0.0 and t[3] == 0.01 + 0.01 + 0.01
    tab = w.circle_controls(amp, ph, t)
    assert tab.shape == (5, 4, 2, 32) and tab.flags["C_CONTIGUOUS"]
    assert np.allclose(np.hypot(tab[:, :, 0], tab[:, :, 1]), amp[None])
    m = w.mindless_controls(4, 3, t)
    assert m.shape == (5, 4, 2, 3) and m[2, 1, 0, 2] == np.cos(t[2])
    r = w.random_hold_controls(4, 8, 45, 3.0, hold=20)
    assert r.shape == (45, 4, 2, 8) and np.abs(r).max() <= 3.0 and np.array_equal(r[0], r[19]) and not np.array_equal(r[19], r[20])


def test_partition_covers_everything_once(rgp):
    from robo_game_py.ensemble import partition
    for n, w_ in ((10, 3), (65536 * 8, 8), (7, 8), (1, 1)):
        parts = partition(n, w_)
        assert parts[0][0] == 0 and parts[-1][1] == n
        assert all(parts[i][1] == parts[i + 1][0] for i in range(w_ - 1))
        sizes = [hi - lo for lo, hi in parts]
        assert max(sizes) - min(sizes) <= 1


def test_bench_starts_one_rank_per_gpu_when_not_under_a_launcher(monkeypatch):
    """`python bench.py --gpus N` (N > 1, no WORLD_SIZE) must start N ranks itself, before touching the GPU."""
    import importlib
    import subprocess
    sys_path_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if sys_path_root not in sys.path:
        sys.path.insert(0, sys_path_root)
    bench = importlib.import_module("bench")
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7

    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "20", "--warmup", "5"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7                                   # the children's return code is passed on
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "20", "--warmup", "5"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"



def test_bench_presets_match_baseline_configs():
    import importlib
    bench = importlib.import_module("bench")
    a = bench.parse_args(["--config", "c4"])
    assert (a.n_per_team, a.arenas, a.dt, a.win, a.steps, a.warmup) == (2, 131072, 0.05, 4, 201, 0)
    a = bench.parse_args([])
    assert (a.n_per_team, a.arenas, a.dt, a.steps, a.warmup) == (2, 65536, 0.01, 1000, 10)
    a = bench.parse_args(["--config", "c5", "--steps", "20", "--warmup", "5"])
    assert (a.n_per_team, a.arenas, a.dt, a.steps, a.warmup) == (4, 32768, 0.05, 20, 5)
    assert bench.algorithmic_bytes_per_arena_step(2) == 416 and bench.algorithmic_bytes_per_arena_step(7) == 1216


def test_mc_inputs_are_a_function_of_the_run_seed_only():
    from robo_game_py import workloads as w
    a1, p1 = w.mc_inputs(np.arange(1, 9), 2)
    a2, p2 = w.mc_inputs(np.arange(5, 9), 2)
    assert np.array_equal(a1[:, 4:], a2) and np.array_equal(p1[:, 4:], p2)      # sharding the runs changes nothing
    assert a1.shape == (4, 8) and np.all((a1 >= 0.5) & (a1 < 4.0)) and np.all((p1 >= 0) & (p1 < 2 * np.pi))


def test_sim_state_view_has_the_reference_accessors(rgp):
    """py/src/SimState.py:4-43: constants and accessors, copies not views; N-generic."""
    st = np.arange(22, dtype=np.float64)
    v = rgp.SimStateView(st)
    assert (v.TAS, v.TBS, v.PK, v.A1, v.A2, v.B1, v.B2) == (0, 1, 2, 6, 10, 14, 18)
    assert np.array_equal(v.get_player_state("B", 1), [14, 15, 16, 17])
    assert np.array_equal(v.get_player_pos("A", 2), [10, 11]) and np.array_equal(v.get_player_vel("A", 2), [12, 13])
    assert np.array_equal(v.get_puck_state(), [2, 3, 4, 5]) and np.array_equal(v.get_puck_pos(), [2, 3])
    assert np.array_equal(v.get_puck_vel(), [4, 5]) and np.array_equal(v.transpose(), st)
    p = v.get_player_pos("A", 1)
    p[0] = -1.0
    st[6] = -2.0
    assert v.state[6] == 6.0                                   # deepcopy semantics in both directions
    v7 = rgp.SimStateView(np.arange(62, dtype=np.float64))     # 7v7
    assert v7.n == 7 and v7.A7 == 6 + 4 * 6 and v7.B1 == 6 + 4 * 7 and np.array_equal(v7.get_player_vel("B", 7), [60, 61])
    with pytest.raises(ValueError):
        v.get_player_state("A", 3)
    with pytest.raises(TypeError):
        rgp.SimStateView(np.zeros(23))


def test_auto_rate_header_is_what_the_committed_sweep_generates(tmp_path):
    """csrc/auto_rates.h (all RGS_VARIANT_AUTO knows) must be tools/rates_to_header.py's output for the committed sweep."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = os.path.join(root, "robo-game-sim_amd", "csrc", "auto_rates.h")
    before = open(hdr).read()
    try:
        out = subprocess.run([sys.executable, os.path.join(root, "tools", "rates_to_header.py")], stdout=subprocess.PIPE,
                             stderr=subprocess.PIPE, timeout=60)
        assert out.returncode == 0, out.stderr.decode()
        assert open(hdr).read() == before, "auto_rates.h is stale: re-run tools/rates_to_header.py and rebuild"
    finally:
        open(hdr, "w").write(before)
    assert "kAutoRate[kAutoDts][2][5][3][8]" in before and "kAutoDt[kAutoDts] = {0.01, 0.05}" in before


def test_bench_dt_override_keeps_the_presets():
    import bench
    a = bench.parse_args(["--config", "c2", "--dt", "0.05"])
    assert a.dt == 0.05 and a.n_per_team == 2 and a.arenas == 65536 and a.steps == 1000
    assert bench.parse_args(["--config", "c4"]).dt == 0.05 and bench.parse_args([]).dt == 0.01
