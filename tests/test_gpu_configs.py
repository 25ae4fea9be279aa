"""BASELINE configs C4 and C5 in their own shapes, through the generators bench.py uses, against the oracle.

C4 (py/mc_example_1.py:13-30, py/MonteCarlo.py:23-45): dt 0.05, winning score 4, x0_ball = (-1, 0, 0, 0),
one zero-command step then ceil(T/dt) = 200 commanded steps FROM RESET, run i's controls generated from
seed i (workloads.mc_inputs).  C5 (py/test_dmpc.py:43-47): 4v4, dt 0.05, a host-side controller loop:
every step uploads that step's commands and downloads the state (rgs_step_roundtrip, pinned buffers).
Both also run bench.py itself on a reduced arena count, including its self-started 2-rank form.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _c4_table(B, n, first_seed, steps=201, dt=0.05):
    """bench.py's C4 inputs: device-evaluated circle table (downloaded, so the oracle sees the same values)."""
    import torch
    from robo_game_py import workloads as w
    amp, ph = w.mc_inputs(np.arange(first_seed, first_seed + B), n)
    table = w.device_circle_table(torch, torch.device("cuda", 0), amp, ph, w.step_times(steps, dt), zero_first=True)
    host = table.cpu().numpy()
    del table
    return host


@pytest.mark.parametrize("win", [4, 1])
def test_c4_monte_carlo_shape_from_reset(rgp, oracle, win):
    """2048 MC runs x (1 + 200) steps; win = 1 makes many games end early (decided arenas must freeze)."""
    B, n = 2048, 2
    table = _c4_table(B, n, first_seed=1)
    assert np.all(table[0] == 0.0)
    x0 = np.repeat(np.array([[-1.0], [0.0], [0.0], [0.0]]), B, axis=1)
    if win == 1:                                               # a fast puck towards either goal mouth: most games end within ~15 steps
        rng = np.random.default_rng(4)
        x0[1] = rng.uniform(-0.7, 0.7, B)
        x0[2] = np.where(np.arange(B) % 2 == 0, 15.0, -15.0)
        x0[3] = rng.uniform(-0.5, 0.5, B)
    sim = rgp.BatchedGameSim(B, n)
    sim.reset(0.05, win, x0, 0.0, False, "", 0, 0.1, 0.5, 1.0, 0.5)
    got = sim.run_table(table, chunk_steps=201)                # one fused launch of all 201 steps
    und, dmg, t, cnt = sim.undecided(), sim.damage().reshape(16, B), sim.time(), sim.counters()
    sim.close()
    ref = oracle.batch_run(n, x0, table, 201, dt=0.05, winning_score=win, n_threads=8, want_damage=True)
    assert np.array_equal(got[:2], ref["state"][:2]), "scores differ"
    assert np.array_equal(got, ref["state"]), "max |delta| %g" % np.max(np.abs(got - ref["state"]))
    assert np.array_equal(dmg, ref["damage"])
    decided = ~((ref["state"][0] < win) & (ref["state"][1] < win))
    assert np.array_equal(und, ~decided)
    if win == 1:
        assert decided.sum() > 20, "the early-decided case is not exercised (%d decided)" % decided.sum()
        assert np.mean(t[decided] < 2.0) > 0.5                  # a decided game stops its clock (cpp:121-123)
        assert np.all(t[~decided] > 10.0)
    assert cnt["resolved"] == ref["stats"]["resolved"] and cnt["cap_hits"] == ref["stats"]["cap_hits"]


def test_c5_host_streamed_controls_4v4(rgp, oracle):
    """2048 x 4v4 arenas driven step by step from pinned host buffers; every downloaded state is checked."""
    from robo_game_py import workloads as w
    B, n, K = 2048, 4, 120
    P, S = 2 * n, rgp.state_size(n)
    x0, amp, ph = w.c2_inputs(B, n)
    table = w.circle_controls(amp, ph, w.step_times(K, 0.05))
    sim = rgp.BatchedGameSim(B, n)
    sim.reset(0.05, 1000000, x0)
    ctl_pin, st_pin = sim.pinned((K, P, 2, B)), sim.pinned((S, B))
    ctl_pin[...] = table
    every = 20
    snaps = []
    for k in range(K):
        sim.step_roundtrip(ctl_pin[k], st_pin)
        if (k + 1) % every == 0:
            snaps.append(np.array(st_pin, copy=True))
    cnt = sim.counters()
    sim.close()
    ref = oracle.batch_run(n, x0, table, K, dt=0.05, winning_score=1000000, rec_every=every, n_threads=8)
    assert np.array_equal(np.stack(snaps), ref["traj"]), "streamed states differ from the oracle"
    assert np.array_equal(snaps[-1], ref["state"])
    assert cnt["resolved"] == ref["stats"]["resolved"] and cnt["resolved"] > 100


def _bench(*argv):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), env=env, cwd=ROOT,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    lines = [l for l in out.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout.decode()[-2000:]
    return json.loads(lines[0])


def test_bench_c4_and_c5_lines_carry_checked_parity():
    d = _bench("--config", "c4", "--arenas", "4096", "--cpu-sample-arenas", "1024")
    assert d["steps"] == 201 and d["warmup"] == 0 and d["parity"]["bit_identical"] and d["parity"]["steps"] == 201
    d = _bench("--config", "c5", "--arenas", "4096", "--steps", "100", "--cpu-sample-arenas", "1024")
    assert d["parity"]["bit_identical"] and d["parity"]["host_streamed_bit_identical"]
    assert d["host_streamed"]["steps"] == 110


def test_bench_starts_its_own_ranks_and_checks_both_shards():
    """`python bench.py --gpus 2` with no launcher: two ranks share this box's one GPU (gloo carries the
    collective there; on a multi-GPU node the same code path runs over RCCL)."""
    d = _bench("--gpus", "2", "--backend", "gloo", "--force-device", "0", "--arenas", "8192", "--steps", "60",
               "--warmup", "5", "--cpu-sample-arenas", "1024")
    assert d["n_gpus"] == 2 and d["collective"]["ranks"] == 2 and d["collective"]["gathered_shape"] == [22, 16384]
    p = d["parity"]
    assert p["bit_identical"] and p["arenas_checked"] == 1024 and "8192..8703" in p["arena_ids"]
    assert d["cpu_baseline"]["value"] > 0


@pytest.mark.parametrize("config,extra", [("c4", []), ("c5", ["--steps", "60", "--no-streamed"])])
def test_sharded_configs_rank1_equals_a_single_process_run_of_its_arenas(config, extra):
    """The configs BASELINE shards over GPUs (C4: Monte-Carlo runs with per-run seeds lo+1..hi, py/MonteCarlo.py:23-45; C5:
    4v4 ensemble), two ranks on this box's one GPU (gloo): what rank 1 computed for its first 1024 arenas -- read from the
    all-gathered ensemble on rank 0, scores and with them undecided() included -- must be exactly what ONE process computes
    for the same global arenas (runs 4097.. resp. arenas 4096..), and both must equal the oracle."""
    two = _bench("--config", config, "--gpus", "2", "--backend", "gloo", "--force-device", "0", "--arenas", "4096",
                 "--cpu-sample-arenas", "2048", "--cpu-seconds", "0.5", *extra)
    assert two["n_gpus"] == 2 and two["collective"]["ranks"] == 2
    p2 = two["parity"]
    assert p2["bit_identical"] and p2["scores_equal"] and p2["arenas_checked"] == 2048 and "4096..5119" in p2["global_arena_ids"]
    one = _bench("--config", config, "--gpus", "1", "--arenas", "4096", "--arena-offset", "4096", "--ensemble-arenas", "8192",
                 "--cpu-sample-arenas", "1024", "--cpu-seconds", "0.5", *extra)
    p1 = one["parity"]
    assert p1["bit_identical"] and p1["arenas_checked"] == 1024 and p1["global_arena_ids"] == "4096..5119"
    assert p1["first_rank_states_sha256"] == p2["last_rank_states_sha256"]
    assert p1["first_rank_states_sha256"] != p2["first_rank_states_sha256"]          # different runs, different games
