"""Committed fixtures under tests/golden/ (see make_golden.py for provenance).

CPU: the oracle must still reproduce them (drift pin) and must match the reference's own sample
outputs in survey_kats.json.  GPU: the HIP stepper, through the C ABI, must reproduce both."""
import glob
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
NPZ = sorted(glob.glob(os.path.join(HERE, "golden", "*.npz")))
KATS = json.load(open(os.path.join(HERE, "golden", "survey_kats.json")))["cases"]
TOL = 1e-5


def same(a, b, what):
    a, b = np.asarray(a), np.asarray(b)
    err = float(np.max(np.abs(a - b))) if a.size else 0.0
    assert err <= TOL, "%s: max |delta| %g > %g" % (what, err, TOL)
    assert np.array_equal(a, b), "%s: not bit-identical (max |delta| %g)" % (what, err)


def test_fixtures_exist():
    assert len(NPZ) >= 6 and len(KATS) == 4


@pytest.mark.parametrize("path", NPZ, ids=[os.path.basename(p)[:-4] for p in NPZ])
def test_oracle_reproduces_golden(oracle, path):
    g = np.load(path)
    n, K = int(g["n"]), g["table"].shape[0]
    tp, tl, mp, mk = [float(v) for v in g["phys"]]
    ref = oracle.batch_run(n, g["x0"], g["table"], K, dt=float(g["dt"]), winning_score=int(g["win"]),
                           rec_every=int(g["rec_every"]), n_threads=4, want_damage=True, tau_puck=tp, tau_player=tl,
                           player_mass=mp, puck_mass=mk)
    same(ref["state"], g["state"], "state")
    same(ref["traj"], g["traj"], "trajectory")
    same(ref["damage"], g["damage"], "damage")
    st = ref["stats"]
    assert [st["resolved"], st["cap_hits"], st["skipped_after_goal"], st["base_overflow"]] == list(g["stats"])


def run_kat(make_sim, case):
    sim = make_sim()
    sim.reset(0.05, int(case["winning_score"]), np.array([float(v) for v in case["x0_ball"]]), 0.0, False, "", 0,
              0.1, 0.5, 1.0, 0.5)
    prev = None
    for step in case["steps"]:
        st = sim.run(*[np.array([float(c) for c in u]) for u in step["u"]])
        for idx, val in step["expect"].items():
            assert st[int(idx)] == float(val), (case["name"], idx, repr(st[int(idx)]), val)
        if "undecided" in step:
            assert sim.undecided() == step["undecided"]
        if step.get("same_as_previous"):
            assert np.array_equal(st, prev)
        prev = st


@pytest.mark.parametrize("case", KATS, ids=[c["name"] for c in KATS])
def test_oracle_matches_reference_samples(oracle, case):
    run_kat(oracle.OracleGameSim, case)


@pytest.mark.gpu
@pytest.mark.parametrize("case", KATS, ids=[c["name"] for c in KATS])
def test_gpu_matches_reference_samples(rgp, case):
    run_kat(rgp.GameSim, case)


@pytest.mark.gpu
@pytest.mark.parametrize("variant", [1, 2, 3, 4, 5], ids=["wave", "hybrid", "hybrid2", "pool", "apool"])
@pytest.mark.parametrize("path", NPZ, ids=[os.path.basename(p)[:-4] for p in NPZ])
def test_gpu_reproduces_golden(rgp, path, variant):
    g = np.load(path)
    n, table = int(g["n"]), g["table"]
    K, P, _, B = table.shape
    tp, tl, mp, mk = [float(v) for v in g["phys"]]
    sim = rgp.BatchedGameSim(B, n)
    sim.set_variant(variant)
    sim.reset(float(g["dt"]), int(g["win"]), g["x0"], 0.0, False, "", 0, tp, tl, mp, mk)
    rec = int(g["rec_every"])
    stride = P * 2 * B
    buf = sim.alloc(K * stride)
    buf.upload(table)
    traj = []
    for k0 in range(0, K, rec):
        sim.step_device(buf.ptr + 8 * k0 * stride, stride, rec)
        traj.append(sim.state())
    buf.free()
    same(np.stack(traj), g["traj"], "trajectory")
    same(sim.state(), g["state"], "state")
    same(sim.damage().reshape(P * P, B), g["damage"], "damage")
    c = sim.counters()
    assert [c["resolved"], c["cap_hits"], c["skipped_after_goal"], c["base_overflow"]] == list(g["stats"])
    sim.close()
