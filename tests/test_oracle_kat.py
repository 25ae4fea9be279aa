"""Pins the CPU oracle to the only reference outputs available in this environment:
the sample values the survey session printed from the compiled reference (SURVEY.md 8c,
"Sample oracle values recorded this session").  reset(0.05,3,x0,0,false,"",0,.1,.5,1,.5)."""
import numpy as np

Z = np.zeros(2)
RESET = dict(dt=0.05, winning_score=3, noise=0.0, log=False, logname="", seed=0, tau_puck=0.1,
             tau_player=0.5, player_mass=1.0, puck_mass=0.5)


def test_one_step_from_rest(oracle):
    s = oracle.OracleGameSim()
    s.reset(x0_ball=np.zeros(4), **RESET)
    st = s.run(np.array([1.0, 0.0]), Z, Z, Z)
    assert st[6] == float("-2.4975812909820143")
    assert st[7] == 1.25
    assert st[8] == float("0.095162581964028339")
    assert st[9] == 0.0


def test_goal_scores_and_zeroes_puck(oracle):
    s = oracle.OracleGameSim()
    s.reset(x0_ball=np.array([4.5, 0.2, 5.0, 0.0]), **RESET)
    st = s.run(Z, Z, Z, Z)
    assert list(st[2:6]) == [float("4.7379064549100676"), 0.2, float("4.5241870901798578"), 0.0]
    assert st[0] == 0 and st[1] == 0
    st = s.run(Z, Z, Z, Z)
    assert st[0] == 1 and st[1] == 0
    assert list(st[2:6]) == [0.0, 0.0, 0.0, 0.0]


def test_bounce_outside_goal_mouth(oracle):
    s = oracle.OracleGameSim()
    s.reset(x0_ball=np.array([4.5, 1.0, 5.0, 0.0]), **RESET)
    s.run(Z, Z, Z, Z)
    st = s.run(Z, Z, Z, Z)
    assert list(st[2:6]) == [float("4.6968234197802463"), 1.0, float("-4.093653765390016"), 0.0]
    assert st[0] == 0 and st[1] == 0


def test_decided_game_is_frozen(oracle):
    s = oracle.OracleGameSim()
    kw = dict(RESET, winning_score=1)
    s.reset(x0_ball=np.array([4.5, 0.2, 5.0, 0.0]), **kw)
    s.run(Z, Z, Z, Z)
    st1 = s.run(Z, Z, Z, Z)
    assert not s.undecided()
    t1 = s.time()
    st2 = s.run(np.array([3.0, 3.0]), Z, Z, Z)
    assert np.array_equal(st1, st2) and s.time() == t1


def test_reset_layout_matches_reference(oracle):
    # src/GameSim.cpp:34-38: players at (-+X/4, +-Y/4), puck at x0_ball, scores 0
    s = oracle.OracleGameSim()
    st = s.reset(0.05, 3, np.array([1.0, 2.0, 3.0, 4.0]))
    assert list(st) == [0, 0, 1, 2, 3, 4, -2.5, 1.25, 0, 0, -2.5, -1.25, 0, 0, 2.5, 1.25, 0, 0, 2.5, -1.25, 0, 0]
