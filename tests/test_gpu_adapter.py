"""INTEGRATION.md "Option B" end to end: the C++ adapter class over the C ABI (examples/adapter_cpp), built with g++ against
include/rgs_hip.h and librgs_hip.so, plays a logged, noisy game; state and log file must be the oracle's."""
import math
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "robo-game-sim_amd", "csrc")
SRC = os.path.join(ROOT, "examples", "adapter_cpp", "gamesim_hip_adapter.cpp")


def build(tmp_path):
    exe = str(tmp_path / "adapter")
    cmd = ["g++", "-O2", "-I", os.path.join(ROOT, "include"), SRC, "-o", exe, "-L", CSRC, "-lrgs_hip", "-Wl,-rpath," + CSRC]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    return exe


def test_header_is_plain_c(tmp_path):
    """include/rgs_hip.h is a C header (the boundary is a C ABI): it compiles as C99 with -pedantic."""
    src = tmp_path / "c99.c"
    src.write_text('#include "rgs_hip.h"\nint main(void) { return rgs_state_size(2) == 22 ? 0 : 1; }\n')
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"),
                        "-fsyntax-only", str(src)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("noise", [0.0, 0.3])
def test_cpp_adapter_plays_the_oracles_game(rgp, oracle, tmp_path, noise):
    if noise != 0.0 and not rgp.BatchedGameSim.noise_is_exact():
        pytest.skip("built without the host's log tables: the handle's default noise mode is host tapes, exact as well; "
                    "this test pins the device generator")
    exe = build(tmp_path)
    log, seed, steps = str(tmp_path / "adapter.log"), 7, 80
    r = subprocess.run([exe, log, repr(noise), str(seed), str(steps)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    vals = [float(v) for v in r.stdout.split()]
    got, open_game = np.array(vals[:22]), int(vals[22])
    o = oracle.OracleGameSim()
    ref_log = str(tmp_path / "oracle.log")
    o.reset(0.05, 2, np.array([3.5, 0.2, 6.0, 0.0]), noise, True, ref_log, seed, 0.1, 0.5, 1.0, 0.5)
    st = o.run(np.zeros(2), np.zeros(2), np.zeros(2), np.zeros(2))
    t = 0.0
    for _ in range(steps):
        u = np.array([math.cos(t), math.sin(t)])
        st = o.run(u, u, u, u)
        t += 0.05
    und = o.undecided()
    o.reset(0.05, 2, np.zeros(4))
    assert np.array_equal(got, st), np.max(np.abs(got - st))
    assert open_game == int(und)
    want = open(ref_log, "rb").read()
    assert len(want) >= 29 * 8 * 10 and open(log, "rb").read() == want
