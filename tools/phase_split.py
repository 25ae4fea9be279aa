#!/usr/bin/env python3
"""Diagnostic: where a workload's cycles go, per kernel phase (stamps build).

usage: RGS_LIB_SUFFIX=_stamps python tools/phase_split.py <n_per_team> <arenas> <steps> [variant] [dt]
Build the stamps library first: make -C robo-game-sim_amd/csrc SUFFIX=_stamps STAMPS=1.  Uses the C2/C3/C5
workload generator (circle controls, rng 20260).  Prints cycles per wavefront-step of the lane pass, of the
wave pass and of its phases, and per wave-pass arena.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "robo-game-sim_amd"))
import robo_game_py as rgp  # noqa: E402
from robo_game_py import workloads as w  # noqa: E402


def main():
    n, B, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    variant = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    dt = float(sys.argv[5]) if len(sys.argv) > 5 else 0.01
    x0, amp, ph = w.c2_inputs(B, n)
    table = w.circle_controls(amp, ph, w.step_times(K + 10, dt))
    sim = rgp.BatchedGameSim(B, n)
    sim.set_variant(variant)
    sim.reset(dt, 1000000, x0)
    buf = sim.alloc(table.size)
    buf.upload(table)
    stride = 2 * n * 2 * B
    sim.step_device(buf.ptr, stride, 10)
    sim.sync()
    sim.stamps(True, clear=True)
    c0 = sim.counters()
    sim.timer_start()
    sim.step_device(buf.ptr + 10 * stride * 8, stride, K)
    ms = sim.timer_stop()
    st, c1 = sim.stamps(True), sim.counters()
    out = {"n": n, "arenas": B, "steps": K, "variant": variant, "ms": round(ms, 3),
           "arena_steps_per_s": round(B * K / ms * 1e3, 0),
           "flagged": c1["flagged"] - c0["flagged"], "event_steps": c1["event_steps"] - c0["event_steps"],
           "resolved": c1["resolved"] - c0["resolved"], "cap_hits": c1["cap_hits"] - c0["cap_hits"]}
    A = max(1, st["g0_arenas"])
    out["per_wave_pass_arena_cycles"] = {k: round(st[k] / A, 1) for k in ("g0_fill", "g0_scan0", "g0_events", "g0_commit")}
    out["wave_pass_arenas"] = st["g0_arenas"]
    E = max(1, st["ev_count"])
    out["per_event_cycles"] = {k: round(st[k] / E, 1) for k in ("ev_head", "ev_carry", "ev_scan")}
    out["events_timed"] = st["ev_count"]
    out["sum_cycles"] = {k: st[k] for k in ("lane_cycles", "wave_cycles", "total_cycles")}
    out["max"] = {k: st[k] for k in ("max_total_cycles", "max_wave_cycles", "max_rounds", "max_event_loop_cycles")}
    out["lane_share"] = round(st["lane_cycles"] / max(1, st["total_cycles"]), 4)
    out["wave_share"] = round(st["wave_cycles"] / max(1, st["total_cycles"]), 4)
    print(json.dumps(out))
    buf.free()
    sim.close()


if __name__ == "__main__":
    main()
