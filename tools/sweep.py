#!/usr/bin/env python3
"""Measures the rates RGS_VARIANT_AUTO chooses by (run on the GPU box; ~13 minutes):

    python tools/sweep.py [dt ...] > gpurun_out/sweep_auto.json      (several outputs: tools/sweep_merge.py a.json b.json > all.json)

For every team size 1..7, every stepper variant (1 wave-per-arena, 2 one-lane hybrid, 3 two-lane hybrid,
4 pooled hybrid, 5 asynchronous pooled hybrid), at 32 768, 65 536 and 131 072 arenas (wave kernel also at 1 024, its launch-latency point), the C2-shaped
workload (bench.py --config c2; at its own dt = 0.01 and at the dt = 0.05 of C4/C5, where five times as many arena-steps
have events) is timed as one fused launch of 500 steps ("long") and of 20 steps after 5 warm-up
steps ("short", the tail-dominated regime).  tools/rates_to_header.py turns the JSON into
robo-game-sim_amd/csrc/auto_rates.h, which is what rgs_hip.hip reads; nothing measured is typed into the library.
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(n, variant, arenas, steps, warmup, dt):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu", "--n-per-team", str(n), "--variant", str(variant),
           "--arenas", str(arenas), "--steps", str(steps), "--warmup", str(warmup), "--dt", str(dt)]
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=600)
    if out.returncode != 0:
        return None
    lines = [l for l in out.stdout.decode().splitlines() if l.startswith("{")]
    return json.loads(lines[-1])["value"] if lines else None


def main():
    dts = [float(x) for x in sys.argv[1:]] or [0.01, 0.05]      # one step length per call keeps a call under the box's time limit
    res = {"unit": "arena-steps/s", "workload": "bench.py --config c2 shape at two step lengths", "long": [500, 10], "short": [20, 5],
           "dts": dts, "rates": []}
    for dt in res["dts"]:
        for n in range(1, 8):
            for variant in (1, 2, 3, 4, 5):
                sizes = (1024, 32768, 65536, 131072) if variant == 1 else (32768, 65536, 131072)
                for arenas in sizes:
                    for regime, (k, w) in (("long", res["long"]), ("short", res["short"])):
                        v = run(n, variant, arenas, k, w, dt)
                        res["rates"].append({"dt": dt, "n": n, "variant": variant, "arenas": arenas, "regime": regime, "value": v})
                        print("dt=%g n=%d variant=%d arenas=%d %s %s" % (dt, n, variant, arenas, regime, v), file=sys.stderr, flush=True)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
