#!/usr/bin/env python3
"""Summarise the rocprofv3 output of tools/prof.sh <tag> <bench args> into gpurun_out/prof/<tag>/summary.json.

Picks the TIMED launch of the stepping kernel (the last k_step_* dispatch: bench.py fuses all timed steps into
one launch and profiling runs skip the host-streamed leg) and records its duration and counters, keyed the way
bench.py looks profiles up (profiles/r03/pmc.json): "<config>:n<n>:B<arenas>:K<steps>:W<warmup>:L<launches>:v<variant>".
"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (argument presets only; nothing here touches the GPU)


def last_step_dispatch(rows, name_key):
    ids = [int(r["Dispatch_Id"]) for r in rows if "k_step_" in r[name_key]]
    return max(ids) if ids else None


def main():
    tag = sys.argv[1]
    args = bench.parse_args(sys.argv[2:])
    O = os.path.join(ROOT, "gpurun_out", "prof", tag)
    launches = -(-args.steps // (args.steps_per_launch if args.steps_per_launch > 0 else max(args.steps, 1)))
    key = "%s:n%d:B%d:K%d:W%d:L%d:v%d" % (args.config, args.n_per_team, args.arenas, args.steps, args.warmup, launches,
                                          args.variant)
    out = {"key": key, "tag": tag, "bench_args": sys.argv[2:]}
    for f in glob.glob(O + "/stats/**/*kernel_trace.csv", recursive=True):
        rows = list(csv.DictReader(open(f)))
        d = last_step_dispatch(rows, "Kernel_Name")
        for r in rows:
            if int(r["Dispatch_Id"]) == d:
                out["kernel"] = r["Kernel_Name"].split("(")[0]
                out["timed_launch_ms"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
                out["grid"] = r.get("Grid_Size", r.get("Grid_Size_X"))
                out["workgroup"] = r.get("Workgroup_Size", r.get("Workgroup_Size_X"))
                for k in ("LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count"):
                    if k in r:
                        out[k] = r[k]
    for f in glob.glob(O + "/stats/**/*kernel_stats.csv", recursive=True):
        out["kernel_stats"] = [r for r in csv.DictReader(open(f))][:6]
    for sub in ("pmc_sq", "pmc_fetch", "pmc_write"):
        for f in glob.glob(O + "/" + sub + "/**/*counter_collection.csv", recursive=True):
            rows = list(csv.DictReader(open(f)))
            d = last_step_dispatch(rows, "Kernel_Name")
            for r in rows:
                if int(r["Dispatch_Id"]) == d:
                    name = r["Counter_Name"]
                    name = name + "_KB" if name in ("FETCH_SIZE", "WRITE_SIZE") else name
                    out[name] = out.get(name, 0.0) + float(r["Counter_Value"])
    json.dump(out, open(os.path.join(O, "summary.json"), "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
