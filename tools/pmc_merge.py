#!/usr/bin/env python3
"""Merge gpurun_out/prof/*/summary.json (tools/prof.sh) into profiles/r03/pmc.json, keyed for bench.py's lookup,
and copy each summary to profiles/r03/prof_<tag>.json."""
import glob
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(ROOT, "profiles", "r03", "pmc.json")
os.makedirs(os.path.dirname(dst), exist_ok=True)
table = json.load(open(dst)) if os.path.exists(dst) else {}
for f in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "prof", "*", "summary.json"))):
    s = json.load(open(f))
    need = ("FETCH_SIZE_KB", "WRITE_SIZE_KB", "SQ_INSTS_VALU", "SQ_WAVES")
    if all(k in s for k in need):
        table[s["key"]] = {k: s[k] for k in s if k not in ("kernel_stats", "bench_args", "key")}
    shutil.copy(f, os.path.join(ROOT, "profiles", "r03", "prof_%s.json" % s["tag"]))
json.dump(table, open(dst, "w"), indent=1, sort_keys=True)
print("\n".join(sorted(table)))
