#!/usr/bin/env python3
"""Merge the outputs of several `tools/sweep.py <dt>` calls into one sweep file (stdout)."""
import json
import sys

parts = [json.load(open(f)) for f in sys.argv[1:]]
out = dict(parts[0])
out["dts"] = sorted({dt for p in parts for dt in p["dts"]})
out["rates"] = [r for p in parts for r in p["rates"]]
print(json.dumps(out, indent=1))
