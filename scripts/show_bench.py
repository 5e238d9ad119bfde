#!/usr/bin/env python3
"""Pretty-print the JSON line of a bench.py log (last line): scripts/show_bench.py LOG"""
import json
import sys

d = json.loads([ln for ln in open(sys.argv[1]).read().splitlines() if ln.startswith('{"metric"')][-1])
ex = d.pop("extras", {})
print(json.dumps(d, indent=1))
print(json.dumps(ex, indent=1))
