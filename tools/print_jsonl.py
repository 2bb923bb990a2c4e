#!/usr/bin/env python3
"""print_jsonl.py FILE key [key ...]: one line per JSON record with the named fields (floats in %.3g)."""
import json
import sys

for line in open(sys.argv[1]):
    line = line.strip()
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    print(" ".join(("%.3g" % d[k]) if isinstance(d.get(k), float) else str(d.get(k)) for k in sys.argv[2:]))
