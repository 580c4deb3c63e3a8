#!/usr/bin/env python3
"""HBM bytes per launch per kernel family from the FETCH_SIZE / WRITE_SIZE PMC summaries of one round.

    python scripts/make_traffic.py <dir with pmc_FETCH_SIZE.summary.csv and pmc_WRITE_SIZE.summary.csv> [suffix] > traffic.json

traffic = (2 * FETCH_SIZE + WRITE_SIZE) KiB / dispatches: on gfx950 FETCH_SIZE counts half of the bytes of wide coalesced
reads (MI355X_MICROARCH.md, HBM section), WRITE_SIZE is exact.  Families are kernel names up to the first '<'.
"""
import csv
import json
import os
import re
import sys
from collections import defaultdict

root = sys.argv[1]
suffix = sys.argv[2] if len(sys.argv) > 2 else ""
fam = defaultdict(lambda: {"launches": 0, "fetch_kb_raw": 0.0, "write_kb": 0.0})
for name, col in (("FETCH_SIZE", "FETCH_SIZE"), ("WRITE_SIZE", "WRITE_SIZE")):
    path = os.path.join(root, f"pmc_{name}{suffix}.summary.csv")
    for r in csv.DictReader(open(path)):
        k = r["kernel"].split("<")[0]
        m = re.match(r"_ZN2ie\d+([a-z0-9_]+?)(I|E)", k)     # names the tool's demangler gave up on (_Float16 template arguments)
        if m:
            k = m.group(1)
        k = k.replace("ie::", "")
        if name == "FETCH_SIZE":
            fam[k]["launches"] += int(r["dispatches"])
            fam[k]["fetch_kb_raw"] += float(r[col])
        else:
            fam[k]["write_kb"] += float(r[col])
out = {}
for k, v in sorted(fam.items(), key=lambda kv: -(2 * kv[1]["fetch_kb_raw"] + kv[1]["write_kb"])):
    if v["launches"] == 0:
        continue
    out[k] = {"launches": v["launches"], "hbm_bytes_per_launch": int((2 * v["fetch_kb_raw"] + v["write_kb"]) * 1024 / v["launches"]),
              "fetch_kb_raw": v["fetch_kb_raw"], "write_kb": v["write_kb"],
              "note": "(2*FETCH_SIZE + WRITE_SIZE) KiB per launch; gfx950 FETCH_SIZE counts half of wide coalesced reads "
                      "(MI355X_MICROARCH.md HBM section), WRITE_SIZE exact"}
json.dump(out, sys.stdout, indent=1)
