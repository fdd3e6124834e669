#!/usr/bin/env python3
"""Turns the FETCH_SIZE / WRITE_SIZE passes of tools/pmc_traffic.sh into the record bench.py reads
(profiles/rNN/pmc_traffic_*.json): bytes per launch, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950."""
import csv
import glob
import json
import os
import re
import sys

out_dir, mode = sys.argv[1], sys.argv[2]
# (name prefixes: the instantiations of the default geometry carry the block's lane count as a last template argument)
want = {"rollout": "k_step<4, 4, true", "step": "k_step<4, 4, false", "mlp": "k_rollout_mlp<4, 4, false, 1"}[mode]
vals = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = []
    for f in glob.glob(os.path.join(out_dir, "%s_%s" % (mode, c), "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r"\b(k_\w+(<[^>]*>)?)", r["Kernel_Name"])
            if m and m.group(1).startswith(want) and r["Counter_Name"] == c:
                kernel_name = m.group(1)
                acc.append(float(r["Counter_Value"]))
    if mode != "step":
        acc = acc[1:] if len(acc) > 1 else acc  # the first launch also pages the state in
    vals[c] = (sum(acc) / len(acc), len(acc))
fetch_raw = vals["FETCH_SIZE"][0] * 1024.0
write = vals["WRITE_SIZE"][0] * 1024.0
steps = 1 if mode == "step" else 100
rec = {"kernel": kernel_name, "workload": "tools/prof_step.py (4096 envs x 10 agents), mode %s, %d env steps per launch" % (mode, steps),
       "FETCH_SIZE_KB_avg": vals["FETCH_SIZE"][0], "WRITE_SIZE_KB_avg": vals["WRITE_SIZE"][0],
       "launches": vals["FETCH_SIZE"][1], "env_steps_per_launch": steps,
       "fetch_bytes_raw": fetch_raw, "write_bytes": write, "fetch_bytes_corrected": 2 * fetch_raw,
       "traffic_bytes_per_launch": 2 * fetch_raw + write,
       "note": "separate --pmc passes; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B read requests at "
               "64 B); per env step: raw %.1f MB, corrected %.1f MB" % ((fetch_raw + write) / steps / 1e6,
                                                                        (2 * fetch_raw + write) / steps / 1e6)}
name = {"rollout": "pmc_traffic_k_step_rollout.json", "step": "pmc_traffic_k_step.json", "mlp": "pmc_traffic_k_rollout_mlp.json"}[mode]
json.dump(rec, open(os.path.join(out_dir, name), "w"), indent=1)
print(json.dumps(rec))
