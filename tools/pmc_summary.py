#!/usr/bin/env python3
"""Average PMC counter values per kernel from a rocprofv3 --pmc output directory (counter_collection csv)."""
import csv
import re
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        mm = re.search(r"\b(k_\w+(<[^>]*>)?)", r["Kernel_Name"])
        k = mm.group(1) if mm else r["Kernel_Name"].split("(")[0][-60:]
        c = acc[k][r["Counter_Name"]]
        c[0] += float(r["Counter_Value"])
        c[1] += 1
for k, cs in acc.items():
    print(k, {c: (v[0] / v[1], v[1]) for c, v in cs.items()})
