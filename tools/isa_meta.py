#!/usr/bin/env python3
"""Register / scratch / spill metadata of the env kernels from a device-only assembly of ec_env.hip (dev build).
usage: python tools/isa_meta.py [extra -D flags]   (writes /tmp/isa/ec_env_dev.s)"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.makedirs("/tmp/isa", exist_ok=True)
out = "/tmp/isa/ec_env_dev.s"
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-function",
       "-Wno-unused-command-line-argument", "-DEC_DEV_BUILD", "-I" + ROOT + "/marl_amd/csrc", "-I" + ROOT + "/include",
       "--offload-device-only", "-S", "-o", out, ROOT + "/marl_amd/csrc/ec_env.hip"] + sys.argv[1:]
subprocess.check_call(cmd)
txt = open(out).read()
for b in txt.split("  - .agpr_count")[1:]:
    n = re.search(r"\.name:\s+(\S+)", b).group(1)
    name = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
    if "k_step" not in name and "k_rollout" not in name:
        continue
    name = re.sub(r"\(anonymous namespace\)::", "", name).split("(")[0]
    g = lambda k: re.search(r"\.%s:\s+(\d+)" % k, b).group(1)
    print("%-40s vgpr %s scratch %s B/lane sgpr_spill %s vgpr_spill %s" % (name, g("vgpr_count"), g("private_segment_fixed_size"),
                                                                          g("sgpr_spill_count"), g("vgpr_spill_count")))
