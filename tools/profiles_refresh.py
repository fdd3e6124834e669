#!/usr/bin/env python3
"""Copy the summaries of a tools/profile_round.sh collection (gpurun_out/r03/prof_<tag>/) into profiles/r03/.
usage: python tools/profiles_refresh.py <tag>"""
import ast
import glob
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(ROOT, "gpurun_out", "r03", "prof_" + sys.argv[1]) + "/"
P = os.path.join(ROOT, "profiles", "r03") + "/"
for m, fn in {"rollout": "pmc_traffic_k_step_rollout.json", "step": "pmc_traffic_k_step.json", "mlp": "pmc_traffic_k_rollout_mlp.json"}.items():
    line = [l for l in open(O + "traffic_%s.txt" % m) if l.startswith("{")][-1]
    d = json.loads(line)
    json.dump(d, open(P + fn, "w"), indent=1)
    print(m, d["note"][-60:])
old = json.load(open(P + "sq_issue_k_step_rollout.json"))
cnt = {}
for l in open(O + "sq_rollout.txt"):
    mm = re.match(r"^(k_step<[^>]*>) (\{.*\})$", l.strip())
    if mm:
        for k, v in ast.literal_eval(mm.group(2)).items():
            cnt[k] = v[0]
steps, simds = 100, 1024
new = dict(old)
new["counters"] = cnt
new["per_simd_per_env_step"] = {"instructions_valu": cnt["SQ_INSTS_VALU"] / steps / simds, "instructions_salu": cnt["SQ_INSTS_SALU"] / steps / simds,
                                "instructions_lds": cnt["SQ_INSTS_LDS"] / steps / simds, "instructions_vmem": cnt["SQ_INSTS_VMEM"] / steps / simds,
                                "instructions_branch": cnt["SQ_INSTS_BRANCH"] / steps / simds}
new["issue_busy_fraction_of_wave_cycles"] = cnt["SQ_ACTIVE_INST_ANY"] / cnt["SQ_WAVE_CYCLES"]
new["waiting_fraction_of_wave_cycles"] = cnt["SQ_WAIT_ANY"] / cnt["SQ_WAVE_CYCLES"]
new["icache_hit_rate"] = cnt["SQC_ICACHE_HITS"] / cnt["SQC_ICACHE_REQ"]
new["wave_quad_cycles_per_env_step_per_wave"] = cnt["SQ_WAVE_CYCLES"] / 2048 / steps
json.dump(new, open(P + "sq_issue_k_step_rollout.json", "w"), indent=1)
print(new["per_simd_per_env_step"])
shutil.copy(glob.glob(O + "bench/**/*kernel_stats.csv", recursive=True)[0], P + "bench_kernel_stats.csv")
d = json.loads([l for l in open(O + "bench.out") if l.startswith("{")][-1])
json.dump(d, open(P + "bench_n1.json", "w"), indent=1)
print("value %.4g us/env-step %.2f frac %.4f kernel us/launch %.1f ppo %s" % (
    d["value"], d["config"]["us_per_env_step"], d["roofline"]["frac"], d["roofline"]["kernel_avg_us"],
    {k: round(v, 2) for k, v in d["ppo_iteration"].items() if k.endswith("_ms")}))
