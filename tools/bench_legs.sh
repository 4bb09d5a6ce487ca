#!/bin/bash
# GPU box: the default bench line with every leg, summarised (and kept under gpurun_out/r04/)
TAG=${1:-x}
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:?run on the GPU box (gpurun sets GRAFT_REPO_ROOT)}
O=gpurun_out/r04; mkdir -p $O
timeout 900 python3 bench.py --no-cpu-baseline > $O/bench_full_$TAG.json 2>$O/bench_full_$TAG.err
python3 - <<PY
import json
d=json.load(open("$O/bench_full_$TAG.json"))
for k in ("fused_rollout","fused_actions","fused_compact","closed_loop","closed_loop_compact","cooking_policy"):
    v=d.get(k,{}); print(k, {kk: v[kk] for kk in v if kk in ("ms_per_step","us_per_step","us_per_launch","env_steps_per_s","env_steps_per_s_per_gpu","error")}, v.get("roofline",{}).get("frac"))
for k,v in d.get("configs",{}).items():
    print(k, v.get("per_step",{}).get("us_per_launch"), v.get("per_step",{}).get("roofline",{}).get("frac"), v.get("fused",{}).get("us_per_step")) if isinstance(v, dict) and "per_step" in v else print(k, v)
print("value", d["value"], d["ms_per_step"], d["roofline"]["kernel_us"], d["roofline"]["frac"], d["roofline"].get("overlapped_launch_interval_us"))
PY
