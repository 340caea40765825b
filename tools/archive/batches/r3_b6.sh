#!/bin/bash
# round 3, GPU batch 6: the driver's default bench line with the new fields
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
( time python bench.py ) > gpurun_out/r3_b6_bench.json 2> gpurun_out/r3_b6_bench.err
tail -3 gpurun_out/r3_b6_bench.err
python - <<'PY'
import json
l=json.loads(open("gpurun_out/r3_b6_bench.json").read().strip().splitlines()[-1])
for k in ("value","ms_per_step","ranks_seen","devices","end_to_end","roofline","cfg5_512ch_on_one_gpu","vs_baseline_context"):
    print(k, json.dumps(l.get(k))[:1500])
print("stages", {k:(v.get("ms_per_step"),v.get("frac")) for k,v in l["stages"].items()})
print("variants", {k:(v.get("ms_per_step")) for k,v in l["variants"].items()})
print("published", json.dumps(l["published_context"]["cases"])[:2500])
print("kernels_single_stream", l["kernels_single_stream"])
PY
