#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5_b9; mkdir -p $O
( time timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 ) 2>&1 | tee $O/pytest.txt
timeout 900 python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r5_b9/bench.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["config"]["workload"][:150])
print({k:(v.get("ms_per_step"), v.get("frac_of_8TBps_at_8B_per_sample"), v.get("max_abs_diff_vs_default_chain_first_4s"), v.get("error")) for k,v in d.get("variants",{}).items()})
print({k:(v.get("ms_per_step"), v.get("frac")) for k,v in d.get("stages",{}).items() if isinstance(v,dict)})
print(d["roofline"])
PY
