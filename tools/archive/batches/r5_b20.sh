#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5_b20; mkdir -p $O
( time timeout 3300 python -m pytest tests -m gpu -q 2>&1 | tail -12 ) 2>&1 | tee $O/pytest.txt
timeout 1200 python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r5_b20/bench.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["config"]["workload"][:160])
print({k:(v.get("ms_per_step"), v.get("frac_of_8TBps_at_8B_per_sample"), v.get("max_abs_diff_vs_default_chain_first_4s"), v.get("fold_error_estimate"), v.get("error")) for k,v in d.get("variants",{}).items()})
print({k:(v.get("ms_per_step"), v.get("frac")) for k,v in d.get("stages",{}).items() if isinstance(v,dict)})
print(d["end_to_end"]); print(d["cfg5_512ch_on_one_gpu"].get("ms_per_step")); print(d["kernels"]); print(d["kernels_single_stream"])
PY
