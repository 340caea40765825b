#!/bin/bash
# run every bench workload on the GPU box and print the one-line summaries (+ wall seconds)
R=${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p $R/gpurun_out
for w in chain sos fir fftconv; do
  s=$SECONDS
  python $R/bench.py --workload $w > $R/gpurun_out/bench_$w.json 2> $R/gpurun_out/bench_$w.err
  echo "$w wall $((SECONDS-s)) s"
done
python - <<'PY'
import json, os
R = os.environ.get("GRAFT_REPO_ROOT", ".")
for w in ("chain", "sos", "fir", "fftconv"):
    try:
        d = json.loads(open(f"{R}/gpurun_out/bench_{w}.json").read().strip().splitlines()[-1])
        print(w, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("step_frac"), json.dumps(d["cpu_baseline"])[:420])
    except Exception as e:
        print(w, "FAILED", e)
PY
