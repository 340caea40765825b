#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel-trace stats + HBM-traffic PMC passes for bench.py.
# Usage: tools/profile_gpu.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-extras $*"
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $R/bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch -o fetch -- python $R/bench.py $ARGS > $OUT/bench_fetch.json 2> $OUT/fetch.err
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/write -o write -- python $R/bench.py $ARGS > $OUT/bench_write.json 2> $OUT/write.err
cd $OUT
# keep only the small summaries (the raw traces can be hundreds of MB)
find . -name "*kernel_stats*" -o -name "*counter_collection*" | head -20
python3 - <<'PY'
import csv, glob, collections, json, os
out = {}
for f in glob.glob("trace/**/*kernel_stats*.csv", recursive=True):
    out["kernel_stats"] = list(csv.DictReader(open(f)))
for tag in ("fetch", "write"):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(f"{tag}/**/*counter_collection*.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name", "?")
            agg[k][0] += 1
            agg[k][1] += float(row.get("Counter_Value", 0) or 0)
    out[tag] = {k: {"dispatches": v[0], "sum": v[1], "per_dispatch": v[1] / max(1, v[0])} for k, v in agg.items()}
json.dump(out, open("summary.json", "w"), indent=1)
print(json.dumps(out)[:3000])
PY

du -sh . ; ls -R . | head -40
