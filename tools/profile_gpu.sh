#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel-trace stats + HBM-traffic PMC passes of bench.py, summarised
# into profiles/<tag>_<workload>.{txt,json} (tools/summarize_prof.py) -- copy them back through gpurun_out/.
# Usage: tools/profile_gpu.sh <tag> <workload> [bench args...]      e.g.  tools/profile_gpu.sh r02 chain
set -u
TAG=${1:-r02}; WL=${2:-chain}; shift 2 || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_${TAG}_${WL}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--workload $WL --steps 3 --warmup 1 --no-cpu-baseline --no-extras $*"
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $R/bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch -o fetch -- python $R/bench.py $ARGS > $OUT/bench_fetch.json 2> $OUT/fetch.err
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/write -o write -- python $R/bench.py $ARGS > $OUT/bench_write.json 2> $OUT/write.err
# effective clock of every kernel: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / duration -- recorded with the device's UUID in
# the summary so that a profile can be matched to the box it was taken on
timeout 900 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d $OUT/clock -o clock -- python $R/bench.py $ARGS > $OUT/bench_clock.json 2> $OUT/clock.err
python3 $R/tools/summarize_prof.py $OUT $OUT/summary > $OUT/summary.log 2>&1
# keep only the small summaries (the raw traces can be hundreds of MB)
mkdir -p $R/gpurun_out/profiles
cp $OUT/summary.txt $R/gpurun_out/profiles/${TAG}_${WL}.txt
cp $OUT/summary.json $R/gpurun_out/profiles/${TAG}_${WL}.json
cp $OUT/bench_trace.json $R/gpurun_out/profiles/${TAG}_${WL}_bench_under_rocprof.json
rm -rf $OUT/trace $OUT/fetch $OUT/write $OUT/clock
