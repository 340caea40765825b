#!/bin/bash
# round 3, batch 11: the direct FIR (cfg 3) with and without rocprofv3, clocks and power sampled beside it
cd ${GRAFT_REPO_ROOT:-.}
R=$(pwd); mkdir -p gpurun_out; OUT=$R/gpurun_out/r3_b11.log
smi() { /opt/rocm/bin/rocm-smi --showclocks --showpower --showperflevel 2>/dev/null | grep -E "sclk|mclk|fclk|Power|Performance" | tr -s ' ' | tr '\n' ';'; echo; }
sample() { while true; do echo "  [smi] $(smi)"; sleep 0.5; done; }
{
echo "idle: $(smi)"
for rep in 1 2; do
  echo "== plain run $rep"
  sample & SP=$!
  python bench.py --workload fir --steps 200 --warmup 5 --no-cpu-baseline --no-extras | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', l['ms_per_step'], 'frac', l['roofline']['frac'], {k:v['avg_ms_per_launch'] for k,v in l['kernels'].items()})"
  kill $SP; wait $SP 2>/dev/null
done
echo "== under rocprofv3 --kernel-trace --stats"
cd /tmp && export TMPDIR=/tmp
sample & SP=$!
rocprofv3 --kernel-trace --stats -d /tmp/prof_fir -o fir -- python $R/bench.py --workload fir --steps 200 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', l['ms_per_step'], 'frac', l['roofline']['frac'], {k:v['avg_ms_per_launch'] for k,v in l['kernels'].items()})"
kill $SP; wait $SP 2>/dev/null
find /tmp/prof_fir -name "*kernel_stats.csv" | head -1 | xargs -r head -5
cd $R
echo "== all-zero input (no data toggling), plain"
TFX_BENCH_ZERO_INPUT=1 python bench.py --workload fir --steps 200 --warmup 5 --no-cpu-baseline --no-extras | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', l['ms_per_step'], 'frac', l['roofline']['frac'])"
} > $OUT 2>&1
grep -v "^\s*\[smi\]" $OUT | head -40; echo; grep "\[smi\]" $OUT | awk 'NR%4==1' | head -30
