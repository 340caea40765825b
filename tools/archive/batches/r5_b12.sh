#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5_b12; mkdir -p $O
( time timeout 3300 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 ) 2>&1 | tee $O/pytest.txt
# the float64 cascade with a third resident wave per SIMD (LC = 32: 167 VGPRs) against the shipping LC = 64 (2 waves)
for v in 4 0 4 0; do
  TFX_SOS_VARIANT=$v timeout 300 python bench.py --workload sos --steps 60 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('variant $v', d['ms_per_step'], d['roofline']['frac'], list(d['kernels'].items())[:1])"
done | tee $O/sos_variants.txt
for v in 4 0; do
  export TFX_SOS_VARIANT=$v
  bash tools/pmc_cmd.sh sos_v$v sos_stream python $GRAFT_REPO_ROOT/bench.py --workload sos --steps 3 --warmup 1 --no-extras --no-cpu-baseline
  cp gpurun_out/pmc_sos_v$v/summary.txt $O/pmc_sos_v$v.txt
done
