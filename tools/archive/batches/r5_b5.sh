#!/bin/bash
# PMC of the cascade-in-pass-A kernel alone on one lane, two workgroups per CU
cd "$GRAFT_REPO_ROOT" || exit 1
export TFX_OLS_SOS_STREAMS=1 TFX_OLS_SOS_PAIRS=512
bash tools/pmc_cmd.sh sosf col_fwd16_sos python $GRAFT_REPO_ROOT/tools/sos_ols_bench.py 1 fused
cp gpurun_out/pmc_sosf/summary.txt gpurun_out/r5_b5_pmc_sosf.txt
