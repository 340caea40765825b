#!/bin/bash
# round 5: rocprofv3 kernel-trace stats + HBM-traffic + clock PMC passes of every workload (device UUID and effective clock recorded
# in each summary), PMC of the recursion-in-pass-A kernel alone
cd ${GRAFT_REPO_ROOT:-.}
R=$(pwd); mkdir -p gpurun_out/profiles
for wl in chain chain_fold chain_iir_kernel sos fir fir_fft fftconv; do
  bash tools/profile_gpu.sh r05 $wl > gpurun_out/prof_$wl.log 2>&1
done
TFX_OLS_SOS_STREAMS=1 bash tools/pmc_cmd.sh r05_sosf col_fwd16_sos python $R/tools/sos_ols_bench.py 1 fused > /dev/null 2>&1
cp gpurun_out/pmc_r05_sosf/summary.txt gpurun_out/profiles/r05_sosf_pmc.txt
timeout 900 python bench.py --steps 20 --warmup 3 > gpurun_out/profiles/r05_bench_default.json 2> gpurun_out/profiles/r05_bench_default.err
rm -rf gpurun_out/pmc_r05_* gpurun_out/prof_r05_*
ls -la gpurun_out/profiles | tail -40
