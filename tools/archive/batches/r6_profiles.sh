#!/bin/bash
# round 6: rocprofv3 kernel-trace stats + HBM-traffic + clock PMC passes of every workload (device UUID and effective clock recorded
# in each summary), PMC of the recursion-in-pass-A kernel alone and of the 16 384-point one-launch kernel, the default bench line, soak
cd ${GRAFT_REPO_ROOT:-.}
R=$(pwd); mkdir -p gpurun_out/profiles
for wl in chain chain_fold chain_iir_kernel sos fir fir_fft fftconv; do
  bash tools/profile_gpu.sh r06 $wl > gpurun_out/prof_$wl.log 2>&1
done
TFX_OLS_SOS_STREAMS=1 bash tools/pmc_cmd.sh r06_sosf col_fwd16_sos python $R/tools/sos_ols_bench.py 1 fused > /dev/null 2>&1
cp gpurun_out/pmc_r06_sosf/summary.txt gpurun_out/profiles/r06_sosf_pmc.txt
bash tools/pmc_cmd.sh r06_lds16k ols_lds16k python $R/tools/experiments/ols_time.py 8192 > /dev/null 2>&1
cp gpurun_out/pmc_r06_lds16k/summary.txt gpurun_out/profiles/r06_lds16k_w8_pmc.txt
timeout 900 python bench.py > gpurun_out/profiles/r06_bench_default.json 2> gpurun_out/profiles/r06_bench_default.err
timeout 900 python tools/soak.py 40 > gpurun_out/profiles/r06_soak.txt 2>&1
rm -rf gpurun_out/pmc_r06_* gpurun_out/prof_r06_*
ls -la gpurun_out/profiles | tail -40
