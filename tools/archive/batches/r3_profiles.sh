#!/bin/bash
# round 3: rocprofv3 kernel-trace stats + HBM-traffic PMC passes of every workload, PMC of pass B (LDS) and of the cascade kernel
cd ${GRAFT_REPO_ROOT:-.}
R=$(pwd); mkdir -p gpurun_out/profiles
for wl in chain chain_iir_kernel sos fir fftconv; do
  bash tools/profile_gpu.sh r03 $wl > gpurun_out/prof_$wl.log 2>&1
done
bash tools/pmc_generic.sh r03_row ols_row4096 --workload fftconv --no-extras > gpurun_out/profiles/r03_row_pmc.txt 2>&1
bash tools/pmc_generic.sh r03_sos sos_stream --workload sos --no-extras > gpurun_out/profiles/r03_sos_pmc_f64.txt 2>&1
TORCHFX_AMD_IIR_PRECISION=f32 bash tools/pmc_generic.sh r03_sos32 sos_stream --workload sos --no-extras > gpurun_out/profiles/r03_sos_pmc_f32.txt 2>&1
rm -rf gpurun_out/pmc_r03_row gpurun_out/pmc_r03_sos gpurun_out/pmc_r03_sos32 gpurun_out/prof_r03_*
ls -la gpurun_out/profiles | tail -20
tail -5 gpurun_out/profiles/r03_row_pmc.txt
