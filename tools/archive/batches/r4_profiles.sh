#!/bin/bash
# round 4: rocprofv3 kernel-trace stats + HBM-traffic PMC passes of every workload, PMC of the one-launch LDS overlap-save kernel
cd ${GRAFT_REPO_ROOT:-.}
R=$(pwd); mkdir -p gpurun_out/profiles
for wl in chain chain_iir_kernel sos fir fir_fft fftconv; do
  bash tools/profile_gpu.sh r04 $wl > gpurun_out/prof_$wl.log 2>&1
done
bash tools/pmc_generic.sh r04_lds ols_lds4096 --workload fir_fft --no-extras > gpurun_out/profiles/r04_lds_pmc.txt 2>&1
bash tools/pmc_generic.sh r04_sos sos_stream --workload sos --no-extras > gpurun_out/profiles/r04_sos_pmc.txt 2>&1
bash tools/pmc_generic.sh r04_row ols_row4096 --workload fftconv --no-extras > gpurun_out/profiles/r04_row_pmc.txt 2>&1
rm -rf gpurun_out/pmc_r04_* gpurun_out/prof_r04_*
ls -la gpurun_out/profiles | tail -30
