bash tools/pmc_generic.sh olsB_map1 ols_row --workload fftconv --no-extras > gpurun_out/pmc_map1.txt 2>&1
TFX_OLS_ROWMAP=0 bash tools/pmc_generic.sh olsB_map0 ols_row --workload fftconv --no-extras > gpurun_out/pmc_map0.txt 2>&1
