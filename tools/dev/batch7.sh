for wl in chain sos fir fftconv chain_iir_kernel; do bash tools/profile_gpu.sh r02 $wl; done
ls -la gpurun_out/profiles
