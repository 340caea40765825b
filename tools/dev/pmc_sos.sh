# VALU / occupancy counters of the cascade kernel on cfg 2 (float64 default and precision=auto -> float32)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_sos
mkdir -p $OUT $R/gpurun_out/profiles
cd /tmp && export TMPDIR=/tmp
for wl in sos sos_auto; do
  prec=f64; [ $wl = sos_auto ] && prec=auto
  for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY"; do
    tag=$(echo $set | tr ' ' '_' | cut -c1-40)
    TORCHFX_AMD_IIR_PRECISION=$prec timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/${wl}_$tag -o p -- python $R/bench.py --workload sos --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $OUT/${wl}_$tag.json 2> $OUT/${wl}_$tag.err
  done
done
python3 - <<'PY' > $R/gpurun_out/profiles/r02_sos_pmc.txt
import csv, glob, os, collections
out = os.environ.get("GRAFT_REPO_ROOT", os.getcwd()) + "/gpurun_out/pmc_sos"
print("rocprofv3 --pmc passes of `bench.py --workload sos|sos_auto --steps 3 --warmup 1` (cfg 2: 64 x 2.88 M, 4 sections); per dispatch of tfx::sos_stream_kernel, mean over the 4 dispatches")
for wl in ("sos", "sos_auto"):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"{out}/{wl}_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "sos_stream_kernel" in r.get("Kernel_Name", ""):
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f"\n[{wl}]  " + ("float64 arithmetic, LC = 64, 2 waves per SIMD" if wl == "sos" else "float32 arithmetic (precision=auto), LC = 32 + register prefetch, 4 waves per SIMD"))
    m = {k: sum(v) / len(v) for k, v in acc.items()}
    for k in sorted(m):
        print(f"  {k:24s} {m[k]:16.0f}")
    if "SQ_WAVES" in m and "SQ_INSTS_VALU" in m:
        print(f"  -> VALU instructions per wave {m['SQ_INSTS_VALU'] / m['SQ_WAVES']:.0f}; per sample-row of 64 lanes: {m['SQ_INSTS_VALU'] * 64 / (64 * 2880000):.2f}")
    if "SQ_ACTIVE_INST_VALU" in m and "SQ_WAVE_CYCLES" in m:
        print(f"  -> SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES = {m['SQ_ACTIVE_INST_VALU'] / m['SQ_WAVE_CYCLES']:.3f} of a wave's lifetime issuing VALU")
    if "SQ_ACTIVE_INST_VALU" in m and "SQ_BUSY_CYCLES" in m:
        print(f"  -> SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES = {m['SQ_ACTIVE_INST_VALU'] / m['SQ_BUSY_CYCLES']:.3f}")
    if "SQ_WAIT_INST_ANY" in m and "SQ_WAVE_CYCLES" in m:
        print(f"  -> SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES = {m['SQ_WAIT_INST_ANY'] / m['SQ_WAVE_CYCLES']:.3f} of a wave's lifetime waiting on any instruction")
PY
rm -rf $OUT/sos_*/
