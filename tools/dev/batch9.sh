mkdir -p gpurun_out/r2b9
timeout 120 ./tools/ubench/bin/xcd_pipeline 400 > gpurun_out/r2b9/xcd.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE WRITE_SIZE --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/r2b9/pmc -o pmc -- $GRAFT_REPO_ROOT/tools/ubench/bin/xcd_pipeline 400 > $GRAFT_REPO_ROOT/gpurun_out/r2b9/xcd_pmc.txt 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'PY' > gpurun_out/r2b9/pmc_summary.txt 2>&1
import sqlite3, glob
for f in glob.glob("gpurun_out/r2b9/pmc/**/*.db", recursive=True):
    con = sqlite3.connect(f)
    for r in con.execute("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection group by kernel_name, counter_name"):
        print(r[0][:50], r[1], r[2], f"{r[3]:.6g} KiB/dispatch", f"dur {r[4]/1e6:.3f} ms")
PY
rm -rf gpurun_out/r2b9/pmc
