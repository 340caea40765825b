#!/bin/bash
# PMC passes for the direct-FIR MFMA kernel: effective clock and matrix-pipe busy.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_fir; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -io "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*" | sort -u > $OUT/mfma_counters.txt
cat $OUT/mfma_counters.txt
B="python $R/bench.py --workload fir --steps 2 --warmup 1 --no-cpu-baseline --no-extras"
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace -d $OUT/a -o a -- $B > $OUT/a.json 2> $OUT/a.err
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CU_CYCLES --kernel-trace -d $OUT/b -o b -- $B > $OUT/b.json 2> $OUT/b.err
timeout 600 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_SALU --kernel-trace -d $OUT/c -o c -- $B > $OUT/c.json 2> $OUT/c.err
python3 - "$OUT" <<'PY'
import sqlite3, glob, sys
for f in sorted(glob.glob(sys.argv[1]+"/*/*.db")):
    con=sqlite3.connect(f)
    try:
        rows=con.execute("select kernel_name,counter_name,count(*),avg(value),avg(duration) from counters_collection where kernel_name like '%fir_direct%' group by kernel_name,counter_name").fetchall()
        for r in rows: print(f.split('/')[-2], r[1], r[2], f"{r[3]:.6g}", f"dur_ns={r[4]:.0f}")
    except Exception as e: print(f, e)
PY
tail -2 $OUT/b.err
