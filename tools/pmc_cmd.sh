#!/bin/bash
# usage: pmc_cmd.sh <tag> <kernel-like> <command ...>   -- three PMC passes of an arbitrary command (separate passes, --kernel-trace only)
TAG=$1; LIKE=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace -d $OUT/a -o a -- "$@" > $OUT/a.out 2> $OUT/a.err
timeout 600 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace -d $OUT/b -o b -- "$@" > $OUT/b.out 2> $OUT/b.err
timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --kernel-trace -d $OUT/c -o c -- "$@" > $OUT/c.out 2> $OUT/c.err
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/d -o d -- "$@" > $OUT/d.out 2> $OUT/d.err
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/e -o e -- "$@" > $OUT/e.out 2> $OUT/e.err
python3 - "$OUT" "$LIKE" <<'PY' | tee $OUT/summary.txt
import sqlite3, glob, sys
out, like = sys.argv[1], sys.argv[2]
for f in sorted(glob.glob(out+"/*/*.db")):
    con=sqlite3.connect(f)
    try:
        rows=con.execute(f"select kernel_name,counter_name,count(*),avg(value),avg(duration) from counters_collection where kernel_name like '%{like}%' group by kernel_name,counter_name").fetchall()
        for r in rows: print(f.split('/')[-2], r[0][:48], r[1], r[2], f"{r[3]:.5g}", f"dur_ns={r[4]:.0f}")
    except Exception as e: print(f, e)
PY
