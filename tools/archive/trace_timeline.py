#!/usr/bin/env python3
"""Timeline of the LAST step in a rocprofv3 --kernel-trace database (rocpd sqlite): one line per dispatch of the
overlap-save kernels (start / end in microseconds relative to the step's first dispatch, stream / queue), then how
much of the step each kernel kind was running alone and together with others.
usage: trace_timeline.py <dir with *.db> [first_kernel_name_substring]"""
import glob
import sqlite3
import sys

d = sys.argv[1]
first = sys.argv[2] if len(sys.argv) > 2 else "col_fwd16"
db = sorted(glob.glob(f"{d}/**/*.db", recursive=True))[0]
con = sqlite3.connect(db)
names = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
view = "kernels" if "kernels" in names else None
if view is None:
    print("views:", names)
    sys.exit(1)
cols = [r[1] for r in con.execute(f"pragma table_info({view})")]
rows = con.execute(f"select name, start, end, queue_id, stream_id from {view} order by start").fetchall() if "stream_id" in cols else \
    [r + (0,) for r in con.execute(f"select name, start, end, queue_id from {view} order by start").fetchall()]
ols = [r for r in rows if "ols_" in r[0]]
# steps are separated by gaps > 1 ms between dispatches of the first pass
starts = [i for i, r in enumerate(ols) if first in r[0] and (i == 0 or r[1] - ols[i - 1][2] > 300_000)]
lo = starts[-1] if starts else 0
step = ols[lo:]
t0 = step[0][1]
short = lambda n: n.split("(")[0].replace("void tfx::", "")[:34]
print(f"{len(step)} dispatches in the last step, {(max(r[2] for r in step) - t0) / 1e3:.1f} us")
for r in step[:400]:
    print(f"{short(r[0]):36s} {(r[1] - t0) / 1e3:9.1f} {(r[2] - t0) / 1e3:9.1f} {(r[2] - r[1]) / 1e3:8.1f}  q{r[3]} s{r[4]}")
# coverage: sweep events
ev = []
for r in step:
    k = short(r[0])
    ev.append((r[1], 1, k)); ev.append((r[2], -1, k))
ev.sort()
active = {}
last = ev[0][0]
alone, mixed = {}, 0.0
for t, dlt, k in ev:
    dt = t - last
    live = [n for n, c in active.items() if c > 0]
    if dt > 0 and live:
        if len(live) == 1:
            alone[live[0]] = alone.get(live[0], 0.0) + dt
        else:
            mixed += dt
            key = "+".join(sorted(live))
            alone[key] = alone.get(key, 0.0) + dt
    active[k] = active.get(k, 0) + dlt
    last = t
print("\ntime by set of kernel kinds in flight (us):")
for k, v in sorted(alone.items(), key=lambda kv: -kv[1]):
    print(f"  {k:90s} {v / 1e3:9.1f}")
