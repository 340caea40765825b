#!/usr/bin/env python3
"""Turn rocprofv3 (ROCm 7.2, rocpd sqlite output) results into a small text + JSON summary.
usage: summarize_prof.py <prof_dir> <out_prefix>
  <prof_dir>/trace/*.db   from  rocprofv3 --kernel-trace --stats
  <prof_dir>/fetch/*.db   from  rocprofv3 --pmc FETCH_SIZE --kernel-trace
  <prof_dir>/write/*.db   from  rocprofv3 --pmc WRITE_SIZE --kernel-trace
HBM traffic correction (MI355X_MICROARCH.md section HBM): on gfx950 FETCH_SIZE counts 64 B per 128-B
request for wide coalesced streams, so read bytes = 2 x FETCH_SIZE x 1024; WRITE_SIZE x 1024 is taken
at face value (uncalibrated)."""
import glob
import json
import sqlite3
import sys


def q(db, sql):
    con = sqlite3.connect(db)
    try:
        cur = con.execute(sql)
        cols = [c[0] for c in cur.description]
        return [dict(zip(cols, r)) for r in cur.fetchall()]
    finally:
        con.close()


def main():
    d, out = sys.argv[1], sys.argv[2]
    res = {}
    t = glob.glob(f"{d}/trace/*.db")
    if t:
        res["top_kernels"] = q(t[0], "select name, total_calls, total_duration, average, percentage from top_kernels")
    for tag in ("fetch", "write"):
        f = glob.glob(f"{d}/{tag}/*.db")
        if f:
            res[tag] = q(f[0], "select kernel_name, counter_name, count(*) as dispatches, sum(value) as total, "
                               "avg(value) as per_dispatch, avg(duration) as avg_ns from counters_collection "
                               "group by kernel_name, counter_name order by total desc")
    c = glob.glob(f"{d}/clock/*.db")
    if c:
        rows = q(c[0], "select kernel_name, count(*) as dispatches, avg(value) as gui_active, avg(duration) as avg_ns from counters_collection "
                       "where counter_name = 'GRBM_GUI_ACTIVE' group by kernel_name order by sum(duration) desc")
        res["clock"] = [dict(r, ghz=round(r["gui_active"] / 8.0 / r["avg_ns"], 3)) for r in rows if r["avg_ns"]]
    import os
    for name in ("bench_trace.json", "bench_clock.json"):
        pth = os.path.join(d, name)
        try:
            line = json.loads(open(pth).read().strip().splitlines()[-1])
            dev = (line.get("devices") or [{}])[0]
            res.setdefault("device", {"name": dev.get("name"), "uuid": dev.get("uuid"), "cus": dev.get("cus")})
            res.setdefault("bench_ms_per_step_under_rocprofv3", {})[name.split(".")[0]] = line.get("ms_per_step")
        except Exception:
            pass
    json.dump(res, open(out + ".json", "w"), indent=1)
    with open(out + ".txt", "w") as fh:
        if "device" in res:
            fh.write(f"device: {res['device'].get('name')}  uuid {res['device'].get('uuid')}  CUs {res['device'].get('cus')}   "
                     f"bench ms/step under rocprofv3: {res.get('bench_ms_per_step_under_rocprofv3')}\n")
        if "clock" in res:
            fh.write("== effective clock per kernel: GRBM_GUI_ACTIVE / 8 XCDs / duration (separate --pmc pass) ==\n")
            for r in res["clock"][:8]:
                fh.write(f"{r['kernel_name'][:70]:70s} {r['dispatches']:6d} disp  {r['avg_ns'] / 1e3:10.2f} us  {r['ghz']:6.3f} GHz\n")
            fh.write("\n")
        fh.write("== rocprofv3 --kernel-trace --stats : per-kernel time (the view reports microseconds) ==\n")
        fh.write(f"{'kernel':70s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'%':>6s}\n")
        for r in res.get("top_kernels", []):
            fh.write(f"{r['name'][:70]:70s} {r['total_calls']:7d} {r['total_duration'] / 1e3:10.3f} "
                     f"{r['average']:10.2f} {r['percentage']:6.2f}\n")
        for tag in ("fetch", "write"):
            if tag not in res:
                continue
            fh.write(f"\n== rocprofv3 --pmc {tag.upper()}_SIZE (KiB as reported; separate pass) ==\n")
            fh.write(f"{'kernel':70s} {'disp':>6s} {'KiB/dispatch':>14s} {'GB/dispatch':>12s}\n")
            for r in res[tag]:
                mult = 2.0 if tag == "fetch" else 1.0
                fh.write(f"{r['kernel_name'][:70]:70s} {r['dispatches']:6d} {r['per_dispatch']:14.1f} "
                         f"{r['per_dispatch'] * 1024 * mult / 1e9:12.4f}{'  (x2 gfx950 correction)' if tag == 'fetch' else ''}\n")
    print(open(out + ".txt").read())


if __name__ == "__main__":
    main()
