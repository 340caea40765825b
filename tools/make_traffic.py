#!/usr/bin/env python3
"""profiles/<tag>_<workload>.json (tools/profile_gpu.sh + tools/summarize_prof.py) -> profiles/<tag>_traffic.json,
the HBM-traffic file bench.py reports as roofline.traffic / step_traffic.

usage: make_traffic.py <tag> [<dir with the per-workload summaries, default profiles/>]

Bytes: FETCH_SIZE and WRITE_SIZE are reported in KiB by rocprofv3.  On gfx950 FETCH_SIZE tallies 64 B per
128-B request for wide coalesced streams (MI355X_MICROARCH.md, HBM section), so read bytes = 2 x FETCH_SIZE;
WRITE_SIZE is taken at face value.  Both factors are validated here against bytes that are known by
construction: the row pass reads and writes its workspace slab exactly once (model_check below).
The profiled command runs W warm-up steps, K timed steps and (since round 3) the same K steps once more with the library's
per-kernel events -- all of them seen by rocprofv3: per-step = totals / (W + 2 K), read from the bench line of the run."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def short(name: str) -> str | None:
    m = re.search(r"tfx::(\w+)", name)
    if not m:
        return None
    k = m.group(1)
    if k == "sos_stream_kernel":
        return "sos_stream_kernel<f64>" if "float, float, double" in name or ", double," in name else "sos_stream_kernel<f32>"
    return k


def main() -> None:
    tag = sys.argv[1]
    src = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles")
    import bench

    out = {"_comment": "HBM bytes per bench step from rocprofv3 PMC passes of `bench.py --workload W --no-extras` (FETCH_SIZE and "
                       "WRITE_SIZE in separate runs, tools/profile_gpu.sh). read = 2 x FETCH_SIZE (gfx950 tallies 64 B per 128-B request "
                       "for coalesced streams), write = WRITE_SIZE. Sources: profiles/%s_<workload>.{txt,json}." % tag,
           "source_digest": bench.source_digest()}
    for wl in ("chain", "chain_fold", "chain_iir_kernel", "sos", "fir", "fir_fft", "fftconv"):
        p = os.path.join(src, f"{tag}_{wl}.json")
        if not os.path.exists(p):
            continue
        d = json.load(open(p))
        under = json.loads(open(os.path.join(src, f"{tag}_{wl}_bench_under_rocprof.json")).read().strip().splitlines()[-1])
        STEPS = under["warmup"] + (2 if "ms_per_step_with_event_profiling" in under else 1) * under["steps"]
        per = {}
        for tagc, key in (("fetch", "read"), ("write", "write")):
            for r in d.get(tagc, []):
                k = short(r["kernel_name"])
                if k is None:
                    continue
                e = per.setdefault(k, {"read": 0.0, "write": 0.0, "launches_per_step": r["dispatches"] / STEPS})
                e[key] = round(r["per_dispatch"] * 1024 * (2.0 if tagc == "fetch" else 1.0) / 1e9, 4)
        times = {short(r["name"]): r["average"] for r in d.get("top_kernels", []) if short(r["name"])}
        for k, e in per.items():
            if k in times:
                e["avg_us_under_rocprofv3"] = round(times[k], 2)
        total = sum((e["read"] + e["write"]) * e["launches_per_step"] for e in per.values()) * 1e9
        seconds = 60.0 if wl in ("sos", "fir", "fir_fft") else 600.0
        alg = 8.0 * 64 * seconds * 48000
        out[wl] = {"seconds": seconds, "bytes_per_step": round(total), "algorithmic_bytes_per_step": alg,
                   "traffic_over_algorithmic": round(total / alg, 3), "per_kernel_GB_per_launch": per}
        row = per.get("ols_row4096_kernel") or per.get("ols_row8192_kernel")
        if row:       # the row pass touches its slab exactly once each way: known bytes
            ols = under["config"]["overlap_save"]
            frames = 64 * ols["blocks_per_row"]
            slab_gb = ((frames + 1) // 2) * ols["fft_block"] * 8 / row["launches_per_step"] / 1e9
            out[wl]["model_check"] = {"row_pass_slab_GB_per_launch": round(slab_gb, 4), "measured_read": row["read"],
                                      "measured_write": row["write"]}
    dst = os.path.join(ROOT, "profiles", f"{tag}_traffic.json")
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out, indent=1)[:3000])


if __name__ == "__main__":
    main()
