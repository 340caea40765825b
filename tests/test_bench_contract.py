"""bench.py pieces that run without a GPU: the CPU baseline leg and the filter set of the workload."""
import json

import numpy as np


def test_cpu_baseline_fields_and_workload_filters():
    import bench
    f1, f2, fir, rev = bench.build_filters()
    assert f1._sos.shape == (3, 6) and f2._sos.shape == (1, 6)            # 4 sections fused
    assert fir.kernel.numel() == 1024 and rev.kernel.numel() == 65536
    ir = bench.reverb_ir()
    assert abs(float(np.abs(ir).sum()) - 1.0) < 1e-5 and ir.dtype == np.float32
    cb = bench.cpu_baseline("chain", 1.0, 2)
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb
    assert cb["kind"] == "port" and 1 <= cb["cores"] <= 2 and cb["value"] > 0 and cb["single_thread_value"] > 0
    json.dumps(cb)


def test_group_stats_of_the_stage_timer():
    """Secondary figures: min / median / max over the groups, a slow FIRST group (and only that) dropped and named."""
    import bench
    st = bench.group_stats([0.3596, 0.3326, 0.8002, 0.3628, 0.3319])            # round 5's bimodal cfg-2 groups: nothing dropped
    assert st == {"min": 0.3319, "median": 0.3596, "max": 0.8002, "groups_used": 5, "dropped_first_group_ms": None}
    st = bench.group_stats([0.81, 0.33, 0.34, 0.33, 0.35, 0.33, 0.34, 0.33, 0.33])
    assert st["dropped_first_group_ms"] == 0.81 and st["groups_used"] == 8 and st["max"] == 0.35 and st["median"] == 0.33
    assert bench.group_stats([9.4, 9.3, 9.5])["groups_used"] == 3

    class Lib:
        def tfx_prof_enable(self, on):
            pass

        def tfx_prof_collect(self):
            return b"{}"
    ms, groups, kern, out = bench.batch_timed(lambda: 1, lambda: None, Lib(), 5, 5, min_group_ms=0.05)
    assert len(groups) >= 9 and set(bench.batch_timed.stats) == {"min", "median", "max", "groups_used", "dropped_first_group_ms"}
    assert ms == bench.batch_timed.stats["median"] and out == 1


def test_traffic_file_is_consistent():
    import os
    from tests.conftest import ROOT
    import bench
    tr = json.load(open(os.path.join(ROOT, bench.TRAFFIC_FILE)))
    assert len(tr["source_digest"]) == 16
    for wl in ("chain", "sos", "fir", "fftconv"):
        c = tr[wl]
        per = c["per_kernel_GB_per_launch"]
        total = sum((v["read"] + v["write"]) * v.get("launches_per_step", 1) for v in per.values())
        assert abs(total * 1e9 - c["bytes_per_step"]) / c["bytes_per_step"] < 0.03
        assert c["bytes_per_step"] > c["algorithmic_bytes_per_step"]
    # the row pass reads and writes its workspace slab exactly once: the counter corrections reproduce known bytes
    mc = tr["chain"]["model_check"]
    assert abs(mc["measured_write"] / mc["row_pass_slab_GB_per_launch"] - 1) < 0.02
    # ... plus the spectrum rows: once per slab with the XCD-aware row map (1 GB slabs, round 2: < 1.05), once per frame pair with
    # the plain map that wins for cache-sized slabs (8 MB of spectrum per 8-pair slab and what L2 loses in between: ~1.23)
    # Tight bounds per row-map mode (advisor, round 3): a launch reads its slab once plus the 8.4 MB spectrum at least once
    # (XCD-aware map, PMC round 4: slab + 1.08 spectra) and at most ~2.2 times (plain map, the default for 8-pair slabs: every
    # spectrum row is re-fetched by about every fourth of the 8 pairs that use it); anything above means workspace re-reads.
    # Round 5: the default chain runs 2^21-point blocks in slabs of 160 frame pairs with the XCD-aware row map: a launch reads its
    # slab once plus the 16.8 MB spectrum about once (PMC: 1.1 spectra); the plain pipeline (fftconv / the fold) keeps 8-pair slabs.
    slab = mc["row_pass_slab_GB_per_launch"]
    spectrum = 256 * 8192 * 8 / 1e9
    pairs = round(slab * 1e9 / (256 * 8192 * 8))
    extra = (mc["measured_read"] - slab) / spectrum
    assert pairs == 160 and 0.9 <= extra <= 2.0, (pairs, extra)
    mp = tr["fftconv"]["model_check"]
    slab4 = mp["row_pass_slab_GB_per_launch"]
    pairs4 = round(slab4 * 1e9 / (256 * 4096 * 8))
    extra4 = (mp["measured_read"] - slab4) / (256 * 4096 * 8 / 1e9)
    assert pairs4 == 8 and 0.95 <= extra4 <= 2.4, (pairs4, extra4)
