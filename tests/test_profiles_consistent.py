"""The measurement evidence under profiles/ is self-consistent: every roofline fraction of the committed bench lines can be recomputed
from the fields next to it, none exceeds 1, the event-timed kernel durations of bench.py agree with the rocprofv3 --kernel-trace
--stats summary of the same command, and kernel time sums to no more than the frame (each kernel is timed alone).  No GPU needed:
this reads what the GPU runs of the round left behind ("profiles from which every frac in the bench line can be
recomputed and none exceeds 1")."""
import csv
import json
import os

import pytest

from conftest import ROOT

PROF = os.path.join(ROOT, "profiles")
BENCHES = {"cfg3": "r03A_bench_cfg3.json", "5m": "r03A_bench_5m.json", "10m_vol": "r03A_bench_10m_vol.json", "div5m": "r03A_bench_div5m.json",
           "div10m_vol": "r03A_bench_div10m_vol.json", "cfg3_vector_issue": "r03C_bench_cfg3.json", "r03q_cfg3": "r03q_bench_cfg3.json", "r03q_div5m": "r03q_bench_div5m.json", "r03j_cfg3": "r03j_bench_cfg3.json", "r02_cfg3": "r02u_bench_cfg3.json", "r02_5m": "r02u_bench_5m.json"}


def bench(tag):
    return json.load(open(os.path.join(PROF, BENCHES[tag])))


@pytest.mark.parametrize("tag", sorted(BENCHES))
def test_bench_line_carries_the_contract(tag):
    b = bench(tag)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in b, k
    assert b["metric"] == "Mrays/s" and b["dtype"] == "f32" and b["data"] == "synthetic" and b["vs_baseline"] is None and "workload" in b["config"]
    if tag in ("cfg3", "cfg3_vector_issue", "r03q_cfg3", "r03j_cfg3", "r02_cfg3"):
        cb = b["cpu_baseline"]
        assert cb["kind"] == "reference" and cb["cores"] >= 1 and cb["value"] > 0 and "spp" in cb["sample"]


@pytest.mark.parametrize("tag", sorted(BENCHES))
def test_every_fraction_recomputes_and_stays_below_one(tag):
    b = bench(tag)
    assert b["roofline"]["kernel"] == b["roofline_kernels"][0]["kernel"]  # dominant = the most solo time
    total = 0.0
    for k in b["roofline_kernels"]:
        achieved = k["algorithmic_bytes_per_launch"] * k["launches"] / (k["total_ms"] * 1e-3) / 1e9
        assert achieved == pytest.approx(k["achieved"], rel=1e-9) and k["frac"] == pytest.approx(k["achieved"] / k["peak"], rel=1e-12)
        assert 0 < k["frac"] < 1 and k["avg_launch_ms"] == pytest.approx(k["total_ms"] / k["launches"], rel=1e-12)
        assert (k["bound"], k["peak"]) in (("l2", 34500.0), ("hbm", 8000.0))
        if k.get("hbm_side"):
            hb = k["traffic"] * k["launches"] / (k["total_ms"] * 1e-3) / 1e9
            assert hb == pytest.approx(k["hbm_side"]["achieved"], rel=1e-9) and 0 < k["hbm_side"]["frac"] < 1
        total += k["total_ms"]
    # the working set decides the bound: below the 256 MiB Infinity Cache the L2, above it HBM
    assert (b["roofline"]["working_set_bytes"] < 256 << 20) == (b["roofline"]["bound"] == "l2")
    # kernels are timed alone (events around each launch): their sum fits into the frames they ran in
    other = sum(v for k, v in b["kernel_ms_per_step"].items() if k in ("resolve", "generate", "film"))
    assert total / b["steps"] + other <= b["ms_per_step"] * 1.001
    g = b["roofline"].get("gather")
    if g and "frac" in g:
        assert g["frac"] == pytest.approx(g["record_fetches_per_s"] / g["ceiling_records_per_s"], rel=1e-12) and 0 < g["frac"] < 1
        if g["l2_hit_rate"] is not None:
            h = g["l2_hit_rate"]
            assert g["ceiling_records_per_s"] == pytest.approx(1 / (h / g["ceiling_l2_resident"] + (1 - h) / g["ceiling_at_working_set"]), rel=1e-9)


@pytest.mark.parametrize("tag,stats", [("cfg3", "r03A_kernel_stats_cfg3.csv"), ("5m", "r03A_kernel_stats_5m.csv"), ("div5m", "r03A_kernel_stats_div5m.csv"), ("r03q_cfg3", "r03q_kernel_stats_cfg3.csv"), ("r03q_div5m", "r03q_kernel_stats_div5m.csv"),
                                       ("div10m_vol", "r03A_kernel_stats_div10m_vol.csv"), ("r03j_cfg3", "r03j_kernel_stats_cfg3.csv"), ("r02_cfg3", "r02u_kernel_stats_cfg3.csv"),
                                       ("r02_5m", "r02u_kernel_stats_5m.csv")])
def test_event_timing_agrees_with_the_rocprof_summary(tag, stats):
    """bench.py times each kernel with HIP events on its own stream; rocprofv3 --kernel-trace --stats of the same command gives the
    same average duration per kernel (the profiler's own overhead stays below a few percent)."""
    rows = {r["Name"]: r for r in csv.DictReader(open(os.path.join(PROF, stats)))}
    def avg_ms(*prefixes, also=()):  # also: kernels launched inside the same pair of events (k_shade_order before k_shade<2, .>)
        hit = [r for n, r in rows.items() if n.startswith(prefixes)]
        extra = [r for n, r in rows.items() if also and n.startswith(also)]
        assert hit, prefixes
        return sum(float(r["TotalDurationNs"]) for r in hit + extra) / sum(int(r["Calls"]) for r in hit) / 1e6
    b = bench(tag)
    by = {k["kernel"].split(" ")[0]: k for k in b["roofline_kernels"]}
    closest = by.get("k_trace<0>") or by["k_trace<false>"]
    assert avg_ms("void k_trace<0,", "void k_trace<false") == pytest.approx(closest["avg_launch_ms"], rel=0.05)
    anyhit = by.get("k_trace<2>") or by.get("k_trace<1>") or by.get("k_trace<true>")
    if anyhit:  # (volpath has no shadow rays: transmittance rays are closest-hit queries)
        assert avg_ms("void k_trace<2,", "void k_trace<1,", "void k_trace<true") == pytest.approx(anyhit["avg_launch_ms"], rel=0.05)
    assert avg_ms("void k_shade<", also=("void k_shade_order",)) == pytest.approx(by["k_shade"]["avg_launch_ms"], rel=0.08)


def test_hbm_regime_of_the_headline_line_is_reproducible():
    """roofline.hbm_regime: the closest-hit kernel on the 5 M-triangle scene, timed inside the headline run by HIP
    events; the kept rocprofv3 summary of the same workload (same launches: whole-frame batches of 64 spp) gives the same average, the
    fraction recomputes from the fields next to it and stays below 1, and it is the north star's >= 0.40 of the HBM roofline."""
    h = bench("cfg3")["roofline"]["hbm_regime"]
    assert h["bound"] == "hbm" and h["peak"] == 8000.0 and h["working_set_bytes"] > 256 << 20
    assert h["achieved"] == pytest.approx(h["algorithmic_bytes_per_launch"] / (h["avg_launch_ms"] * 1e-3) / 1e9, rel=1e-9)
    assert h["frac"] == pytest.approx(h["achieved"] / 8000.0, rel=1e-12) and 0.40 <= h["frac"] < 1
    rows = list(csv.DictReader(open(os.path.join(PROF, "r03A_kernel_stats_5m.csv"))))
    k = [r for r in rows if r["Name"].startswith("void k_trace<0,")]
    avg = sum(float(r["TotalDurationNs"]) for r in k) / sum(int(r["Calls"]) for r in k) / 1e6
    assert avg == pytest.approx(h["avg_launch_ms"], rel=0.05)


def test_replayed_values_are_labelled():
    """whatever the bench line takes from a committed PMC pass instead of measuring it says where it came from."""
    b = bench("cfg3")
    for k in b["roofline_kernels"]:
        if k.get("traffic") is not None:
            assert k["traffic_source"]["source"].startswith("profiles/pmc_traffic.json") and "not measured in this run" in k["traffic_source"]["note"]
            assert k["hbm_side"]["source"].startswith("profiles/pmc_traffic.json")
    g = b["roofline"].get("gather")
    if g and g.get("l2_hit_rate") is not None:
        assert g["l2_hit_rate_source"].startswith("profiles/pmc_traffic.json")
    cal = json.load(open(os.path.join(PROF, "fetch_size_calibration.json")))
    assert cal["gather_factor"] == 2.0 and "k_gather_pair" in cal["raw"]["kernels"]


def test_vector_issue_fraction_recomputes():
    """roofline_kernels[].vector_issue: wave-wide vector instructions per second (count replayed from the committed PMC pass, SQ_INSTS_VALU)
    against 1024 SIMD-32 units x 2.4 GHz / 2 cycles per 64-wide instruction (MI355X_MICROARCH.md, "Wave scheduling")."""
    b = bench("cfg3_vector_issue")
    seen = 0
    for k in b["roofline_kernels"]:
        v = k.get("vector_issue")
        if not v: continue
        seen += 1
        assert v["peak"] == pytest.approx(256 * 4 * 2.4e9 / 2) and v["source"].startswith("profiles/pmc_traffic.json")
        assert v["achieved"] == pytest.approx(v["insts_per_launch"] * k["launches"] / (k["total_ms"] * 1e-3), rel=1e-9)
        assert v["frac"] == pytest.approx(v["achieved"] / v["peak"], rel=1e-12) and 0 < v["frac"] < 1 and 1 <= v["lanes_active_of_64"] <= 64
    assert seen >= 2


# ---- round 5: the timed frames overlap any-hit with closest-hit launches; per-kernel figures come from one serialised frame --------------
R05 = {"cfg3": "r05z_bench.json", "div5m": "r05z_bench_div5m.json", "div10m_vol": "r05z_bench_div10m_vol.json"}


@pytest.mark.parametrize("tag", sorted(R05))
def test_round5_lines_recompute(tag):
    """kernel_ms_per_step / roofline_kernels of a round-5 line are ONE serialised frame's HIP-event times (kernel_times says so): they sum
    to no more than that frame, every fraction recomputes and stays below 1, and `l2_memory_side` (what round 4 called hbm_side: the L2s'
    fabric ports, Infinity-Cache hits included) is traffic / time beside the 8 TB/s yardstick."""
    b = json.load(open(os.path.join(PROF, R05[tag])))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in b, k
    kt = b["kernel_times"]
    assert "every kernel alone on the chip" in kt["from"] and kt["sum_of_kernels_ms"] <= kt["serialized_frame_ms"] * 1.001
    assert kt["sum_of_kernels_ms"] == pytest.approx(sum(b["kernel_ms_per_step"].values()), rel=1e-9)
    assert kt["overlapped_frame_ms"] == pytest.approx(b["ms_per_step"], rel=1e-12)
    assert b["value"] == pytest.approx(b["config"]["rays_per_sample"] * b["samples_per_s"] / 1e6, rel=1e-6)
    assert b["roofline"]["kernel"] == b["roofline_kernels"][0]["kernel"]
    for k in b["roofline_kernels"]:
        achieved = k["algorithmic_bytes_per_launch"] * k["launches"] / (k["total_ms"] * 1e-3) / 1e9
        assert achieved == pytest.approx(k["achieved"], rel=1e-9) and k["frac"] == pytest.approx(k["achieved"] / k["peak"], rel=1e-12) and 0 < k["frac"] < 1
        assert "hbm_side" not in k
        m = k.get("l2_memory_side")
        if m:
            assert m["achieved"] == pytest.approx(k["traffic"] * k["launches"] / (k["total_ms"] * 1e-3) / 1e9, rel=1e-9) and m["compared_with"] == 8000.0
            assert "Infinity-Cache hits included" in m["note"] and m["source"].startswith("live")


def test_round5_event_timing_agrees_with_the_rocprof_summary():
    """profiles/r05z_kernel_stats_cfg3.csv: rocprofv3 --kernel-trace --stats of `bench.py --no-overlap` (every kernel alone, as in the
    serialised frame the line's per-kernel times come from) gives the same average duration for the dominant kernel."""
    rows = {r["Name"]: r for r in csv.DictReader(open(os.path.join(PROF, "r05z_kernel_stats_cfg3.csv")))}
    hit = [r for n, r in rows.items() if n.startswith("void k_trace<0,")]
    avg = sum(float(r["TotalDurationNs"]) for r in hit) / sum(int(r["Calls"]) for r in hit) / 1e6
    b = json.load(open(os.path.join(PROF, R05["cfg3"])))
    assert avg == pytest.approx(b["roofline"]["avg_launch_ms"], rel=0.05)
    h = b["roofline"]["hbm_regime"]
    assert h["bound"] == "hbm" and h["frac"] == pytest.approx(h["achieved"] / 8000.0, rel=1e-12) and 0.40 <= h["frac"] < 1
    assert "l2_memory_side" in h and "hbm_side" not in h


@pytest.mark.parametrize("tag", ["r05z", "r06z"])
def test_whole_frame_parity_of_configs_4_and_5(tag):
    """profiles/r0{5,6}z_fullframe_parity_config4_5.json (the closing trees of rounds 5 and 6): the device's whole 1920x1080 frame of both stand-ins at their own 256 / 128 spp against the
    fingerprint of the reference binary's image (tests/golden_large/fullframe_reference_fingerprint_config{41,51}.json)."""
    res = json.load(open(os.path.join(PROF, tag + "_fullframe_parity_config4_5.json")))
    assert [r["config"] for r in res] == [41, 51]
    for r in res:
        fp = json.load(open(os.path.join(ROOT, "tests", "golden_large", f"fullframe_reference_fingerprint_config{r['config']}.json")))
        assert r["same_scene_file"] and r["sha256_equal"] and r["tiles_differing"] == 0 and r["pixels_differing"] == 0 and r["window"] is None
        assert r["compared_pixels"] == 1920 * 1080 and r["spp"] == (256 if r["config"] == 41 else 128) and len(fp["tile_crc32"]) == r["tiles"] == 8160
        for c in ("camera_rays", "closest_rays", "shadow_rays"):
            assert r["device_counters"][c] == fp["reference_counters"][c]

