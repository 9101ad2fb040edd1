"""The recorded-counter figures bench.py puts into its JSON line (roofline.traffic, roofline.valu_busy) come from files
committed under profiles/: they must parse, belong to the kernels named, and be physically plausible."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_recorded_traffic_and_valu():
    b = load_bench()
    t = b.recorded_traffic("k_extend")
    assert t is not None and t["bytes_per_launch"] > 1e8 and os.path.exists(os.path.join(ROOT, t["source"]))
    rec = json.load(open(os.path.join(ROOT, t["source"])))
    assert rec["k_extend"]["traffic_bytes_per_launch"] == int((2 * rec["k_extend"]["fetch_size_kb"] + rec["k_extend"]["write_size_kb"]) * 1024)
    v = b.recorded_valu("k_extend")
    assert v is not None and 0.3 < v["frac"] <= 1.0 and os.path.exists(os.path.join(ROOT, v["source"]))
    s = b.recorded_valu("k_shade")
    assert s is not None and 0.3 < s["frac"] <= 1.0


def test_recorded_counters_belong_to_the_sources_in_the_tree():
    """Every counter recording bench.py quotes is stamped with the sha256 of the kernel / launch / tree sources it was
    taken with (scripts/source_stamp.py).  A recording of other sources is STALE: bench.py says so on its line, and this
    test fails until the passes are collected again (scripts/gpu_profile.sh, scripts/gpu_pmc.sh)."""
    b = load_bench()
    for kernel in ("k_extend", "k_shade"):
        t, v = b.recorded_traffic(kernel), b.recorded_valu(kernel)
        assert t is not None and not t["stale"], "profiles/*_pmc_traffic.json was recorded with other sources: %s" % t["source"]
        assert v is not None and not v["stale"], "profiles/*_pmc_valu.json was recorded with other sources: %s" % v["source"]


def test_bench_lines_in_profiles_carry_the_contract_fields():
    names = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_bench.json"))
    assert names
    line = json.load(open(os.path.join(ROOT, "profiles", names[-1])))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    r = line["roofline"]
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["unit"] == "GB/s"
    if "hbm_physical" in r:      # round 2 on: the physical side beside the nominal (algorithmic) figure
        h = r["hbm_physical"]
        assert h is None or (abs(h["frac"] - h["GB/s"] / r["peak"]) < 1e-3 and h["bytes_per_launch"] == r["traffic"])
        assert r["limiter"] in (None, "hbm", "valu_issue", "latency")
    c = line["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    assert "workload" in line["config"] and "model" not in line["config"]
