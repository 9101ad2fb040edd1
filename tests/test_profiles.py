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
    assert t is not None and t[0] > 1e8 and os.path.exists(os.path.join(ROOT, t[1]))
    rec = json.load(open(os.path.join(ROOT, t[1])))
    assert rec["k_extend"]["traffic_bytes_per_launch"] == int((2 * rec["k_extend"]["fetch_size_kb"] + rec["k_extend"]["write_size_kb"]) * 1024)
    v = b.recorded_valu("k_extend")
    assert v is not None and 0.3 < v["frac"] <= 1.0 and os.path.exists(os.path.join(ROOT, v["source"]))
    s = b.recorded_valu("k_shade")
    assert s is not None and 0.3 < s["frac"] <= 1.0


def test_bench_lines_in_profiles_carry_the_contract_fields():
    names = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_bench.json"))
    assert names
    line = json.load(open(os.path.join(ROOT, "profiles", names[-1])))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    r = line["roofline"]
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["unit"] == "GB/s"
    c = line["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    assert "workload" in line["config"] and "model" not in line["config"]
