"""The bench line committed with the round's profiles (profiles/r01_bench_under_rocprof.json: what `python3 bench.py`
printed under rocprofv3 on an MI355X) carries every field of the measurement contract, and its numbers hang together."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_follows_the_contract():
    d = json.load(open(os.path.join(ROOT, "profiles", "r01_bench_under_rocprof.json")))
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "Mpixels/s" and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["metric"].split(" at ")[0] in base["metric"] or "4K PBR ShaderBall" in d["metric"]
    assert d["n_gpus"] == 1 and d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"]
    assert "model" not in d["config"]
    # value = pixels per step / time per step
    px = d["config"]["width"] * d["config"]["height"]
    assert abs(d["value"] - px / (d["ms_per_step"] * 1e-3) / 1e6) / d["value"] < 1e-3
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_kernel_ms"] * 1e-3) / 1e9) / r["achieved"] < 1e-3
    assert r["algorithmic_bytes_per_launch"] == r["n_shaded"] * 36 + 6432 + 144      # 16 B + 5 texels per shaded pixel
    assert r["traffic"] is None or r["traffic"] > r["algorithmic_bytes_per_launch"] * 0.5
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["unit"] == d["unit"] and c["cores"] >= 1
    assert d["parity_vs_oracle"]["bit_exact"] is True
    # the per-kernel table of the same run is committed beside it and agrees on the dominant kernel
    rows = [l.split(",") for l in open(os.path.join(ROOT, "profiles", "r01_bench_kernel_stats.csv")).read().splitlines()[1:]]
    assert rows[0][0] == r["kernel"] == "k_shade"
    phases = open(os.path.join(ROOT, "profiles", "r01_bench_kernel_phases.txt")).read()
    timed = float(phases.split("k_shade")[1].split("mean=")[1].split("us")[0])
    assert 0.85 < timed * 1e-3 / r["avg_kernel_ms"] < 1.05      # the events also see the launch gap in front of the kernel
