"""CPU: the bench.py output contract, checked on the JSON lines committed under profiles/ (what the driver parses)."""
import json
import os

import pytest

from conftest import ROOT

REQUIRED = {"metric": str, "value": (int, float), "unit": str, "n_gpus": int, "steps": int, "warmup": int, "ms_per_step": (int, float),
            "higher_is_better": bool, "scaling": str, "dtype": str, "data": str, "config": dict, "roofline": dict}


def _latest_round():
    rounds = sorted(d for d in os.listdir(os.path.join(ROOT, "profiles")) if os.path.isdir(os.path.join(ROOT, "profiles", d)))
    return os.path.join(ROOT, "profiles", rounds[-1])


def _headline():
    with open(os.path.join(_latest_round(), "bench.json")) as f:
        lines = [ln for ln in f.read().splitlines() if ln.strip()]
    assert len(lines) == 1, "bench.py prints ONE JSON line"
    return json.loads(lines[0])


@pytest.mark.parametrize("dtype", ["f32", "f16", "f8"])
def test_committed_bench_lines_follow_the_contract(dtype):
    """The headline line, and the secondary configurations it carries since round 2 (same keys, run-level constants inherited)."""
    d = _headline()
    if dtype != "f32":
        sec = [x for x in d.get("secondary", []) if x.get("dtype") == dtype]
        assert len(sec) == 1, "one secondary entry per precision"
        d = dict({k: d[k] for k in ("n_gpus", "higher_is_better", "scaling", "vs_baseline", "data", "unit")}, **sec[0])
    for k, t in REQUIRED.items():
        assert k in d and isinstance(d[k], t), k
    assert "vs_baseline" in d and d["vs_baseline"] is None          # BASELINE.md publishes no number for this metric
    assert d["unit"] == "images/sec" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["data"] == "synthetic"
    assert d["dtype"] == dtype and d["n_gpus"] == 1 and d["value"] > 0
    assert abs(d["value"] - d["config"]["global_batch"] * 1e3 / d["ms_per_step"]) / d["value"] < 1e-3      # value = images / time
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and r["unit"] == ("GB/s" if r["bound"] == "hbm" else "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0 < r["frac"] < 1
    assert r["peak"] == (8000.0 if r["bound"] == "hbm" else 157.3)
    assert r["traffic"] is None or r["traffic"] > 0
    if dtype == "f32":                                                 # cpu_baseline: rank 0 at N=1 on the headline run
        c = d["cpu_baseline"]
        assert c["kind"] in ("port", "reference") and c["unit"] == "images/sec" and c["value"] > 0 and c["cores"] >= 1 and c["sample"]
