"""bench.py's contract pieces that do not need a GPU: the synthetic batch is the one SURVEY.md §8(d) specifies, the CLI
defaults are the driver's (N = 1, minutes), the CPU baseline leg runs the oracle and reports the required fields, and the
line checked into profiles/ carries every key the driver and the judge read."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_synthetic_batch_follows_the_measurement_spec():
    sl = bench.synthetic_batch(32, rank=0, device=None)
    assert tuple(sl["input_ids"].shape) == (32, 128) and sl["input_ids"].dtype == torch.int64
    assert bool((sl["input_ids"][:, 0] == 101).all()) and int(sl["input_ids"].max()) < 30522
    assert bool((sl["input_mask"] == 1).all()) and bool((sl["segment_ids"] == 0).all())
    f = sl["image_feature_0"]
    assert tuple(f.shape) == (32, 100, 2048) and f.dtype == torch.float32 and float(f.min()) >= 0.0 and float(f.max()) < 1.0
    assert bool((sl["image_info_0"]["max_features"] == 100).all())
    t = sl["targets"]
    assert tuple(t.shape) == (32, 3129)
    for row in t:
        vals = sorted(row[row > 0].tolist())
        assert len(vals) == 3 and all(abs(a - b) < 1e-6 for a, b in zip(vals, (0.3, 0.6, 1.0)))
    other = bench.synthetic_batch(32, rank=1, device=None)
    assert not torch.equal(other["input_ids"], sl["input_ids"])          # seed 1234 + rank: ranks see different samples


def test_cli_defaults_are_the_drivers(monkeypatch):
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = bench.parse()
    assert (a.gpus, a.batch) == (1, 32) and a.steps > 0 and a.warmup > 0 and not a.no_optimizer and not a.no_graph
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "7", "--warmup", "3"])
    a = bench.parse()
    assert (a.gpus, a.steps, a.warmup) == (8, 7, 3)


def test_cpu_baseline_leg_reports_the_oracle_on_the_host_cores():
    r = bench.cpu_baseline(batch=1, steps=1, budget_s=1.0)
    assert r["kind"] == "port" and r["unit"] == "samples/s" and r["value"] > 0 and 1 <= r["cores"] <= 64
    assert "oracle" in r["sample"] and "B=1" in r["sample"]


def test_committed_bench_line_has_every_contract_key():
    line = json.load(open(os.path.join(ROOT, "profiles", "r01_bench_line.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["metric"].startswith("samples/sec/node VisualBERT VQA2") and line["unit"] == "samples/s"
    assert line["dtype"] == "bf16" and line["data"] == "synthetic" and line["vs_baseline"] is None and line["scaling"] == "weak"
    assert "workload" in line["config"] and "model" not in line["config"]
    roof = line["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in roof, k
    assert roof["bound"] in ("hbm", "mfma") and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
    cb = line["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert abs(line["value"] - line["config"]["global_batch"] / line["ms_per_step"] * 1e3) <= 0.01 * line["value"]
