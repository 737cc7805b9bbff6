"""The machine-read bench line (bench.final_line): size bound, strict JSON round trip, contract keys -- built from canned records
(the round-4 record the driver could NOT parse, and a synthetic N > 1 record).  No GPU, no oracle."""
import json
import math
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import bench  # noqa: E402

CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                 "dtype", "data", "config", "roofline", "cpu_baseline")


def _strict(text):
    def no_const(name):
        raise ValueError(f"non-finite constant {name} in the bench line")
    return json.loads(text, parse_constant=no_const)


def _check(text, n_gpus):
    assert "\n" not in text
    assert len(text.encode()) < bench.FINAL_LINE_MAX_BYTES
    line = _strict(text)
    assert json.dumps(line, allow_nan=False)                 # round trip
    for k in CONTRACT_KEYS:
        assert k in line, k
    assert line["n_gpus"] == n_gpus and isinstance(line["config"].get("workload"), str)
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in line["roofline"], k
    assert line["roofline"]["bound"] in ("hbm", "mfma")
    assert abs(line["roofline"]["frac"] - line["roofline"]["achieved"] / line["roofline"]["peak"]) < 1e-4
    return line


def test_round4_record_fits_and_round_trips():
    detail = json.load(open(os.path.join(ROOT, "profiles", "r04_bench_n1.json")))
    assert len(json.dumps(detail)) > 20000                   # the record that broke the driver's parser
    line = _check(bench.final_line(detail), 1)
    assert line["cpu_baseline"]["kind"] in ("port", "reference") and line["cpu_baseline"]["cores"] >= 1
    assert set(line["configs"]) == {"cora", "gat", "tgcn", "dynamic"}
    for name, summ in line["configs"].items():
        assert len(json.dumps(summ)) <= 400, name
        assert summ["value"] > 0 and "frac" in summ["roofline"]
    assert line["roofline"]["north_star"]["frac"] > 0


def test_non_finite_numbers_become_null():
    detail = json.load(open(os.path.join(ROOT, "profiles", "r04_bench_n1.json")))
    detail["roofline"]["traffic"] = float("nan")
    detail["ms_per_step"] = float("inf")
    line = _strict(bench.final_line(detail))
    assert line["roofline"]["traffic"] is None and line["ms_per_step"] is None


def test_multi_rank_record_headlines_tgcn():
    tg = {"metric": "epochs/s", "value": 55.0, "seconds_per_epoch": 1 / 55.0, "n_gpus": 8, "us_per_snapshot": 140.0,
          "edges_feat_per_s": 1.7e12, "roofline": {"bound": "hbm", "frac": 0.4, "kernel": "x" * 500},
          "windows_per_epoch": 40, "optimizer_steps_per_epoch": 5}
    detail = {"metric": "epochs/s (static-temporal TGCN, BASELINE configs[3], BPTT windows sharded over the ranks)", "value": 55.0,
              "unit": "epochs/s", "n_gpus": 8, "steps": 20, "warmup": 3, "ms_per_step": 18.2, "higher_is_better": True, "scaling": "strong",
              "vs_baseline": None, "dtype": "f32", "data": "synthetic",
              "config": {"workload": "static-temporal TGCN", "parallelism": "dp8", "windows_per_epoch": 40, "note": "y" * 5000},
              "roofline": {"bound": "mfma", "achieved": 50.0, "peak": 157.3, "unit": "TFLOP/s", "frac": 50.0 / 157.3, "traffic": None,
                           "kernel": "stg::tgcn_step_bwd_kernel", "table": ["z" * 100] * 100},
              "cpu_baseline": None,
              "allreduce": {"bytes": 132868, "calls_per_epoch": 5.0, "share_of_epoch": 0.01, "in_graph": False, "collective": "w" * 300},
              "process_group": {"world_size": 8, "backend": "nccl", "distinct_devices": 8,
                                "ranks": [{"rank": r, "name": "AMD Instinct MI355X"} for r in range(8)]},
              "tgcn": tg, "dynamic": {"metric": "epochs/s", "value": 300.0, "T": 160, "roofline": {"frac": 0.2}},
              "gcn_replicas": {"metric": "edges*feat/s", "value": 1e13, "ms_per_step": 4.8, "n_gpus": 8, "roofline": {"bound": "hbm", "frac": 0.95}}}
    line = _check(bench.final_line(detail), 8)
    assert line["scaling"] == "strong" and line["unit"] == "epochs/s" and line["cpu_baseline"] is None
    assert line["process_group"]["world_size"] == 8 and len(line["process_group"]["devices"]) == 8
    assert line["allreduce"]["share_of_epoch"] == 0.01
    assert "gcn_replicas" in line["configs"] and line["configs"]["tgcn"]["us_per_snapshot"] == 140.0


def test_prose_is_clipped_never_fatal():
    detail = json.load(open(os.path.join(ROOT, "profiles", "r04_bench_n1.json")))
    detail["config"]["workload"] = "w" * 7000
    for k in ("cora", "gat", "tgcn", "dynamic"):
        detail[k]["metric"] = "m" * 900
        detail[k]["roofline"]["kernel"] = "k" * 900
    detail["cpu_baseline"]["sample"] = "s" * 4000
    line = _check(bench.final_line(detail), 1)
    assert len(line["config"]["workload"]) <= 240 and line["config"]["workload"].endswith("...")
    assert all(len(v["metric"]) <= 120 for v in line["configs"].values())


def test_sig_rounds_and_keeps_integers():
    assert bench._sig(1.23456789e12) == 1.23457e12 and bench._sig(8840000004) == 8840000004 and bench._sig(True) is True
    assert bench._sig(float("nan")) is None and math.isclose(bench._sig(0.000123456789), 0.000123457)


def test_gpus_without_launcher_starts_child_ranks(tmp_path):
    """`python3 bench.py --gpus 2` with WORLD_SIZE unset must start its own ranks (the driver's BENCH form), not exit: here, with no
    GPU, each child rank stops at the device check -- AFTER the launcher relayed two ranks' worth of that message."""
    import torch
    if torch.cuda.device_count() > 0:
        pytest.skip("a GPU is visible: the launcher would run the real benchmark")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0
    assert "starting 2 ranks" in r.stderr
    assert "WORLD_SIZE=1" not in r.stderr                     # the round-4 failure mode
    assert "HIP device(s) visible" in r.stderr or "no HIP device" in r.stderr
