"""bench.py's final stdout line: the driver's record keeps the last 8081 characters of stdout, so the headline must be ONE line of at
most bench.LINE_CAP characters that still carries every contract key (round 4's 20 KB line left BENCH_r04.parsed null)."""
import glob
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

RECORDED = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0[3456]_bench_line_driver_command*.json")))
CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
            "roofline", "cpu_baseline")
ROOFLINE = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "alg_bytes_per_launch", "traffic_source", "step", "kernels")
BASELINE = ("value", "unit", "cores", "host_cores", "kind", "impl", "sample", "compress_s", "decompress_s", "median_s")


def _load(path):
    with open(path) as f:
        text = f.read().strip()
    try:
        return json.loads(text)  # one document (possibly indented)
    except json.JSONDecodeError:
        return json.loads(text.splitlines()[-1])  # a capture of stdout: the line is the last one


@pytest.mark.parametrize("path", RECORDED, ids=[os.path.basename(p) for p in RECORDED])
def test_recorded_result_serialises_under_the_cap(path):
    full = _load(path)
    if "kernels_other" not in full:  # already a compact line (a round-5 capture): it must simply be one
        assert len(json.dumps(full, separators=(",", ":"))) <= bench.LINE_CAP
        return
    full["host_path"] = "native"
    text = bench.headline_line(full)
    assert "\n" not in text and len(text) <= bench.LINE_CAP <= 6000
    line = json.loads(text)
    for k in CONTRACT + ("host_path",):
        assert k in line, k
    for k in ROOFLINE:
        assert k in line["roofline"], k
    for k in BASELINE:
        assert k in line["cpu_baseline"], k
    for k in ("metric", "value", "unit", "ms_per_step", "dtype"):
        assert line[k] == full[k]
    assert line["config"]["workload"] == full["config"]["workload"]
    assert line["roofline"]["frac"] == full["roofline"]["frac"] and line["roofline"]["traffic"] == full["roofline"]["traffic"]
    assert line["cpu_baseline"]["kind"] == full["cpu_baseline"]["kind"]
    rows = line["roofline"]["kernels"]
    assert 2 <= len(rows) <= 18
    assert all({"kernel", "config", "us", "frac"} <= set(r) for r in rows)
    assert all(k.startswith("reference_") for k in line["cpu_baseline"]["median_s"])


def test_round_four_result_keeps_every_named_row():
    path = os.path.join(ROOT, "profiles", "r04_bench_line_driver_command.json")
    rows = json.loads(bench.headline_line(_load(path)))["roofline"]["kernels"]
    text = " | ".join(r["config"] for r in rows)
    for needle in ("8192x8192 bf16", "cfg3", "cfg4", "cfg5", "4096^2", "cfg1", "ModelCompressor", "asymmetric"):
        assert needle in text, needle


def test_multi_gpu_line_is_under_the_same_cap():
    full = _load(os.path.join(ROOT, "profiles", "r04_bench_line_driver_command.json"))
    full["n_gpus"] = 8
    full["config"]["ranks_seen"] = 8
    full["config"]["per_rank_GBps"] = [5956.7] * 8
    full["row_sharded"] = {"rows_this_rank": [0, 1024], "ranks": 8, "workload": "x" * 300,
                           "w4a16": {"us_per_tensor": 20.1, "GBps_all_ranks": 16000.0, "frac_of_hbm_peak_per_gpu": 0.25, "shard_equals_slice_of_single_rank_result": True},
                           "sparse_bitmask": {"us_per_tensor": 30.1, "GBps_all_ranks": 14000.0, "frac_of_hbm_peak_per_gpu": 0.21, "row_offsets": "y" * 80,
                                              "shard_equals_slice_of_single_rank_result": True}}
    full["row_sharded"]["w4a16"].update(sets=128, cache="z" * 120, us_per_tensor_blocks=[20.1] * 5)
    full["row_sharded"]["sparse_bitmask"].update(sets=64, cache="z" * 120, us_per_tensor_blocks=[30.1] * 5)
    full["tinyllama_checkpoint"].update(modules_per_rank=[19, 19, 19, 19, 19, 19, 20, 20], every_module_on_exactly_one_rank=True, rotating_copies=8, cache="c" * 100)
    full["w4a16_4096"] = {"workload": "w" * 200, "us_per_step": 18.4, "us_per_step_blocks": [18.4] * 5, "GBps_all_ranks": 36700.0, "frac_of_hbm_peak_per_gpu": 0.573,
                          "ranks": 8, "round_trip_equals_fake_quantize": True}
    text = bench.headline_line(full)
    line = json.loads(text)
    assert len(text) <= bench.LINE_CAP
    assert line["w4a16_4096"]["ranks"] == 8 and line["w4a16_4096"]["GBps_all_ranks"] == 36700.0 and "workload" not in line["w4a16_4096"]
    assert line["tinyllama_checkpoint"]["every_module_on_exactly_one_rank"] is True and sum(line["tinyllama_checkpoint"]["modules_per_rank"]) == 154
    assert line["row_sharded"]["w4a16"]["sets"] == 128
    assert line["config"]["per_rank_GBps"] == [5956.7] * 8
    assert line["row_sharded"]["w4a16"]["shard_equals_slice_of_single_rank_result"] is True
    assert "tinyllama_checkpoint" in line


def test_an_oversized_result_sheds_rows_not_contract_keys():
    full = _load(os.path.join(ROOT, "profiles", "r04_bench_line_driver_command.json"))
    text = bench.headline_line(full, cap=3200)
    line = json.loads(text)
    assert len(text) <= 3200 and len(line["roofline"]["kernels"]) >= 2
    for k in CONTRACT:
        assert k in line
    with pytest.raises(RuntimeError):
        bench.headline_line(full, cap=500)
