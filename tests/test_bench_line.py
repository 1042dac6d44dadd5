"""The ONE line `bench.py` prints (round-5 review, item 1: the driver's parser lost the 26 KB line of round 5): compact_line()
keeps it below 4 KB whatever the full record holds, with the contract's keys, `roofline` and `cpu_baseline` complete -- for the
N = 1 record, for an N > 1 record and for the provisional line of a stuck N > 1 run.  The GPU tier runs the real program
(tests/test_distributed_gloo.py: the 2- and 4-rank rehearsals assert the same on their lines; test_bench_default_line_on_the_gpu
below runs the N = 1 command)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline")
ROOFLINE = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "hbm_frac", "bytes_per_unit")


def bench_module():
    import importlib.util
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)      # (not __main__: neither the launcher nor the environment set-up runs)
    return mod


def full_record():
    """A full record of the shape main() builds, as large as round 5's (26 KB): that round's committed line + this round's keys."""
    rec = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_default.json")))
    rec.update({"parity_max_abs_dtheta": 8.39e-5, "parity_tolerance": 1e-4, "value_smooth": 8.1e8, "detail": "bench_detail.json",
                "detail_scalars": {f"scalar_{i}": 1.0 / 3.0 * i for i in range(20)}})
    rec["config"]["env"] = {"GPU_MAX_HW_QUEUES": "22", "HSA_ENABLE_IPC_MODE_LEGACY": "0"}
    return rec


def check_line(s, n_gpus):
    assert len(s.encode()) < 4096 and "\n" not in s and s.startswith('{"metric"')
    b = json.loads(s)
    assert all(k in b for k in CONTRACT) and b["n_gpus"] == n_gpus and b["higher_is_better"] is True and b["dtype"] == "f64"
    assert all(k in b["roofline"] for k in ROOFLINE) and {"workload", "variant", "streams"} <= set(b["config"])
    assert "NaN" not in s and "Infinity" not in s
    return b


def test_compact_line_of_a_one_gpu_record():
    bench = bench_module()
    rec = full_record()
    assert len(json.dumps(rec)) > 20000
    b = check_line(bench.compact_line(rec), 1)
    assert b["roofline"]["frac"] == rec["roofline"]["frac"] and b["roofline"]["hbm_frac"] == pytest.approx(rec["roofline"]["hbm"]["frac"], rel=1e-5)
    assert b["roofline"]["traffic"] == pytest.approx(rec["roofline"]["traffic"], rel=1e-5) and b["roofline"]["lane_utilisation"] > 0.4
    assert b["cpu_baseline"]["kind"] == "port" and b["cpu_baseline"]["cores"] == 16 and b["cpu_baseline"]["python_scipy_pool_value"] > 100
    assert b["parity_max_abs_dtheta"] == 8.39e-5 and b["value_single_job"] > 1e8 and b["value_smooth"] == 8.1e8
    assert b["config"]["env"]["GPU_MAX_HW_QUEUES"] == "22" and b["detail"] == "bench_detail.json"
    assert b["value"] == pytest.approx(rec["value"], rel=1e-5) and b["ms_per_step"] == pytest.approx(rec["ms_per_step"], rel=1e-5)
    # whatever a later round adds to the record, the line stays below the limit: the optional parts are dropped first
    rec["detail_scalars"] = {f"a_rather_long_scalar_name_{i}": 1.0 / 7.0 * i for i in range(200)}
    b2 = check_line(bench.compact_line(rec), 1)
    assert "detail_scalars" not in b2 and b2["roofline"] == b["roofline"]


def test_compact_line_of_an_n_gpu_record_and_of_a_provisional_one():
    bench = bench_module()
    rec = full_record()
    for k in ("cpu_baseline", "value_single_job", "value_smooth", "single_job", "variants", "configs", "parity", "strong_projection"):
        rec.pop(k, None)
    rec.update({"n_gpus": 8, "value": 2.9e9, "ms_per_step": 2.07})
    rec["config"].update({"workload": "config 3 literally: the FIXED problem of synthetic 1M frames x 6 legs IN TOTAL (15625 sequences of 64 "
                                      "frames), sequences split over the 8 ranks, joint angles gathered on rank 0", "streams": 20,
                          "stage_pipeline": 1, "gather": "grouped RCCL point-to-point"})
    rec["multi_gpu"] = {"backend": "nccl (RCCL)", "rccl_ranks": 8, "devices_distinct": 8,
                        "ranks_seen": [{"rank": r, "host": "node-with-a-long-name", "device": r, "gpu": "GPU-%032x" % r} for r in range(8)],
                        "rank_ms_per_step": {"min": 2.01, "max": 2.07, "by_rank": [2.05] * 8},
                        "n1_reference": {"value": 5.1e8, "ms_per_step": 11.8, "steps": 20}, "efficiency_vs_n1": 0.71, "speedup_vs_n1": 5.7,
                        "gather_compare": {"peer": {}, "rccl": {}}, "weak": {"value": 1.0}}
    b = check_line(bench.compact_line(rec), 8)
    m = b["multi_gpu"]
    assert m["ranks_seen_n"] == 8 and m["rccl_ranks"] == 8 and m["devices_distinct"] == 8 and m["efficiency_vs_n1"] == 0.71
    assert m["n1_value"] == 5.1e8 and m["gather"] == "grouped RCCL point-to-point" and m["legs"] == ["gather_compare", "weak"]
    assert m["rank_ms_per_step_min_max"] == [2.01, 2.07] and "timed_out" not in m
    rec["multi_gpu"]["timed_out"] = "a leg behind the headline did not finish within 300 s"
    assert json.loads(bench.compact_line(rec))["multi_gpu"]["timed_out"].startswith("a leg behind")
    # the provisional line of a run that got stuck behind its first measurement: no kernel events, no PMC figures
    prov = {k: rec[k] for k in bench.HEAD_KEYS}
    prov["config"] = {"workload": rec["config"]["workload"], "variant": "iid", "streams": 3, "stage_pipeline": 0, "provisional": True,
                      "gather": "grouped RCCL point-to-point", "env": {}}
    prov["roofline"] = {"bound": "hbm", "kernel": None, "achieved": 200.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.025, "traffic": None,
                        "avg_launch_ms": None, "bytes_per_unit": 392}
    s = bench.compact_line(prov)
    b = check_line(s, 8)
    assert '"provisional":true' in s and b["roofline"]["hbm_frac"] == 0.025 and "multi_gpu" not in b


def test_bench_files_stay_small_and_split():
    """Round-5 review, item 9: the headline and its roofline are what bench.py holds (<= 500 lines); the other legs live in
    bench_extras.py and are imported only for --detail / --legs all / --one-recording."""
    text = open(os.path.join(ROOT, "bench.py")).read()
    assert len(text.splitlines()) <= 500
    top_level = [l for l in text.splitlines() if l.startswith(("import ", "from "))]
    assert not any("bench_extras" in l for l in top_level) and text.count("import bench_extras") == 3
    assert os.path.exists(os.path.join(ROOT, "bench_extras.py")) and os.path.exists(os.path.join(ROOT, "bench_support.py"))


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_bench_default_line_on_the_gpu(tmp_path):
    """The driver's command on the real program, at a size that takes seconds: ONE stdout line below 4 KB with every key of the
    contract, roofline and cpu_baseline filled, the full record in the detail file."""
    detail = tmp_path / "detail.json"
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2", "--frames", "64000",
                        "--cpu-sample-seqs", "64", "--detail-path", str(detail)], env=env, capture_output=True, text=True, timeout=800)
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert r.returncode == 0 and len(lines) == 1, r.stdout[-2000:] + r.stderr[-4000:]
    b = check_line(lines[0], 1)
    assert b["steps"] == 6 and b["warmup"] == 2 and abs(b["value"] - 64000 * 6 / (b["ms_per_step"] * 1e-3)) < 1e-5 * b["value"]
    assert b["roofline"]["avg_launch_ms"] > 0 and b["roofline"]["kernel"] == "seqik_fused_kernel<true>" and b["roofline"]["hbm_frac"] > 0
    assert b["cpu_baseline"]["value"] > 0 and b["cpu_baseline"]["kind"] == "port" and b["cpu_baseline"]["cores"] >= 1
    assert 0 < b["parity_max_abs_dtheta"] < 1e-4 and b["value_single_job"] > 0 and b["value_smooth"] > 0
    assert b["config"]["env"]["GPU_MAX_HW_QUEUES"] == "22" and "extras_error" not in b
    full = json.loads(detail.read_text())
    assert full["value"] == pytest.approx(b["value"], rel=1e-5) and "depth_calibration" in full["config"] and "single_job" in full
