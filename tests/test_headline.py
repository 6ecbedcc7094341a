"""bench.py's LAST stdout line must be one compact JSON record the driver can keep whole (the r05 line was
20 KB, the driver's stdout tail is about 8 KB, and the round's headline went unmeasured). Protocol mirrored: one result line per
run, cli/Benchmark.cpp:105-111 of the reference."""
import io
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "benchmarks"))
import headline  # noqa: E402

CANNED = os.path.join(ROOT, "profiles", "r05", "bench_default_n1_with_configs.json")   # a full r05 record (22 KB)


def _canned():
    return json.load(open(CANNED))


def test_the_final_line_is_under_4k_and_has_every_contract_field():
    full = _canned()
    assert len(json.dumps(full)) > 16000            # the record that broke the driver's parser
    line = headline.headline_line(full)
    assert len(line.encode()) < 4096
    rec = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "parity", "configs"):
        assert k in rec, k
    assert rec["vs_baseline"] is None and rec["dtype"] == "f32"
    assert rec["config"]["workload"].startswith("BASELINE configs[1] (C2)")
    assert rec["config"]["blocks_per_step"] == 1024 and rec["config"]["ranks_seen"] == 1
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "dominant_kernel"):
        assert k in rec["roofline"], k
    assert abs(rec["roofline"]["frac"] - full["roofline"]["frac"]) < 1e-4
    assert abs(rec["roofline"]["dominant_kernel"]["us_per_launch"] - full["roofline"]["dominant_kernel"]["us_per_launch"]) < 0.1
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in rec["cpu_baseline"], k
    assert rec["parity"]["ok"] is True and rec["parity"]["max_abs_err"] <= 1e-6
    assert abs(rec["value"] / full["value"] - 1) < 1e-5 and abs(rec["ms_per_step"] / full["ms_per_step"] - 1) < 1e-5
    # both modes side by side: the offline launch-set figure and the synchronous call
    assert rec["config"]["sync_process_us_per_block_p50"] > rec["config"]["offline_us_per_block"]


def test_every_configuration_is_reduced_to_a_handful_of_scalars():
    rec = json.loads(headline.headline_line(_canned()))
    for name in ("c1", "c3", "c4", "c5", "c5_churn", "taps"):
        c = rec["configs"][name]
        for k in ("value", "ms_per_step", "steps", "roofline_frac", "cpu_value", "parity_ok"):
            assert k in c, (name, k)
        assert all(not isinstance(v, (dict, list)) for v in c.values()), name
    assert rec["configs"]["all_parity_ok"] is True
    assert rec["configs"]["c5"]["commit_to_first_block_ms_p99"] > 0
    assert rec["configs"]["c1"]["us_per_call_p50"] > 0 and rec["configs"]["c1"]["cpu_us_per_call_p50"] > 0


def test_emit_prints_full_records_first_and_the_compact_line_last(tmp_path):
    full = _canned()
    buf = io.StringIO()
    rp = tmp_path / "records.jsonl"
    line = headline.emit(full, stream=buf, records_path=str(rp))
    lines = buf.getvalue().splitlines()
    assert lines[-1] == line and len(lines[-1].encode()) < 4096
    names = [json.loads(ln).get("record") for ln in lines[:-1]]
    assert names == ["c1", "c3", "c4", "c5", "c5_churn", "taps", "headline_full"]
    assert json.loads(lines[-2])["roofline"]["note"] == full["roofline"]["note"]      # nothing lost: the full record is a line above
    assert "record" not in json.loads(lines[-1]) and "configs" not in json.loads(lines[-2])
    assert len(open(rp).read().splitlines()) == 7
    # what an 8 KB tail of stdout keeps still ends with the whole compact line
    tail = buf.getvalue()[-8192:]
    assert json.loads(tail.splitlines()[-1])["metric"] == full["metric"]


def test_a_failed_configuration_and_the_c4_and_multi_gpu_lines_stay_parseable():
    full = _canned()
    full["configs"]["c3"] = {"error": "no result within 240 s", "wall_s": 240.0}
    rec = json.loads(headline.headline_line(full))
    assert rec["configs"]["c3"] == {"error": "no result within 240 s"}
    # `--workload c4` prints its own record through the same path
    c4 = _canned()["configs"]["c4"]
    line = headline.headline_line(c4)
    rec = json.loads(line)
    assert len(line.encode()) < 4096 and rec["config"]["workload"].startswith("BASELINE configs[3] (C4)")
    assert rec["roofline"]["launch_us_per_step"] and rec["cpu_baseline"]["cores"] == 128 and rec["parity"]["ok"] is True
    # an N > 1 line: no cpu_baseline, no configs — still the same compact shape
    n8 = {k: v for k, v in _canned().items() if k not in ("configs", "cpu_baseline", "parity", "cpu_baseline_all_cores")}
    n8["n_gpus"] = 8
    n8["config"]["ranks_seen"] = 8
    rec = json.loads(headline.headline_line(n8))
    assert rec["n_gpus"] == 8 and rec["config"]["ranks_seen"] == 8 and "configs" not in rec


def test_a_pathologically_long_record_still_fits():
    full = _canned()
    full["config"]["workload"] = "x" * 5000
    full["cpu_baseline"]["sample"] = "y" * 5000
    for i in range(12):
        full["configs"][f"extra{i}"] = dict(full["configs"]["c5"])
    assert len(headline.headline_line(full).encode()) < 4096
