"""The committed bench lines (profiles/r0N_*.json, printed by bench.py on an MI355X) carry every field of the driver's
contract, name BASELINE.json's metric and workload, and are internally consistent (value = frames / time, roofline
fraction = achieved / peak)."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BASE = json.load(open(os.path.join(ROOT, "BASELINE.json")))
LINES = ["profiles/r05_bench_bf16.json", "profiles/r05_bench_step_fence_line.json", "profiles/r05_bench_bf16_before_step_fence.json", "profiles/r05_driver_cmd_final_line.json", "profiles/r05_bench_bf16_side_stream_masks.json", "profiles/r05_driver_cmd_side_stream_masks_line.json", "profiles/r05_driver_exact_cmd_line.json", "profiles/r04_bench_bf16.json", "profiles/r04_bench_bf16_evidence_pass.json", "profiles/r04_driver_exact_cmd_line.json",
         "profiles/r03_bench_bf16.json", "profiles/r02_bench_bf16.json", "profiles/r01_bench_bf16.json", "profiles/r01_bench_fp32_with_cpu_baseline.json",
         "profiles/r01_timit_lstm_8wave_bench.json", "profiles/r01_timit_lstm_4wave_bench.json"]


def _line(rel):
    return json.loads(open(os.path.join(ROOT, rel)).read().strip().split("\n")[-1])


@pytest.mark.parametrize("rel", LINES)
def test_contract_fields(rel):
    d = _line(rel)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["unit"] == "frames/s" and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None                      # BASELINE.md holds no published number for this metric
    assert d["data"] == "synthetic" and d["dtype"] in ("bf16", "fp32")
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    # whole-job throughput = frames of all timed steps / their wall time
    frames_per_step = d["config"]["global_batch"] * d["config"]["seq_len"]
    assert abs(d["value"] - frames_per_step / (d["ms_per_step"] * 1e-3)) < 0.01 * d["value"]


def test_headline_line_names_the_baseline_workload():
    d = _line(LINES[0])
    assert "frames" in BASE["metric"].lower() or "frames" in json.dumps(BASE).lower()
    assert d["config"]["workload"].startswith("timit_ligru") and d["config"]["seq_len"] == 500 and d["config"]["global_batch"] == 128
    assert d["dtype"] == "bf16" and d["n_gpus"] == 1
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] == "port" and c["unit"] == "frames/s" and c["cores"] >= 1
    assert c["B"] == 128                                 # the sample is taken at the metric's batch
    assert d["roofline"]["traffic"] is not None          # PMC-measured HBM bytes of the dominant launch


def test_round2_line_carries_the_parity_mode_and_every_baseline_configuration():
    """The default single-GPU run also reports the same workload in the exact-fp32 (1e-4-grade) mode and the other four
    BASELINE.json configurations."""
    d = _line("profiles/r02_bench_bf16.json")
    pm = d["parity_mode"]
    assert pm["dtype"] == "fp32" and pm["unit"] == "frames/s" and pm["roofline"]["peak"] == 157.3
    assert abs(pm["value"] - 128 * 500 / (pm["ms_per_step"] * 1e-3)) < 0.01 * pm["value"]
    got = {o["recipe"]: o for o in d["other_configs"]}
    assert sorted(got) == ["libri_gru", "timit_lstm", "timit_mlp", "timit_sincnet"]
    for name, o in got.items():
        assert "error" not in o, (name, o.get("error"))
        assert o["ms_per_step"] > 0 and o["value"] > 0 and "workload" in o["config"] and "kernel" in o["roofline"]
    r = d["roofline"]
    assert r["dependent_steps_per_launch"] == 500 and 0 < r["latency_frac"] < 1 and 0 < r["structure_frac"] <= 1
    assert "pmc_traffic" in r["traffic_source"]


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus N` with no rendezvous in the environment re-executes itself under torch.distributed.run
    with N ranks (here on CPU over gloo; on a GPU node the same path gives N RCCL ranks): rank 0 reports n_gpus = N and
    an all-reduce over all ranks' (rank + 1)."""
    import subprocess
    import sys

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["sum_of_ranks_plus_1"] == 3.0


def test_round3_line_carries_the_chunk_loop_and_the_reference_caller():
    """Round 3: the nearer roof by arithmetic intensity + a latency record, the same workload through run_nn_dp and as the
    reference's own caller drives it, parity-mode records of the LSTM / GRU configurations, the CPU port at the metric's
    full shape."""
    d = _line("profiles/r03_bench_bf16.json")
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["arithmetic_intensity"] < r["ridge"] and r["latency"]["bound"] == "latency"
    assert abs(r["hbm"]["frac"] - r["frac"]) < 1e-9 and 0 < r["mfma"]["frac"] < r["hbm"]["frac"]
    t = d["through_run_nn"]
    assert "error" not in t and abs(t["value"] - 128 * 500 / (t["ms_per_step"] * 1e-3)) < 0.01 * t["value"]
    assert t["ms_per_step"] < 1.05 * d["ms_per_step"]   # the chunk loop costs a few per cent at most
    rc = d["reference_caller"]
    assert "error" not in rc and rc["config"]["optimizer"] == "torch" and rc["config"]["host_sync"] == "every step"
    got = {o["recipe"]: o for o in d["other_configs"]}
    for name in ("timit_lstm", "libri_gru"):
        assert len(got[name]["regions_ms_per_step"]) == 3 and got[name]["steps"] >= 50
        assert got[name]["parity_mode"]["dtype"] == "fp32"
    full = json.load(open(os.path.join(ROOT, "profiles", "r03_cpu_full_shape.json")))
    assert (full["T"], full["B"]) == (500, 128) and full["kind"] == "port" and full["value"] > 0


def test_round4_line_carries_the_step_trace_the_prewarm_and_the_chunk_median():
    """Round 4: per-step GPU times of the timed region (where a slow phase would show), the untimed pre-warm the line was
    taken behind, the chunk-loop record as a median over chunks, and a full-shape CPU record that says it was not measured
    in this run."""
    d = _line("profiles/r04_bench_bf16.json")
    sm = d["step_ms"]
    assert sm["min"] <= sm["median"] <= sm["max"] and abs(sm["median"] - d["ms_per_step"]) < 0.05 * d["ms_per_step"]
    assert d["config"]["prewarm_s"] > 0 and d["config"]["prewarm_steps"] >= 4
    rn = d["through_run_nn"]
    assert len(rn["chunks_s"]) >= 3 and rn["elapsed_time_chunk_s"] == sorted(rn["chunks_s"][1:])[(len(rn["chunks_s"]) - 1) // 2]
    fs = d["cpu_baseline"]["full_shape"]
    assert fs["measured_in_run"] is False and "source" in fs
    r = d["roofline"]
    assert r["latency"]["floor_source"] == "profiles/r04_rec_step_floor.json" and "r04_pmc_traffic" in r["traffic_source"]
    assert 0.9 < r["structure_frac"] < 1.1  # the step IS its hand-off structure
    # the driver's command exactly as typed, first process on a fresh box
    e = _line("profiles/r04_driver_exact_cmd_line.json")
    assert e["steps"] == 20 and e["warmup"] == 5 and e["n_gpus"] == 1 and e["ms_per_step"] < 17.5


def test_round5_line_measures_the_cpu_full_shape_in_the_run_and_carries_the_round5_sources():
    """Round 5: the CPU port at the metric's FULL shape timed inside the bench run of the closing pass (--cpu-full-in-run),
    the roofline's latency floor / counter traffic from this round's profiles, the other configurations with the LSTM
    helpers and the bf16 convolutions on."""
    d = _line("profiles/r05_bench_bf16.json")
    fs = d["cpu_baseline"]["full_shape"]
    assert fs["measured_in_run"] is True and (fs["T"], fs["B"]) == (500, 128) and 100 < fs["value"] < 1000 and fs["seconds"] > 60
    r = d["roofline"]
    assert r["latency"]["floor_source"] == "profiles/r05_rec_step_floor.json" and "r05_pmc_traffic" in r["traffic_source"]
    assert d["ms_per_step"] < 17.5
    got = {o["recipe"]: o for o in d["other_configs"]}
    assert got["timit_lstm"]["ms_per_step"] < 24.5 and got["timit_sincnet"]["ms_per_step"] < 3.3
    assert got["libri_gru"]["parity_mode"]["ms_per_step"] < 600   # the step-wise fp32 products split over the grid (was 1 293)
    dc = json.load(open(os.path.join(ROOT, "profiles", "r05_driver_cmd.json")))
    first = dc["first_process"]
    assert first["steps"] == 20 and first["warmup"] == 5 and first["n_gpus"] == 1 and first["ms_per_step"] < 17.5


def test_round6_line_is_complete_per_configuration():
    """Round 6 (round-5 review, item 3): every other configuration carries its own CPU baseline, an fp32 row and PMC traffic;
    the headline's bounded CPU sample keeps the metric's sequence length (it no longer flatters the CPU: within 20 % of the
    full-shape step measured in the same run, and on the slow side - 16 sequences use the host's cores less well than 128); `structure_frac` is gone; the padded-batch variant is timed; the fp32 rows of
    the LSTM / GRU configurations run on the fourth-generation kernels."""
    d = _line("profiles/r06_bench_bf16.json")
    assert d["config"]["workload"].startswith("timit_ligru") and d["dtype"] == "bf16" and d["ms_per_step"] < 17.5
    r = d["roofline"]
    assert "structure_frac" not in r and "structure_frac" not in r["latency"] and 0 < r["latency_frac"] < 1
    assert r["traffic"] is not None and "r06_pmc_traffic" in r["traffic_source"] and 0 < r["frac"] < 1
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and (c["T"], c["B"]) == (500, 16) and c["cores"] >= 1
    fs = c["full_shape"]
    assert fs["measured_in_run"] is True and (fs["T"], fs["B"]) == (500, 128)
    assert abs(c["value"] - fs["value"]) < 0.20 * fs["value"] and c["value"] < 1.05 * fs["value"]
    pb = d["padded_batches"]
    assert "error" not in pb and 0 < pb["padding_share"] < 0.2 and pb["value"] > 0
    got = {o["recipe"]: o for o in d["other_configs"]}
    assert sorted(got) == ["libri_gru", "timit_lstm", "timit_mlp", "timit_sincnet"]
    for name, o in got.items():
        assert "error" not in o, (name, o.get("error"))
        assert o["cpu_baseline"]["value"] > 0 and o["cpu_baseline"]["kind"] in ("port", "reference"), name
        assert o["parity_mode"]["dtype"] == "fp32" and o["parity_mode"]["ms_per_step"] > o["ms_per_step"], name
        assert o["roofline"]["traffic"] is not None and "r06_pmc_traffic_" + name in o["roofline"]["traffic_source"], name
    # (fp32 rows: fourth-generation recurrences, weight gradients on the side stream, self-filling exchange)
    assert got["timit_lstm"]["parity_mode"]["ms_per_step"] < 140 and got["libri_gru"]["parity_mode"]["ms_per_step"] < 160
    assert d["parity_mode"]["ms_per_step"] < 88 and d["forward_mode"]["ms_per_step"] < 6.8
    assert len(pb["chunks_s"]) >= 2
    dc = json.load(open(os.path.join(ROOT, "profiles", "r06_driver_cmd.json")))
    assert dc["first_process"]["steps"] == 20 and dc["first_process"]["warmup"] == 5 and dc["first_process"]["ms_per_step"] < 17.5
