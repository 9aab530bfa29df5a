"""The bench line says what ran: a recorded driver line (profiles/r05_bench.json, written by `python bench.py` on the
MI355X) carries the contract's keys, the renamed sample fractions, the whole-frame fraction and one frame time per parameter set -
and none of the labels round 4 was faulted for."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LINE = next((p for p in (os.path.join(ROOT, "profiles", f"{r}_bench.json") for r in ("r06", "r05")) if os.path.exists(p)),
            os.path.join(ROOT, "profiles", "r05_bench.json"))
ROUND6 = os.path.basename(LINE).startswith("r06")


@pytest.fixture(scope="module")
def line():
    if not os.path.exists(LINE):
        pytest.skip("no recorded bench line")
    with open(LINE) as f:
        return json.load(f)


def test_contract_keys(line):
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["n_gpus"] == 1 and line["unit"] == "rays/s" and line["higher_is_better"] is True and line["vs_baseline"] is None
    assert line["data"] == "synthetic" and "workload" in line["config"] and "model" not in line["config"]
    assert abs(line["value"] - 512 * 512 / (line["ms_per_step"] * 1e-3)) < 1e-6 * line["value"]
    rf = line["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] == "mfma" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9 and 0.0 < rf["frac"] < 1.0
    cb = line["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"]
    if ROUND6:
        # the torch restatement (north_star's "reference CPU PyTorch path") is THE cpu_baseline, the C port beside it;
        # the roofline block carries the whole frame's fraction and the matrix pipe's busy share next to the kernel's
        assert "torch" in cb["sample"] and line["cpu_baseline_c"]["kind"] == "port" and "OpenMP" in line["cpu_baseline_c"]["sample"]
        assert abs(rf["whole_frame_frac"] - line["whole_frame"]["frac"]) < 1e-12 and 0.0 < rf["whole_frame_frac"] < rf["frac"]
        assert rf["mfma_busy"] is None or 0.3 < rf["mfma_busy"] < 1.0
        assert "4e-6" in line["dtype"] and "1e-4" in line["dtype"]


def test_sample_fractions_say_what_each_stage_ran_on(line):
    c = line["config"]
    assert "evaluated_sample_fraction" not in c and "dense_equivalent_tflops" not in c
    nt, fw, rv, sh, ps = (c[k] for k in ("non_transparent_sample_fraction", "forward_sample_fraction", "reverse_sample_fraction",
                                         "shaded_sample_fraction", "positive_density_sample_fraction"))
    assert 0.0 < rv <= ps <= fw <= nt < 1.0 and sh <= ps
    assert "dense_equivalent_work_rate_NOT_throughput_tflops" in c


def test_whole_frame_fraction_and_per_checkpoint_times(line):
    wf = line["whole_frame"]
    flop = wf["forward_samples"] * 2.0 * 458880.0 + wf["reverse_samples"] * 2.0 * 425728.0
    assert abs(wf["tflop_executed"] - flop / 1e12) < 1e-9 * flop
    assert abs(wf["frac"] - wf["tflop_executed"] / (line["ms_per_step"] * 1e-3) / wf["peak"]) < 1e-9
    assert 0.0 < wf["frac"] < line["roofline"]["frac"] < 1.0          # the whole frame cannot beat its dominant kernel
    by = line["ms_per_frame_by_weights"]
    assert set(by) >= {"default", "w4"} and by[line["config"]["weights"]] == pytest.approx(line["ms_per_step"])
    # the reverse kernel as the sliced frames run it, the one-pass figure kept beside it
    rk = line["roofline"]["reverse_kernel"]
    if line["early_stop"]["enabled"]:
        assert rk["samples_per_launch"] == wf["reverse_samples"]
        assert "reverse_kernel" in line["roofline"]["single_launch_on_all_non_transparent_samples"]
    assert "ms_per_frame" in line["config"]["host_to_host_ms_pipelined"]
