"""The benchmark's output contract: one JSON line with the fields the driver and the reviewers read."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_prints_one_json_line_with_the_contract_fields():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--ntt-batch", "4",
                        "--merkle-n", "8192", "--proofs", "6", "--sharded-log-n", "22"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "merkle", "batch_verify", "parity"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["value"] > 0
    assert "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    cb = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] in ("port", "reference") and cb["cores"] == 1
    assert all(d["parity"].values())
    assert d["batch_verify"]["accept_vector_as_constructed"] is True and d["batch_verify"]["value"] > 0
    # forward and inverse are reported separately, so are BASELINE's configurations 2 and 3
    assert d["inverse"]["value"] > 0 and d["inverse"]["launch_us"] > 0
    assert d["cfg2"]["forward_value"] > 0 and d["cfg2"]["inverse_value"] > 0 and d["cfg2"]["inverse_of_forward_equals_input"] is True
    sh = d["sharded_ntt"]
    assert sh["equals_single_gpu_call"] is True and sh["value"] > 0 and sh["all_to_all_bytes_sent_per_gpu"] == 0
    for k in ("stage_a_ms", "all_to_all_ms", "stage_c_ms", "single_gpu_call_ms_per_step", "contiguous_in_and_out_ms_per_step"):
        assert k in sh, k
