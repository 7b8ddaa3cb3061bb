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
                        "--merkle-n", "8192", "--merkle-total", "8192", "--proofs", "6", "--proofs-total", "12", "--sharded-log-n", "22"], capture_output=True, text=True, timeout=600, cwd=ROOT)
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
    assert "regions_failed" not in d
    # the configurations that are stated as totals (strong scaling): 1 M branches and 10 k proofs in the default run
    ms, bs = d["merkle_strong"], d["batch_verify"]["strong"]
    assert ms["scaling"] == "strong" and ms["branches_total"] == 8192 and ms["accept_vector_as_constructed"] is True
    assert [v["branches_per_rank"] for v in ms["slice_of_one_rank_at_n_gpus"].values()] == [4096, 2048, 1024]
    assert bs["scaling"] == "strong" and bs["proofs_total"] == 12 and bs["accept_vector_as_constructed"] is True
    assert [v["proofs_per_rank"] for v in bs["slice_of_one_rank_at_n_gpus"].values()] == [6, 3, 2]
    assert rf["traffic_source"]["file"] == "profiles/traffic.json" and rf["traffic_source"]["round"]
    assert cb["merkle_all_cores_value"] > 0 and cb["all_cores_value"] > 0
    # forward and inverse are reported separately, so are BASELINE's configurations 2 and 3
    assert d["inverse"]["value"] > 0 and d["inverse"]["launch_us"] > 0
    assert d["cfg2"]["forward_value"] > 0 and d["cfg2"]["inverse_value"] > 0 and d["cfg2"]["inverse_of_forward_equals_input"] is True
    sh = d["sharded_ntt"]
    assert sh["equals_single_gpu_call"] is True and sh["value"] > 0 and sh["all_to_all_bytes_sent_per_gpu"] == 0
    for k in ("stage_a_ms", "all_to_all_ms", "stage_c_ms", "single_gpu_call_ms_per_step", "contiguous_in_and_out_ms_per_step"):
        assert k in sh, k


GUARD_SCRIPT = r"""
import sys, time
sys.path.insert(0, %r)
import bench
g = bench.RegionGuard(int(sys.argv[1]), 2, {"metric": "m", "value": 1.0, "parity": {"all_ranks": True}} if sys.argv[1] == "0" else None)
g.arm("first", 30.0)
g.disarm()                      # a region that finished: its timer must never act
if sys.argv[2] == "hang":
    g.arm("sharded_ntt", 0.5)
    time.sleep(20)              # stands for a collective that never completes
    print("not reached")
elif sys.argv[2] == "alone":
    g.arm("batch_verify", 30.0)
    g.failed_alone("batch_verify", "RuntimeError: only this rank")
    print("not reached")
else:
    g.arm("sharded_ntt", 30.0)
    g.disarm()
    g.print_line()
    g.print_line()              # the line is printed once
"""


@pytest.mark.parametrize("rank,mode", [(0, "hang"), (1, "hang"), (0, "alone"), (1, "alone"), (0, "fine")])
def test_region_guard_keeps_the_line_when_a_later_region_does_not_finish(rank, mode):
    """bench.py's regions after the headline run under a timer: when one does not finish (or a rank fails alone in a
    region with barriers), rank 0 prints the line it holds with an error record for that region, the region listed in
    regions_failed and parity.all_ranks false, and leaves with a non-zero status once the line is flushed; the other
    ranks print nothing and leave with status 0 (anything else makes the launcher stop rank 0 before it has printed)."""
    r = subprocess.run([sys.executable, "-c", GUARD_SCRIPT % ROOT, str(rank), mode], capture_output=True, text=True, timeout=300, cwd=ROOT)
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    if rank != 0:
        assert r.returncode == 0, r.stderr[-2000:]
        assert lines == []
        return
    assert r.returncode == (0 if mode == "fine" else 3), (r.returncode, r.stderr[-2000:])
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["metric"] == "m" and d["value"] == 1.0 and "first" not in d
    if mode == "hang":
        assert "did not finish" in d["sharded_ntt"]["error"]
        assert d["regions_failed"] == ["sharded_ntt"] and d["parity"]["all_ranks"] is False
    elif mode == "alone":
        assert d["batch_verify"]["error"] == "RuntimeError: only this rank"
        assert d["regions_failed"] == ["batch_verify"] and d["parity"]["all_ranks"] is False
    else:
        assert "sharded_ntt" not in d and "regions_failed" not in d and d["parity"]["all_ranks"] is True


@pytest.mark.gpu
def test_bench_fails_loudly_when_a_region_fails_on_one_rank():
    """STARKHIP_BENCH_FAIL=sharded_ntt:0 makes the sharded region raise: the line is still printed, names the region in
    regions_failed, and the process status is non-zero."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--ntt-batch", "2",
                        "--merkle-n", "4096", "--merkle-total", "0", "--proofs", "0", "--cfg2-batch", "0", "--sharded-log-n", "16",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=ROOT,
                       env=dict(os.environ, STARKHIP_BENCH_FAIL="sharded_ntt:0"))
    assert r.returncode != 0
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, (lines, r.stderr[-2000:])
    d = json.loads(lines[0])
    assert d["regions_failed"] == ["sharded_ntt"] and "rehearsal" in d["sharded_ntt"]["error"]
    assert d["parity"]["all_ranks"] is False and d["value"] > 0
