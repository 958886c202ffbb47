"""bench.py's workloads run end to end at reduced user counts (one JSON line each, the contract's keys, the right kernel):
keeps every branch of the benchmark - headline, default arrays, beam power, time domain, rx_filter, Doppler - from
rotting between the rounds' full-size runs."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CASES = [("c3_headline", 300, "k2_fd_mfma", "hbm"), ("c5_massive", 12, "k2_fd_mfma", "hbm"), ("d8_default_arrays", 2000, "k2_fd_fold", "hbm"),
         ("d64_k256", 500, "k2_fd_fold", "hbm"), ("c3_beam_power", 300, "k2c_beam_power", "mfma"),
         ("c3_time_domain", 500, "k4_td", "hbm"), ("c3_rx_filter", 200, "k3_lpf_fft512 + k2_fd_mfma (table-fed)", "hbm")]


@pytest.mark.parametrize("workload,users,kernel,bound", CASES, ids=[c[0] for c in CASES])
def test_bench_workload_line(workload, users, kernel, bound):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--users", str(users), "--steps", "2",
                        "--warmup", "1", "--cpu-users", "0"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["metric"] == "user-channels/sec" and d["n_gpus"] == 1 and d["steps"] == 2 and d["vs_baseline"] is None
    assert d["config"]["fd_kernel"] == kernel and d["config"]["users_total"] == users
    rf = d["roofline"]
    assert rf["bound"] == bound and rf["achieved"] > 0 and rf["frac"] == pytest.approx(rf["achieved"] / rf["peak"])
    assert d["value"] == pytest.approx(users / (d["ms_per_step"] * 1e-3), rel=1e-6)
    if workload == "c5_massive":
        assert "Doppler" in d["config"]["workload"]
    # the adaptive-precision rule's share is on the line for the kernels that have it, and only for them
    has_rule = kernel in ("k2_fd_mfma", "k2_fd_fold", "k2c_beam_power")
    assert ("kernel_ms_adaptive_terms" in rf) == has_rule
    if has_rule:
        assert rf["kernel_ms_adaptive_terms"] > 0 and rf["frac_adaptive_terms"] == pytest.approx(rf["frac"] * rf["kernel_ms"] / rf["kernel_ms_adaptive_terms"])
        assert "DMX_FLAG_ADAPTIVE_TERMS" in d["config"]["arithmetic"]
