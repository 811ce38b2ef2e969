"""The driver's contract on the line bench.py prints (the prompt's (4)): one JSON object with the metric, the
whole-job value, cold and sustained timing with spread, a `roofline` object for the dominant kernel measured with
hipEvents inside the timed region, and a `cpu_baseline` object on rank 0 at N = 1.  A short run of the real
script on the GPU box."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags, timeout=600):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line"
    return json.loads(lines[0])


def test_bench_line_fields_and_consistency():
    # (--no-residency: the in-run co-runner probe loads the EXPERIMENTS build in a child process; the test suite runs
    #  the product library only -- the probe has a test of its own below, off unless asked for)
    d = _run("--steps", "6", "--warmup", "2", "--regions", "2", "--no-extras", "--no-residency")
    assert d["metric"].startswith("Msamp/s/antenna") and d["unit"] == "Msamp/s" and d["higher_is_better"] is True
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2 and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert "configs[1]" in d["config"]["workload"] and "model" not in d["config"]
    # value = dual-pol samples of one second / median region
    assert abs(d["value"] - 128e6 / (d["ms_per_step"] * 1e-3) / 1e6) / d["value"] < 1e-3
    tr = d["timed_regions"]
    assert tr["n"] == 2 and tr["ms_per_step_min"] <= d["ms_per_step"] <= tr["ms_per_step_max"]
    assert d["ms_per_step_cold"] > 0 and d["precondition_steps"] == 100
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s"
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) / r["achieved"] < 1e-2
    assert r["avg_launch_ms"] <= d["ms_per_step"] * 1.05            # a kernel of the step cannot outlast the step
    assert 0.05 < r["frac"] < 1.0 and 0.05 < r["pipeline"]["frac"] < 1.0
    assert r["kernel"] == "channelize" and "kurtosis" not in d["stage_ms_per_step"]      # the channeliser flags its own rows
    # what the step runs into (round 4): the package power cap, reported beside the pipes
    pw = r["power"]
    if "error" not in pw:                   # (amdsmi may be unavailable to an unprivileged user on some hosts)
        assert 300 < pw["socket_w"] <= pw["cap_w"] * 1.02 and pw["cap_w"] >= 500
        assert 500 <= pw["gfx_mhz"] <= 2500 and 0.0 <= pw["power_throttle_residency"] <= 1.0
    assert r["residency"] is None           # (switched off above)
    sm = r["survey_model"]                  # SURVEY 8(d)'s unfused-chain bytes at this run's rate
    assert sm["bytes_per_step"] == 8459000000 and abs(sm["frac"] - sm["implied"] / 8000.0) < 1e-3
    assert sm["implied"] > r["pipeline"]["achieved"]
    v = r["valu"]
    if v is not None:                       # (None when no committed PMC summary matches the kernels' source hash)
        assert "error" not in v, v
        assert 0.05 < v["frac"] < 1.0 and v["insts_per_step"] > 1e8
        assert r["traffic"] is None or 0.9 < r["traffic"] / r["algorithmic_bytes_per_launch"] < 1.2
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "Msamp/s" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    assert d["value"] / c["value"] > 50                                # (reported only; sanity of units)
    hp = c["hot_path"]                                                 # the oracle's K1-K11 chain on the same workload, one core
    assert "error" not in hp, hp
    assert hp["kind"] == "port" and hp["cores"] == 1 and 1 < hp["value"] < c["value_1core"] * 2 and "configs[1]" in hp["sample"]


@pytest.mark.skipif(os.environ.get("PB_TEST_EXPERIMENTS_BUILD") != "1",
                    reason="loads libpb_hip_exp.so in a child process: only on request (PB_TEST_EXPERIMENTS_BUILD=1)")
def test_bench_measures_residency_in_run_with_the_experiments_build():
    """roofline.residency: measured by the run where the experiments build (make -C vlite-fast_amd/csrc exp) and
    build/libcorun.so are in the tree and as new as the kernel sources; quoted from a committed record of THESE kernel
    sources otherwise; absent else.  Only a measurement is asserted on (a quote is a committed file compared with
    itself)."""
    d = _run("--steps", "6", "--warmup", "2", "--regions", "1", "--no-extras", "--no-cpu-baseline", "--no-power")
    rs = d["roofline"]["residency"]
    assert rs is not None and "error" not in rs, rs
    assert rs["kernel_source_sha16"] and rs["measured_in_run"] is True, rs
    cm = rs["channelize_ms_per_launch"]
    assert cm["alone"] < cm["beside_256_sleeping_57KB_workgroups"] <= cm["beside_detect"] * 1.1
    assert rs["socket_w"]["beside_256_sleeping_57KB_workgroups"] < rs["socket_w"]["alone"]


def test_bench_two_kernel_path_still_reports_kurtosis_stage():
    env = dict(os.environ, PB_FUSE_KURTOSIS="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "1", "--regions", "1",
                        "--no-extras", "--no-cpu-baseline", "--no-residency"], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([l for l in p.stdout.strip().splitlines() if l.startswith("{")][0])
    assert set(d["stage_ms_per_step"]) >= {"kurtosis", "channelize", "detect"} and "cpu_baseline" not in d
