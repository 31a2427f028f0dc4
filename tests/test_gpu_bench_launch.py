"""bench.py as the driver starts it: `python bench.py --gpus N` with no launcher around it must start N ranks itself
(it used to run ONE rank and print n_gpus 1).  On the single-GPU test box the two ranks share device 0 over
gloo (the debug switches); with >= 2 GPUs the same command without them runs one rank per GPU on RCCL."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, timeout=900):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, capture_output=True, text=True, timeout=timeout, env=env)


@pytest.mark.gpu
def test_bench_gpus2_self_launches_two_ranks():
    shared = torch.cuda.device_count() < 2
    extra = ["--gpus", "2", "--steps", "2", "--warmup", "1", "--log-len", "22"]
    if shared:
        extra += ["--dist-backend", "gloo", "--force-device", "0"]
    r = _run(extra)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2
    assert out["check"]["sharded_root_equals_unsharded_root"] is True
    assert out["shard_ms"] is not None and out["value"] > 0
    assert ("DEBUG" in out["devices"]) == shared


@pytest.mark.gpu
@pytest.mark.parametrize("n", [2, 3])
def test_bench_native_exchange_across_processes(tmp_path, n):
    """`bench.py --gpus N` exactly as the driver runs it on a multi-GPU node -- one process per rank, the library's OWN exchange
    (lcpc_comm_init with the id carried by torch.distributed, the timed loop of LCPC_COMMIT_ASYNC_TAIL commits on two commitments
    per rank, the cross-check against the torch exchange and against an unsharded commit) -- on the one-GPU box: the ranks share
    device 0, torch.distributed runs over gloo, and RCCL (which refuses two ranks on one device) is replaced by the
    shared-memory stand-in tests/native/fake_rccl_shm.cpp through LCPC_RCCL_LIB.  No fallback to the torch exchange may happen."""
    so = str(tmp_path / "libfake_rccl_shm.so")
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cc = subprocess.run(["g++", "-O2", "-shared", "-fPIC", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I" + rocm + "/include",
                         os.path.join(ROOT, "tests", "native", "fake_rccl_shm.cpp"), "-o", so, "-L" + rocm + "/lib", "-lamdhip64", "-lrt"],
                        capture_output=True, text=True, timeout=300)
    assert cc.returncode == 0, cc.stderr[-3000:]
    env = dict(os.environ, LCPC_RCCL_LIB=so)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "3", "--warmup", "1", "--log-len", "22",
                        "--dist-backend", "gloo", "--force-device", "0", "--exchange", "native"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == n and out["ranks_seen"] == n
    assert "exchange_fallback" not in out, out.get("exchange_fallback")
    assert "RCCL inside the library" in out["config"]["sharding"] and "LCPC_COMMIT_ASYNC_TAIL" in out["config"]["sharding"]
    assert out["check"]["sharded_root_equals_unsharded_root"] is True
    assert out["shard_ms"]["async_tail"] is True and out["min_ms_per_step"] is None and out["value"] > 0
    # one run tells the whole scaling story: the serial figure beside the pipelined one, the wire measured on the real
    # payload, per-rank phase times, the communicator library's version, and the async-tail roots looked at
    assert out["serial_ms_per_step"] > 0 and out["serial_min_ms_per_step"] > 0 and out["serial_value"] > 0
    assert out["serial_min_ms_per_step"] <= out["serial_ms_per_step"] * 1.001
    pr = out["exchange_probe"]
    n_cols = out["config"]["n_cols"]
    assert pr["reps"] == 20 and pr["ms"] > 0 and pr["GBps_in"] > 0
    assert pr["bytes_in"] % (n_cols * 32) == 0 and n_cols * 32 * (n - 1) <= pr["bytes_in"] <= n_cols * 32 * (n + 1)
    per = out["per_rank_ms"]
    for k in ("encode", "hash", "exchange_exposed", "finish_tree"):
        lo, hi = per[k]
        assert 0 <= lo <= hi
    assert per["encode"][1] > 0
    assert "rccl_version" in out and "async_tail_roots_checked" in out
    assert "ChaCha20Rng" in out["config"]["input"]


@pytest.mark.gpu
def test_bench_n1_json_contract():
    """the N = 1 line the driver records: one JSON object carrying the metric, `roofline` (bound / achieved / peak / frac with
    frac == achieved / peak, per-launch algorithmic bytes and launch time) and `cpu_baseline` (value, cores, kind, sample, and
    the same coefficients' root through the HIP path == the CPU port's root)"""
    r = _run(["--gpus", "1", "--steps", "2", "--warmup", "1", "--log-len", "24", "--no-power-sample"])
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline"):
        assert k in out, k
    assert out["n_gpus"] == 1 and out["steps"] == 2 and out["warmup"] == 1 and out["higher_is_better"] is True and out["vs_baseline"] is None
    assert out["unit"] == "field-elements/s" and out["data"] == "synthetic"
    assert abs(out["value"] - (1 << 24) / (out["ms_per_step"] * 1e-3)) <= 1e-6 * out["value"]
    cfg = out["config"]
    assert "lcpc-ligero-pc commit" in cfg["workload"] and "Ft255" in cfg["workload"] and "2^24" in cfg["workload"] and "model" not in cfg
    assert cfg["n_rows"] * cfg["n_per_row"] == 1 << 24 and cfg["n_cols"] == 2 * cfg["n_per_row"]
    rf = out["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "algorithmic_GB_per_launch", "avg_launch_ms"):
        assert k in rf, k
    assert rf["peak"] == 8000.0 and rf["unit"] == "GB/s" and rf["achieved"] > 0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) <= 1e-3 * rf["frac"] + 1e-4
    assert abs(rf["achieved"] - rf["algorithmic_GB_per_launch"] / (rf["avg_launch_ms"] * 1e-3)) <= 0.01 * rf["achieved"]
    cb = out["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample", "root_equals_hip_root"):
        assert k in cb, k
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == out["unit"] and cb["root_equals_hip_root"] is True
    # the inputs are SURVEY 8(d)'s (named in config.input) and the baseline leg committed the timed vector itself
    assert "ChaCha20Rng::from_seed([0; 32])" in cfg["input"] and cb["coeffs_equal_timed_coeffs"] is True
    # the host-pointer entry from pinned AND from pageable memory, same root as the device-resident commit
    e2e = out["e2e_host"]
    assert e2e["root_matches_device_commit"] is True and e2e["staged_slices"] == 0
    assert e2e["pageable"]["root_matches_device_commit"] is True and e2e["pageable"]["staged_slices"] > 0 and e2e["pageable"]["vs_pinned"] > 0
    # BASELINE.json's other configs, timed on this box in the untimed region and checked against the oracle on the timed inputs
    cf = out["configs"]
    assert set(cf) == {"C1", "C2", "C3", "C5"}
    for k in ("C1", "C2", "C3"):
        c = cf[k]
        assert c["checked"] is True and c["reps"] >= 10 and 0 < c["min_ms"] <= c["ms"] and c["algorithmic_GBps"] > 0, (k, c)
    assert cf["C1"]["dims"] == [32, 2048, 4096] and cf["C2"]["dims"] == [256, 65536, 131072] and cf["C3"]["dims"] == [101, 166292, 252931]
    assert "brakedown" in cf["C3"]["workload"] and cf["C3"]["encoder_build_s"] > 0
    c5 = cf["C5"]
    assert c5["checked"] is True and "oracle verifier accepts" in c5["check"]
    for k in ("prove", "verify"):
        assert c5[k]["reps"] >= 10 and 0 < c5[k]["min_ms"] <= c5[k]["ms"]


@pytest.mark.gpu
def test_bench_refuses_more_ranks_than_devices():
    n = torch.cuda.device_count() + 1
    r = _run(["--gpus", str(n), "--steps", "1", "--warmup", "0", "--log-len", "20"], timeout=300)
    assert r.returncode != 0
    assert "refusing" in r.stderr and not any(l.startswith("{") for l in r.stdout.splitlines())


def test_bench_without_device_fails_loudly():
    """no GPU in this container: N = 1 and N > 1 both exit non-zero with a message, never a fabricated line"""
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    for n in ("1", "2"):
        r = _run(["--gpus", n, "--steps", "1", "--warmup", "0"], timeout=300)
        assert r.returncode != 0 and not any(l.startswith("{") for l in r.stdout.splitlines())
        assert "GPU" in r.stderr or "device" in r.stderr
