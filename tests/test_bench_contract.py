"""bench.py prints one JSON line with the contract's keys (short run on the GPU box)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT


@pytest.mark.gpu
def test_bench_line(gpu):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "50", "--warmup", "10",
                        "--no-finest", "--cpu-samples", "1"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in out, k
    assert out["n_gpus"] == 1 and out["steps"] == 50 and out["dtype"] == "f64" and out["vs_baseline"] is None
    assert "workload" in out["config"] and "model" not in out["config"]
    rf = out["roofline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    # the three fractions side by side; the fused pass streams the face data once, so its unique bytes are fewer
    assert 0 < rf["frac_unique"] < rf["frac"] and rf["unique_bytes_per_launch"] < rf["algorithmic_bytes_per_launch"]
    assert "traffic_source" in rf and (rf["traffic"] is None or rf["traffic_source"]["file"].startswith("profiles/"))
    assert out["config"]["baseline_config"] == "dualgrid.12" and out["scaling"] == "strong"
    cb = out["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["value"] > 0
    if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "ref_dump_raw")):
        assert cb["kind"] == "reference" and cb["port"]["kind"] == "port", cb
    assert out["value"] > 0 and out["ms_per_step"] > 0


def test_bench_configs_name_the_baseline_workloads(pkg):
    """BASELINE.json configs -> bench workloads: defaults by --gpus, whole domains per GPU, scaling label"""
    from cfd_proxy_amd import multigpu as mg
    assert [mg.default_bench_config(n) for n in (1, 2, 4, 8)] == ["dualgrid.12", "dualgrid.24", "dualgrid.48", "dualgrid.384"]
    c = mg.bench_config("dualgrid.48", 4)
    assert c["workload"].startswith("dualgrid.48 lvl 2 stand-in (64^3, 48 domains, 12 per GPU") and c["scaling"] == "strong"
    c = mg.bench_config("dualgrid.192", 8)
    assert c["dims"] == (64, 64, 64) and c["ndomains"] == 192 and "24 per GPU" in c["workload"] and c["scaling"] == "strong"
    c = mg.bench_config("dualgrid.384", 8)
    assert c["dims"] == (128, 128, 128) and "48 per GPU" in c["workload"] and c["scaling"] == "weak"
    assert mg.bench_config("weak", 4)["dims"] == (128, 128, 64)
    with pytest.raises(ValueError):
        mg.bench_config("dualgrid.12", 8)  # 12 domains do not divide over 8 GPUs
    with pytest.raises(ValueError):
        mg.bench_config("dualgrid.13", 1)


def _bench_two_ranks(transport, extra_env=None, expect=None, weak=False):
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), CFDP_SHARED_GPU="1", **(extra_env or {}))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20",
                                       "--warmup", "3", "--transport", transport, "--no-files"] + ([] if weak else ["--no-weak"]),
                                      env=env, cwd=ROOT,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, so[-1500:] + se[-1500:]
    line = [l for l in outs[0][0].splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    # --gpus 2 continues the strong series on the level-2 mesh: 64^3 cut into 24 domains, 12 per GPU
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["config"]["points_per_gpu"] == 131072
    assert out["config"]["baseline_config"] == "dualgrid.24" and "24 domains, 12 per GPU" in out["config"]["workload"]
    assert abs(out["value"] - out["config"]["mesh_iterations_per_s"]) <= 1e-9 * out["value"]  # one level-2 mesh = one unit
    if weak:  # the weak-scaling point of the same run rides along
        w = out["weak_scaling"]
        assert w["scaling"] == "weak" and w["config"]["points_per_gpu"] == 262144 and w["exchange_check"]["ok"], w
        assert abs(w["value"] - 2 * w["config"]["mesh_iterations_per_s"]) <= 1e-9 * w["value"]
    assert out["config"]["ghost_points_per_gpu"] > 0 and "overlap" in out
    assert out["overlap"]["efficiency_async"] > 0  # (its size means nothing between ranks time-slicing one GPU)
    assert out["exchange_check"]["ok"], out["exchange_check"]
    assert out["config"]["transport"] == (expect or ("ipc" if transport == "auto" else transport))
    assert out["config"]["fused_iterations"]
    assert "cpu_baseline" not in out
    return out


@pytest.mark.gpu
def test_bench_two_ranks_staged_transport(gpu):
    """the N>1 code path of bench.py (mesh partitioning across ranks, request exchange, overlapped
    halo exchange, overlap report) with 2 ranks sharing this GPU and the host-staged transport"""
    _bench_two_ranks("staged")


@pytest.mark.gpu
def test_bench_two_ranks_xgmi_write_notify(gpu):
    """bench.py --gpus 2 with its default transport (xGMI write + notify through HIP IPC, steps
    replayed from hipGraphs), the two ranks sharing this GPU; every sent row must have arrived"""
    out = _bench_two_ranks("auto", weak=True)  # on a shared GPU only the ipc transport can be set up (RCCL needs one device per rank)
    assert out["exchange_check"]["wait_timeouts"] == 0 and "ipc" in out["config"]["transport_probe_us_per_iteration"]


@pytest.mark.gpu
def test_bench_falls_back_when_the_exchange_check_fails(gpu):
    """a transport whose exchange check fails after the timed region is dropped and the measurement is
    repeated on the next one (here: injected failure of the first check; ipc -> host-staged, the only
    other transport two ranks sharing one GPU have)"""
    out = _bench_two_ranks("auto", extra_env={"CFDP_BENCH_REJECT_FIRST": "1"}, expect="staged")
    assert out["exchange_check"]["transports_rejected"] == ["ipc"]
