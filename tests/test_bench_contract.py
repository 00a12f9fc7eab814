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
    assert 0 < rf["movement_only_us"] < rf["us_per_launch"]  # the pass without its face loops: the data-movement floor
    assert "traffic_source" in rf and (rf["traffic"] is None or rf["traffic_source"]["file"].startswith("profiles/"))
    # socket power and shader clock beside the dominant kernel (rocm-smi; the face loops run at the power cap: DESIGN 8)
    pw = rf["power"]
    assert "socket_power_w" in pw and (pw["socket_power_w"] is None and pw.get("note")
                                       or 200 < pw["socket_power_w"] <= 1.1 * (pw["power_cap_w"] or 1400) and 400 < pw["shader_clock_mhz"] <= 2500), pw
    assert out["config"]["baseline_config"] == "dualgrid.12" and out["scaling"] == "strong"
    cb = out["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["value"] > 0
    assert cb["cpu_model"] and cb["threads"] == cb["cores"] and cb["n_median"] == 1 and cb["niter"] == 25 and cb["thread_binding"], cb
    if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "ref_dump_raw")):
        assert cb["kind"] == "reference" and cb["port"]["kind"] == "port", cb
    assert out["value"] > 0 and out["ms_per_step"] > 0
    # the timed region sits behind ~0.25 s of the same iterations (sustained clock); the figure without is reported beside it
    cc = out["clock_conditioning"]
    assert cc["seconds"] > 0 and cc["steps"] >= 100 and cc["steps"] % 50 == 0 and cc["unconditioned_ms_per_step"] > 0, cc
    # the exchange protocol's own cost, measured in loopback on this GPU (an upper bound of the overlap efficiency)
    lb = out["exchange_protocol_loopback"]
    for name in ("dualgrid.384", "dualgrid.192"):
        e = lb[name]
        assert e["wait_timeouts"] == 0 and 0.5 < e["efficiency_bound"] <= 1.05 and e["protocol"]["notify"] == "per partner", e
        assert e["us_per_iteration_with_exchange"] >= 0.95 * e["us_per_iteration_comm_free"], e
    assert "wall_s" in out and out["wall_s"]["exchange protocol in loopback"] < 90
    # notification by counters (the default) beside the flag form and the RCCL fall-back, same partitions, same table
    for name in ("dualgrid.384", "dualgrid.192"):
        e = lb[name]
        assert e["protocol"]["notify_by"].startswith("counters") and e["graph_replay"]["captures_failed"] == 0, e
        assert e["flag_notification"]["wait_timeouts"] == 0 and 0.5 < e["flag_notification"]["steps20_ratio"] <= 1.05, e
        r_ = e["rccl_self_sendrecv"]
        assert "error" in r_ or (r_["rccl_nranks"] == 1 and r_["steps20_with_exchange"] > 0), r_
    # the same kernels on an irregular mesh of the same size (the lattice is the best case): three fractions, the floor, the
    # tiles the tiler made of it and which kernel instantiations ran
    im = out["irregular_mesh"]
    assert "error" not in im, im
    assert im["points"] == 262144 and 6.5 < im["faces_per_point"] < 7.2 and im["incidences_per_point"]["max"] >= 40, im
    assert im["points_per_tile"] > 56 and im["tiles"] < 4700 and im["launch_groups"][0]["class"] in ("small", "large"), im
    assert 0 < im["fused"]["frac_unique"] < im["fused"]["frac"] < 1 and 0 < im["gradient_kernel"]["frac"] < 1, im
    assert 0 < im["fused"]["movement_only_us"] < im["fused"]["us_per_launch"], im
    assert any(f.startswith("fused_split<") for f in im["kernel_forms"]) and not any("generic" in f for f in im["kernel_forms"]), im
    assert 0.9 < im["lattice_over_irregular"]["fused_frac"] < 1.6, im
    # what stood in front of the timed region is in `config` too (the part of the line a driver's record keeps)
    assert out["config"]["unconditioned_value"] == cc["unconditioned_value"] and out["config"]["untimed_steps_in_front_of_the_timed_region"] >= 110
    assert "cgroup" in cb["cores_note"]
    # one rank, one device, nothing shared; the CFDP_* variables the run saw are in the line
    assert out["shared_gpu"] is False and out["config"]["distinct_devices"] == 1 and len(out["config"]["device_of_rank"]) == 1
    assert isinstance(out["config"]["env"], dict) and all(k.startswith("CFDP_") for k in out["config"]["env"])


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
    # what rides along with a run: the weak point at 2 / 4 GPUs, BASELINE config 4 at 8 (and nothing twice)
    assert mg.bench_extra("dualgrid.24", 2) == ("weak_scaling", "weak") and mg.bench_extra("dualgrid.48", 4) == ("weak_scaling", "weak")
    assert mg.bench_extra("dualgrid.384", 8) == ("strong_scaling", "dualgrid.192")
    assert mg.bench_extra("dualgrid.192", 8) is None and mg.bench_extra("weak", 4) is None and mg.bench_extra("dualgrid.12", 1) is None
    assert mg.bench_config(mg.bench_extra("dualgrid.384", 8)[1], 8)["ndomains"] == 192
    with pytest.raises(ValueError):
        mg.bench_config("dualgrid.12", 8)  # 12 domains do not divide over 8 GPUs
    with pytest.raises(ValueError):
        mg.bench_config("dualgrid.13", 1)


def _visible_devices():
    import torch
    return max(torch.cuda.device_count(), 1)


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
                                       "--warmup", "3", "--transport", transport, "--no-files", "--no-cpu"] + ([] if weak else ["--no-weak"]),
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
    # the post-run leg in the scaled field: no flux phase of 50 more steps read a ghost row of an earlier exchange
    sr = out["exchange_check"]["stale_read_check"]
    assert sr["ok"] and sr["stale_reads"] == 0 and sr["steps"] == 50 and sr["flux_fields_compared_per_rank"] >= 49, sr
    assert out["config"]["transport"] == (expect or ("ipc" if transport == "auto" else transport))
    assert out["config"]["fused_iterations"]
    assert "cpu_baseline" not in out  # (--no-cpu here; the self-launched run below carries it)
    assert out["clock_conditioning"]["steps"] % 50 == 0 and out["clock_conditioning"]["unconditioned_ms_per_step"] > 0
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


# ---- `python bench.py --gpus N` started plainly (no launcher, no rank environment) starts its N ranks itself ----
_FAKE_RANK = r"""
import json, os, sys, time
r, n = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["MASTER_PORT"]) > 0 and "LOCAL_RANK" in os.environ
mode = sys.argv[1]
if mode == "fail" and r == 1:
    sys.exit(7)
if mode == "hang" or (mode == "fail" and r != 1):
    time.sleep(600)
if r == 0:
    print(json.dumps({"n_gpus": n, "local_rank": os.environ["LOCAL_RANK"], "argv": sys.argv[1:]}))
"""


def test_launcher_starts_the_ranks_and_forwards_rank0(capfd):
    import bench
    cmd = [sys.executable, "-c", _FAKE_RANK]
    assert bench.launch_ranks(3, ["ok", "--steps", "5"], ndev=3, child_cmd=cmd, timeout=60) == 0
    out = json.loads([l for l in capfd.readouterr().out.splitlines() if l.startswith("{")][-1])
    assert out == {"n_gpus": 3, "local_rank": "0", "argv": ["ok", "--steps", "5"]}


def test_launcher_fails_the_job_when_a_rank_fails_or_hangs(capfd, monkeypatch):
    import time
    import bench
    cmd = [sys.executable, "-c", _FAKE_RANK]
    t = time.time()
    assert bench.launch_ranks(3, ["fail"], ndev=3, child_cmd=cmd, timeout=60) == 7  # ranks 0 and 2 are killed, not waited for
    assert time.time() - t < 30
    assert "rank 1 exited with code 7" in capfd.readouterr().err
    assert bench.launch_ranks(2, ["hang"], ndev=2, child_cmd=cmd, timeout=2) == 124
    # more ranks than devices: refused with a message, unless the ranks are told to share devices (rehearsals)
    monkeypatch.delenv("CFDP_SHARED_GPU", raising=False)
    assert bench.launch_ranks(2, ["ok"], ndev=1, child_cmd=cmd, timeout=60) == 2
    assert "2 but 1 GPU(s) are visible" in capfd.readouterr().err
    monkeypatch.setenv("CFDP_SHARED_GPU", "1")
    assert bench.launch_ranks(2, ["ok"], ndev=1, child_cmd=cmd, timeout=60) == 0


def test_bench_without_launcher_refuses_more_ranks_than_gpus():
    """started as the driver starts it (`python bench.py --gpus N`, no rank environment): never a silent 1-rank run"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "CFDP_SHARED_GPU")}
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this box has the devices")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode != 0 and "GPU(s) are visible" in r.stderr and not r.stdout.strip()


@pytest.mark.gpu
def test_bench_gpus2_started_plainly_launches_two_ranks(gpu):
    """`python bench.py --gpus 2 --steps 20 --warmup 3` with NO rank environment (how the driver starts it): the
    process launches its two ranks itself -- sharing this box's one GPU -- and rank 0's line says n_gpus 2, the
    exchange check holds, the overlap report and the CPU baseline are there"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "CFDP_SHARED_GPU")}
    import torch
    shared = torch.cuda.device_count() < 2
    if shared:
        env["CFDP_SHARED_GPU"] = "1"  # (on a box with >= 2 devices this is the real thing: one rank per GPU over xGMI)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "3",
                        "--no-weak", "--cpu-samples", "1"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["steps"] == 20 and out["warmup"] == 3
    assert out["exchange_check"]["ok"] and out["overlap"]["efficiency_async"] > 0
    assert out["config"]["baseline_config"] == "dualgrid.24" and out["config"]["ghost_points_per_gpu"] > 0
    cb = out["cpu_baseline"]
    assert cb["value"] > 0 and "whole 64^3 mesh of this config as one domain" in cb["sample"]
    # what the set-up validation of every attempted transport saw, and which check a rejected one failed
    val = out["config"]["transport_probe_validation"]
    assert any(k.startswith("ipc") and v["ok"] and v["failed"] is None and v["stale_reads"] == 0 and v["steps"] >= 186
               for k, v in val.items()), val
    assert out["exchange_check"]["stale_read_check"]["ok"], out["exchange_check"]
    if not shared:
        assert "rccl" in val, val


def test_eight_gpu_line_is_config_5_with_config_4_riding_along(pkg):
    """what `bench.py --gpus 8` measures (BASELINE.json configs 5 and 4), decided by pure functions of the rank count --
    and what a rehearsal by fewer ranks (CFDP_BENCH_AS_GPUS=8) takes over"""
    from cfd_proxy_amd import multigpu as mg
    assert mg.default_bench_config(8) == "dualgrid.384" and mg.bench_extra("dualgrid.384", 8) == ("strong_scaling", "dualgrid.192")
    c = mg.bench_config("dualgrid.384", 8)
    assert c["dims"] == (128, 128, 128) and c["ndomains"] == 384 and c["scaling"] == "weak" and "48 per GPU" in c["workload"]
    c = mg.bench_config("dualgrid.192", 8)
    assert c["dims"] == (64, 64, 64) and c["ndomains"] == 192 and "24 per GPU" in c["workload"]
    os.environ["CFDP_BENCH_AS_GPUS"] = "8"
    try:
        assert mg.bench_role(4) == 8 and mg.bench_role(6) == 8
    finally:
        del os.environ["CFDP_BENCH_AS_GPUS"]
    assert mg.bench_role(4) == 4
    for n in (4, 6, 8):  # the rank counts a rehearsal can use: whole domains per rank
        assert 384 % n == 0 and 192 % n == 0


def test_launcher_starts_eight_ranks(capfd):
    import bench
    cmd = [sys.executable, "-c", _FAKE_RANK]
    assert bench.launch_ranks(8, ["ok", "--gpus", "8"], ndev=8, child_cmd=cmd, timeout=60) == 0
    out = json.loads([l for l in capfd.readouterr().out.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 8 and out["argv"] == ["ok", "--gpus", "8"]


@pytest.mark.gpu
def test_bench_eight_gpu_code_path_rehearsed_by_four_ranks_on_a_tiny_mesh(gpu):
    """the code path of the driver's `bench.py --gpus 8 --steps 20 --warmup 5` -- dualgrid.384 as the workload,
    dualgrid.192 riding along as `strong_scaling`, the CPU baseline on the config's whole mesh, exchange_check with its
    scaled-field leg, overlap -- by FOUR self-launched ranks standing in for eight (a GPU box admits 6 GPU processes
    and this one already is one), on lattices halved per axis so that it takes about a minute"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(CFDP_SHARED_GPU="1", CFDP_BENCH_AS_GPUS="8", CFDP_BENCH_MESH_DIVISOR="2")
    import time
    t = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "20", "--warmup", "5",
                        "--cpu-samples", "1"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    wall = time.time() - t
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 4 and out["steps"] == 20 and out["warmup"] == 5 and "standing in for the 8-GPU line" in out["config"]["rehearsal"]
    assert out["config"]["baseline_config"] == "dualgrid.384" and out["scaling"] == "weak" and out["config"]["domains"] == 384
    assert out["config"]["domains_per_gpu"] == 96 and "TINY MESH" in out["config"]["workload"]
    assert out["exchange_check"]["ok"] and out["exchange_check"]["stale_read_check"]["stale_reads"] == 0
    assert out["overlap"]["efficiency_async"] > 0 and out["roofline"]["frac"] > 0
    st = out["strong_scaling"]
    assert st["config"]["baseline_config"] == "dualgrid.192" and st["scaling"] == "strong" and st["exchange_check"]["ok"], st
    assert st["exchange_check"]["stale_read_check"]["ok"] and st["overlap"]["efficiency_async"] > 0
    cb = out["cpu_baseline"]
    assert cb["value"] > 0 and "whole 64^3 mesh of this config as one domain" in cb["sample"]
    val = out["config"]["transport_probe_validation"]
    assert any(k.startswith("ipc") and v["ok"] for k, v in val.items()), val
    assert out["config"]["exchange_protocol"]["notify"] == "per partner", out["config"]["exchange_protocol"]
    assert out["config"]["exchange_protocol"]["notify_by"].startswith("counters"), out["config"]["exchange_protocol"]
    # the line says WHICH hardware it ran on: four ranks on however many devices this box has -- with one device a
    # rehearsal that can never be read as a scaling point
    c = out["config"]
    ndev = _visible_devices()
    assert len(c["device_of_rank"]) == 4 and all(isinstance(b, str) and b for b in c["device_of_rank"]), c["device_of_rank"]
    assert c["distinct_devices"] == len(set(c["device_of_rank"])) == min(ndev, 4) and c["ranks_per_device"] == -(-4 // min(ndev, 4))
    assert out["shared_gpu"] == (c["ranks_per_device"] > 1) and st["shared_gpu"] == out["shared_gpu"]
    assert ("shared_gpu_note" in out) == out["shared_gpu"]
    assert c["rccl_nranks"] is None and c["process_group"] == {"backend": "gloo", "world_size": 4}  # (CFDP_SHARED_GPU=1: gloo)
    assert c["device_of_rank"][0] in c["peer_access_of_rank0"] and c["env"]["CFDP_BENCH_AS_GPUS"] == "8"
    assert wall < 240, wall


# ------------------------------------------------------------------ experiment switches are locked out of the product
def test_experiment_switches_need_the_master_key(pkg):
    """host/experiments.c: a switch that can make the library compute something other than the product path (wrong
    values, ablated protocol, injected faults, test delays) is honoured only with CFDP_EXPERIMENTS=1, and says so"""
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from __graft_entry__ import load_package\n"
            "p = load_package(); v = p.host_lib().cfdp_experiment_getenv(b'CFDP_EXP_OWNED_NORMALS')\n"
            "print('VALUE', v, p.experiments_active())\n" % ROOT)
    env = {k: v for k, v in os.environ.items() if not k.startswith("CFDP_")}
    r = subprocess.run([sys.executable, "-c", code], env=dict(env, CFDP_EXP_OWNED_NORMALS="1"), capture_output=True, text=True)
    assert r.returncode == 0 and "VALUE None []" in r.stdout and "IGNORED" in r.stderr, r.stdout + r.stderr
    r = subprocess.run([sys.executable, "-c", code], env=dict(env, CFDP_EXP_OWNED_NORMALS="1", CFDP_EXPERIMENTS="1"),
                       capture_output=True, text=True)
    assert r.returncode == 0 and "VALUE b'1' ['CFDP_EXP_OWNED_NORMALS=1']" in r.stdout, r.stdout + r.stderr
    assert "EXPERIMENT SWITCH ACTIVE: CFDP_EXP_OWNED_NORMALS=1" in r.stderr, r.stderr
    for name in ("CFDP_EXP_OWNED_NORMALS", "CFDP_DEBUG_ABLATE", "CFDP_IPC_FAULT", "CFDP_PLAN_FAIL_STAGE", "CFDP_IPC_JITTER_US"):
        assert name in pkg.experiment_switches()
    # every getenv of a registered switch in the C / HIP sources goes through the gate
    import glob
    import re
    for f in glob.glob(os.path.join(ROOT, "cfd-proxy_amd", "csrc", "*")) + glob.glob(os.path.join(ROOT, "cfd-proxy_amd", "host", "*")):
        if f.endswith("experiments.c") or not f.endswith((".c", ".h", ".hip")):
            continue
        src = open(f).read()
        for name in pkg.experiment_switches():
            assert not re.search(r'[^_]getenv\("%s"\)' % name, src), (f, name)


@pytest.mark.parametrize("gpus", [1, 2])
def test_bench_refuses_a_line_under_an_experiment_switch(gpus):
    """no bench line can come from a run with an experiment switch active: non-zero exit, nothing on stdout -- decided
    before anything touches the GPU (runs here)"""
    env = {k: v for k, v in os.environ.items() if not k.startswith("CFDP_") and k != "WORLD_SIZE"}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "2"],
                       env=dict(env, CFDP_EXPERIMENTS="1", CFDP_DEBUG_ABLATE="256"), capture_output=True, text=True, timeout=120)
    assert r.returncode == 3 and not r.stdout.strip() and "refusing" in r.stderr and "CFDP_DEBUG_ABLATE=256" in r.stderr, r.stderr


@pytest.mark.gpu
def test_experiment_switch_without_the_master_key_changes_nothing(gpu):
    """CFDP_EXP_OWNED_NORMALS=1 (the cut-face-normals experiment: values WRONG) and CFDP_DEBUG_ABLATE without
    CFDP_EXPERIMENTS=1: the plan, the gradients and the flux are those of the product, bit for bit; with the key the
    experiment really is on (the values differ) -- so the first half of this test is not vacuous"""
    code = ("import sys, os, hashlib; sys.path.insert(0, %r)\n"
            "os.environ['CFDP_PLAN_DEVICE'] = '1'\n"
            "from __graft_entry__ import load_package\n"
            "p = load_package(); d = p.gen_domain(p.gen_params(20, 18, 16, ndomains=1), 0); p.fill_var(d, None, p.VAR_HASH)\n"
            "g = p.GpuPartition(d); g.set_fusion(True); g.run_iterations(3, True, 0, use_graph=False); g.pull_fields()\n"
            "print('HASH', hashlib.sha256(d.grad.tobytes() + d.psd_flux.tobytes()).hexdigest())\n" % ROOT)
    env = {k: v for k, v in os.environ.items() if not k.startswith("CFDP_")}
    def run(extra):
        r = subprocess.run([sys.executable, "-c", code], env=dict(env, **extra), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        return [l for l in r.stdout.splitlines() if l.startswith("HASH")][-1], r.stderr
    clean, _ = run({})
    locked, err = run({"CFDP_EXP_OWNED_NORMALS": "1", "CFDP_DEBUG_ABLATE": "64"})
    assert locked == clean and "IGNORED" in err, err
    live, err = run({"CFDP_EXP_OWNED_NORMALS": "1", "CFDP_EXPERIMENTS": "1"})
    assert live != clean and "EXPERIMENT SWITCH ACTIVE" in err, err


def test_power_reading_degrades_to_a_note_without_a_device():
    """roofline.power: where rocm-smi has no device to read (this container) the block is null-valued with the reason --
    never an exception, never a made-up number; with a fake rocm-smi on PATH it parses what the real one prints"""
    sys.path.insert(0, ROOT)
    import importlib
    bench = importlib.import_module("bench")
    calls = []
    pw = bench.power_beside(lambda: calls.append(1), seconds=0.05)
    assert calls and "socket_power_w" in pw and (pw["socket_power_w"] is None and pw.get("note") or pw["socket_power_w"] > 0), pw


def test_power_reading_parses_rocm_smi_output(tmp_path, monkeypatch):
    sys.path.insert(0, ROOT)
    import importlib
    bench = importlib.import_module("bench")
    fake = tmp_path / "rocm-smi"
    fake.write_text("""#!/bin/sh
case "$*" in
  *showmaxpower*) echo "GPU[0]		: Max Graphics Package Power (W): 1400.0";;
  *) echo "device,fclk clock speed:,fclk clock level:,sclk clock speed:,sclk clock level:,Current Socket Graphics Package Power (W)"
     echo "card0,(1250Mhz),0,(2029Mhz),1,1388.0";;
esac
""")
    fake.chmod(0o755)
    monkeypatch.setenv("PATH", f"{tmp_path}:{os.environ['PATH']}")
    pw = bench.power_beside(lambda: None, seconds=0.3)
    assert pw["socket_power_w"] == 1388.0 and pw["shader_clock_mhz"] == 2029 and pw["power_cap_w"] == 1400.0 and pw["samples"] >= 1, pw
