"""N>1 path: world_size-2/3 process groups (gloo).  CPU: host logic with the oracle as the
device stand-in.  GPU: the real RankSolver (staged transport), ranks sharing cuda:0."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _launch(world, extra, extra_env=None, rejected_ok=False):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="2", **(extra_env or {}))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_rank_worker.py")] + extra,
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    if rejected_ok and all(p.returncode == 77 for p in procs):  # every rank alike: the validation rejected the mode
        ev = [l for l in outs[0].splitlines() if l.startswith("MODE_REJECTED ")]
        return ev[-1] if ev else "MODE_REJECTED"
    if any(p.returncode != 0 or f"RANK_OK {r}" not in out for r, (p, out) in enumerate(zip(procs, outs))):
        try:  # keep every rank's whole output where a GPU box's results are collected
            import time
            d = os.path.join(ROOT, "gpurun_out")
            os.makedirs(d, exist_ok=True)
            with open(os.path.join(d, f"multirank_failure_{int(time.time())}.log"), "w") as fh:
                fh.write(f"world {world} extra {extra} env {extra_env}\n")
                for r, (p, out) in enumerate(zip(procs, outs)):
                    fh.write(f"\n===== rank {r} rc {p.returncode}\n{out}\n")
        except OSError:
            pass
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"RANK_OK {r}" in out, out[-3000:]
    return None


@pytest.mark.parametrize("world,extra", [(2, []), (3, ["--files"]), (2, ["--dims", "12,12,12", "--ndomains", "8"])])
def test_multirank_host_logic_gloo(pkg, orc, world, extra):
    _launch(world, extra)


@pytest.mark.gpu
def test_multirank_ranksolver_staged_on_one_gpu(gpu):
    _launch(2, ["--gpu"])
    _launch(3, ["--gpu", "--files"])


@pytest.mark.gpu
def test_multirank_xgmi_write_notify_between_processes_on_one_gpu(gpu):
    """the "ipc" transport for real: 2, 3 and 4 processes sharing this GPU map each other's landing
    arenas through HIP IPC handles, push rows and flags into them from kernels, poll flags on the
    device, replay steps from hipGraphs; owned AND ghost gradient rows and the flux are checked
    against the un-partitioned mesh"""
    # (ranks that share a device use the wait kernel by default -- waiting boundary tiles of several ranks can fill the
    # device; these meshes are small, and the wait INSIDE the fused pass is the path a rank with its own GPU takes)
    inkernel = {"CFDP_IPC_WAIT_INKERNEL": "1"}
    _launch(2, ["--gpu", "--transport", "ipc", "--soak", "600"], extra_env=inkernel)  # + 600 steps in the scaled field
    _launch(3, ["--gpu", "--transport", "ipc", "--files"], extra_env=inkernel)
    _launch(4, ["--gpu", "--transport", "ipc", "--dims", "16,16,12", "--ndomains", "8"], extra_env=inkernel)


@pytest.mark.gpu
def test_multirank_xgmi_write_notify_under_random_skew(gpu):
    """the same checks while every rank idles for a pseudo-random time (up to 40 us: several passes of these meshes) in
    front of every step -- drawn on the device, so every hipGraph replay draws new delays (CFDP_IPC_JITTER_US): the ranks
    drift against each other step by step, and a hole in the wait / notify / double-buffer protocol that lockstep runs never
    hit would show up in the scaled-field soak (400 steps) and in the value checks of every schedule"""
    env = {"CFDP_IPC_WAIT_INKERNEL": "1", "CFDP_IPC_JITTER_US": "40", "CFDP_EXPERIMENTS": "1"}
    _launch(3, ["--gpu", "--transport", "ipc", "--files", "--soak", "400"], extra_env=env)
    _launch(4, ["--gpu", "--transport", "ipc", "--dims", "16,16,12", "--ndomains", "8", "--soak", "400"], extra_env=env)


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{"CFDP_IPC_MODE": "split"}, {"CFDP_IPC_MODE": "fine"}, {"CFDP_IPC_PER_PARTNER": "0"},
                                 {"CFDP_IPC_WAIT_INKERNEL": "0"}, {"CFDP_IPC_INKERNEL": "0"}, {"CFDP_IPC_NOTIFY": "flag"},
                                 {"CFDP_IPC_NOTIFY": "flag", "CFDP_IPC_WAIT_INKERNEL": "0"},
                                 {"CFDP_IPC_NOTIFY": "counter", "CFDP_IPC_INKERNEL": "0"},
                                 {"CFDP_IPC_NOTIFY": "flag", "CFDP_IPC_INKERNEL": "2"}, {"CFDP_IPC_INKERNEL": "2", "CFDP_IPC_MODE": "coarse"}])
def test_multirank_xgmi_write_notify_other_rungs(gpu, env):
    """the same value checks (and the scaled-field check of every schedule) on the other rungs of the exchange: flags in a
    fine-grained block of their own with the arenas coarse-grained and an explicit invalidate ("split"), everything
    fine-grained, one completion counter for all partners instead of one per partner, the wait as a kernel of its own,
    push / notify as kernels of their own, notification by flags instead of counters (with either wait), counters raised
    by the notify kernel, and the copy-engine put (pack kernel + one copy per partner slice into its landing slice + the
    notify kernel: MPI_Put's pattern, src/exchange_data_mpidma.c:93-127)"""
    _launch(3, ["--gpu", "--transport", "ipc", "--files"], extra_env=dict({"CFDP_IPC_WAIT_INKERNEL": "1"}, **env))


@pytest.mark.gpu
def test_multirank_xgmi_write_notify_on_an_irregular_mesh(gpu):
    """the same flow on the generator's irregular mesh (random tetrahedralisation, hub points of 50-75 faces, scrambled
    numbering): tiles of the large image, incidence lists cut into chunks for helper lane groups -- also in BOUNDARY tiles,
    whose send rows leave from the registers of the point's own lanes after the helpers' sums have joined them -- values of
    every rank against the un-partitioned mesh and the scaled-field check of every schedule, between processes over real IPC
    mappings"""
    _launch(3, ["--gpu", "--transport", "ipc", "--irregular", "--dims", "20,18,16", "--ndomains", "6", "--files", "--soak", "200"],
            extra_env={"CFDP_IPC_WAIT_INKERNEL": "1"})
    _launch(2, ["--gpu", "--transport", "staged", "--irregular", "--dims", "20,18,16", "--ndomains", "4"])


@pytest.mark.gpu
@pytest.mark.parametrize("forms", ["counter,flag,counter", "flag,counter,flag,flag"])
def test_neighbours_with_different_notification_forms_understand_each_other(gpu, forms):
    """every rank decides counters or flags on its own (the per-partner protocol depends on its partition) and the two
    forms store different things in a partner's word: tiles x exchanges, or the exchange number.  What a waiting rank
    compares the word with is therefore the SENDER's statement (word NEED_IN of the slot line: what the word advances by
    per exchange) -- mixed neighbourhoods pass every value check and the scaled-field check of every schedule, under skew.
    (Round 5 only the set-up validation stood between a mixed neighbourhood and a silent stale read -- advisor finding.)"""
    world = forms.count(",") + 1
    extra = ["--gpu", "--transport", "ipc", "--notify-by-rank", forms, "--soak", "300"] + (["--dims", "16,16,12", "--ndomains", "8"] if world == 4 else ["--files"])
    _launch(world, extra, extra_env={"CFDP_IPC_WAIT_INKERNEL": "1", "CFDP_IPC_JITTER_US": "30", "CFDP_EXPERIMENTS": "1"})


@pytest.mark.gpu
@pytest.mark.parametrize("world,mode", [(2, "coarse"), (2, "fine"), (2, "split"), (3, "coarse"), (4, "fine")])
def test_scaled_field_check_sees_a_ghost_row_read_one_exchange_early(gpu, world, mode):
    """fault injection: CFDP_IPC_FAULT=skip_wait makes the boundary tiles of the fused pass read their ghost rows without
    waiting for the partners' rows of the previous exchange (the wait at the top of the pass is skipped).  The checks of
    rounds 1-3 -- sums of sent vs received rows after the run, final gradients and flux against the oracle -- PASS on
    that broken path, because the field is constant in time and the arena still holds the row of two exchanges ago.
    The scaled-field check (var x 2, 2, 1/4 per iteration, the flux of every step compared on the device) must FAIL
    it.  Reference analogue: the stage / flag lock-step asserts at every receive, src/exchange_data_mpi.c:189,439."""
    extra = ["--gpu", "--inject-early-read"] + (["--dims", "16,16,12", "--ndomains", "8"] if world == 4 else [])
    _launch(world, extra, extra_env={"CFDP_IPC_FAULT": "skip_wait", "CFDP_EXPERIMENTS": "1", "CFDP_IPC_MODE": mode,
                                    "CFDP_IPC_WAIT_INKERNEL": "1"})


@pytest.mark.gpu
def test_write_notify_setup_is_retried_with_a_fine_grained_block(gpu):
    """a failed exchange validation tears the IPC mappings down on every rank and the set-up is retried on the next rung
    -- per memory mode of the landing block (fine -> coarse -> split, CFDP_IPC_MODE) the notification by counters, then by
    flags (CFDP_IPC_NOTIFY) -- before any other transport is tried; the rungs are configured by argument, the process
    environment stays as the user left it"""
    _launch(2, ["--gpu", "--fail-first-validation"])


def _device_count():
    try:
        import torch
        return torch.cuda.device_count()
    except Exception:
        return 0


two_devices = pytest.mark.skipif(_device_count() < 2, reason="needs two GPUs (a 1-GPU box shares cuda:0 between the ranks instead)")


@pytest.mark.gpu
@two_devices
@pytest.mark.parametrize("transport", ["ipc", "rccl"])
def test_transports_between_two_devices_against_the_oracle(gpu, transport):
    """the first thing to run on a multi-GPU node (also: tools/multigpu_selftest.py): one rank per DEVICE.  "ipc": the
    xGMI write + notify exchange in the first memory mode of its landing block (fine -> coarse -> split) that passes the
    scaled-field validation -- one MUST; "rccl": the C library's RCCL send/recv group (cfdp_gpu_step_rccl).  1000 steps
    in the scaled field (no flux phase may read a ghost row of an earlier exchange), then owned AND ghost rows and the
    flux of every rank against the un-partitioned mesh"""
    env = {k: v for k, v in os.environ.items() if k not in ("CFDP_IPC_MODE", "CFDP_IPC_FINEGRAINED")}
    _launch(2, ["--gpu", "--per-device", "--transport", transport, "--soak", "1000"],
            extra_env={"CFDP_IPC_MODE": ""} if "CFDP_IPC_MODE" in env else None)


@pytest.mark.gpu
@two_devices
@pytest.mark.parametrize("mode", ["coarse", "split", "fine"])
def test_each_memory_mode_of_the_landing_block_between_two_devices(gpu, mode):
    """each memory mode on its own between two devices: either it carries the 1000 scaled steps and the value checks, or
    the set-up validation REJECTS it with a reason (skip, with the evidence: whether a coarse-grained arena behind
    another device's stores is coherent inside a kernel is a question only hardware answers) -- never a wrong flux"""
    ev = _launch(2, ["--gpu", "--per-device", "--transport", "ipc", "--soak", "1000", "--mode-may-be-rejected"],
                 extra_env={"CFDP_IPC_MODE": mode}, rejected_ok=True)
    if ev:
        pytest.skip(f"memory mode {mode} rejected by the scaled-field validation on this node: {ev[:400]}")


MPIEXEC = "/opt/conda/bin/mpiexec"
MPI_DRIVER = os.path.join(ROOT, "cfd-proxy_amd", "bin", "hybrid.f6.hip.mpi")


@pytest.mark.skipif(not (os.path.exists(MPIEXEC) and os.path.exists(MPI_DRIVER)), reason="no MPI in this image")
@pytest.mark.parametrize("nranks,extra", [(2, []), (3, []), (4, ["--cluster"])])
def test_mpi_launched_driver_control_plane(pkg, tmp_path, nranks, extra):
    """mpiexec -n G hybrid.f6.hip.mpi --dry-run: rank/size, domain -> rank map, merge, and the MPI
    exchange of the (domain, idx) request lists (the reference's index exchange,
    src/comm_data.c:203-249) -- everything before the GPU is touched"""
    prefix = str(tmp_path / "dualgrid")
    pkg.write_mesh(pkg.gen_params(16, 14, 12, ndomains=8), prefix, 2)
    r = subprocess.run([MPIEXEC, "-n", str(nranks), MPI_DRIVER, "-lvl", "2", prefix, "--dry-run"] + extra,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "*** SUCCESS (dry run)" in r.stdout and "halo tables consistent" in r.stdout
    assert r.stdout.count(f"/{nranks}: ") == nranks


@pytest.mark.gpu
@pytest.mark.skipif(not (os.path.exists(MPIEXEC) and os.path.exists(MPI_DRIVER)), reason="no MPI in this image")
def test_mpi_launched_driver_one_rank_on_gpu(gpu, tmp_path):
    """the MPI-launched main() end to end with one rank (RCCL refuses two ranks on one device, so
    the exchange between processes cannot run on a 1-GPU box)"""
    prefix = str(tmp_path / "dualgrid")
    gpu.write_mesh(gpu.gen_params(16, 14, 12, ndomains=4), prefix, 2)
    r = subprocess.run([MPIEXEC, "-n", "1", MPI_DRIVER, "-lvl", "2", prefix, "--var", "hash"], capture_output=True,
                       text=True, timeout=300)
    if r.returncode == 127:
        pytest.skip("MPI runtime libraries not resolvable on this box")
    assert r.returncode == 0, r.stdout + r.stderr
    assert "*** SUCCESS" in r.stdout and "comm_free:" in r.stdout


@pytest.mark.gpu
@pytest.mark.skipif(not (os.path.exists(MPIEXEC) and os.path.exists(MPI_DRIVER)), reason="no MPI in this image")
@pytest.mark.parametrize("nranks", [2, 3])
def test_mpi_launched_driver_xgmi_write_notify_on_one_gpu(gpu, tmp_path, nranks):
    """mpiexec -n G hybrid.f6.hip.mpi with the ranks sharing this GPU: MPI control plane, HIP IPC
    handles exchanged with MPI_Allgather, the reference's entry points driving push / notify / wait
    kernels; the driver itself checks that every sent row arrived"""
    prefix = str(tmp_path / "dualgrid")
    gpu.write_mesh(gpu.gen_params(20, 16, 12, ndomains=6), prefix, 2)
    r = subprocess.run([MPIEXEC, "-n", str(nranks), MPI_DRIVER, "-lvl", "2", prefix, "--var", "hash"],
                       capture_output=True, text=True, timeout=400)
    if r.returncode == 127:
        pytest.skip("MPI runtime libraries not resolvable on this box")
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "exchange: xGMI write + notify" in r.stdout and "*** SUCCESS" in r.stdout
    assert "exchange_dbl_gaspi_async:" in r.stdout and "exchange check:" in r.stdout and " ok" in r.stdout
    assert "validated" in r.stdout  # the data path was checked against the owners' rows before it was timed


def _read_rank_dump(path):
    raw = np.fromfile(path, np.uint8)
    nown, nall = np.frombuffer(raw[:8], np.int32)
    pairs = np.frombuffer(raw[8:8 + 8 * nall], np.int32).reshape(nall, 2)
    off = 8 + 8 * nall
    grad = np.frombuffer(raw[off:off + nall * 168], np.float64).reshape(nall, 7, 3)
    flux = np.frombuffer(raw[off + nall * 168:off + nall * 192], np.float64).reshape(nall, 3)
    return int(nown), int(nall), pairs, grad, flux


@pytest.mark.gpu
@pytest.mark.skipif(not (os.path.exists(MPIEXEC) and os.path.exists(MPI_DRIVER)), reason="no MPI in this image")
@pytest.mark.parametrize("nranks", [2, 3])
def test_mpi_launched_driver_values_match_the_whole_mesh_oracle(gpu, orc, tmp_path, nranks):
    """VALUES of the MPI-launched run (ranks sharing this GPU, exchange over HIP IPC): every rank's own rows,
    the ghost rows its partners delivered, and its flux against the C oracle on the un-partitioned mesh"""
    from conftest import TOL, rel_err
    pkg = gpu
    dims, nd = (20, 16, 12), 6
    gp = pkg.gen_params(*dims, ndomains=nd)
    prefix = str(tmp_path / "dualgrid")
    pkg.write_mesh(gp, prefix, 2)
    out = str(tmp_path / "dump")
    r = subprocess.run([MPIEXEC, "-n", str(nranks), MPI_DRIVER, "-lvl", "2", prefix, "--var", "volume", "--dump", out],
                       capture_output=True, text=True, timeout=400)
    if r.returncode == 127:
        pytest.skip("MPI runtime libraries not resolvable on this box")
    assert r.returncode == 0 and "*** SUCCESS" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
    whole = pkg.gen_domain(pkg.gen_params(*dims, ndomains=1), 0)  # local id == global id
    var = 1.0 + 0.01 * np.arange(1, 8)[None, :] * np.fmod(whole.pvolume * 1e9, 97.0)[:, None]
    ref = orc.CpuRef(whole.fpoint, whole.fnormal, whole.pvolume, whole.nown, nthreads=2)
    g_ref = ref.gradients(var)
    f_ref = ref.flux(g_ref, mode=0)
    ref.close()
    scale = np.maximum(np.abs(g_ref), orc.np_scale(whole.fpoint, whole.fnormal, whole.pvolume, var))
    gids = {d: pkg.gen_global_ids(gp, d, pkg.load_domain(prefix, d, 2).nall) for d in range(nd)}
    seen = 0
    for rk in range(nranks):
        nown, nall, pairs, grad, flux = _read_rank_dump(f"{out}_rank_{rk}.bin")
        gid = np.array([gids[d][i] for d, i in pairs])
        assert nall > nown > 0
        err = np.abs(grad - g_ref[gid]) / scale[gid]          # own rows AND delivered ghost rows
        assert err.max() <= TOL, (rk, err.max())
        assert np.abs(flux[:nown] - f_ref[gid[:nown]]).max() <= TOL * np.abs(f_ref).max(), rk
        seen += nown
    assert seen == whole.nown
    whole.free()


REF_MPI_MAIN = os.path.join(ROOT, "oracle", "_ref", "hybrid.f6.mpi.dropin")


@pytest.mark.gpu
@pytest.mark.skipif(not (os.path.exists(MPIEXEC) and os.path.exists(REF_MPI_MAIN)),
                    reason="oracle/_ref/hybrid.f6.mpi.dropin is built where /root/reference and an MPI exist")
@pytest.mark.parametrize("nranks", [1, 2])
def test_reference_main_and_harness_unchanged_under_mpiexec(gpu, tmp_path, nranks):
    """the reference's own hybrid.f6.c AND solver.c (its ten-variant harness: every OpenMP thread calls the
    entry points, MPI_Barrier around the samples), compiled unchanged against include/compat/ and linked with
    libcfdproxy_mpi.so + libcfdproxy_hip.so: one MPI rank per domain file, as the reference runs"""
    pkg = gpu
    pkg.write_mesh(pkg.gen_params(20, 16, 12, ndomains=nranks), str(tmp_path / "dualgrid"), 2)
    r = subprocess.run([MPIEXEC, "-n", str(nranks), REF_MPI_MAIN, "-lvl", "2", "dualgrid"], cwd=str(tmp_path),
                       env=dict(os.environ, OMP_NUM_THREADS="3"), capture_output=True, text=True, timeout=600)
    if r.returncode == 127:
        pytest.skip("MPI runtime libraries not resolvable on this box")
    assert r.returncode == 0 and "*** SUCCESS" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
    assert "NTHREADS: 3" in r.stdout and f"nProc: {nranks}" in r.stdout
    assert "exchange_dbl_mpi_pscw_async:" in r.stdout and "comm_free:" in r.stdout
    if nranks > 1:
        assert "validated" in r.stdout


MPI_LIB = os.path.join(ROOT, "cfd-proxy_amd", "lib", "libcfdproxy_mpi.so")


@pytest.mark.skipif(not (os.path.exists(MPIEXEC) and os.path.exists(MPI_LIB)), reason="no MPI in this image")
@pytest.mark.parametrize("nranks", [2, 3, 4])
def test_mpi_hooks_index_exchange_on_cpu(pkg, tmp_path, nranks):
    """world_size > 1 on the CPU, under mpiexec: a C host linked with libcfdproxy_mpi.so + libcfdproxy_host.so calls
    init_communication (MPI_Init_thread through the hook), the loader, compute_communication_tables (the sendindex
    exchange of src/comm_data.c:203-249), free_communication_ressources; every rank's send list must name, by global
    lattice id and in message order, exactly the ghosts its partner expects from it"""
    lib = os.path.join(ROOT, "cfd-proxy_amd", "lib")
    exe = str(tmp_path / "host_mpi_tables")
    r = subprocess.run(["gcc", "-std=gnu99", "-O1", "-Wall", "-Werror", "-fopenmp", "-Wno-stringop-overflow", "-I/opt/conda/include",
                        os.path.join(ROOT, "tests", "host_mpi_tables.c"), "-I" + os.path.join(ROOT, "include"), "-L" + lib,
                        "-Wl,--no-as-needed", "-lcfdproxy_mpi", "-lcfdproxy_hip", os.path.join(lib, "mpi", "libmpi.so.12"),
                        "-Wl,-rpath," + lib, "-Wl,-rpath," + os.path.join(lib, "mpi"), "-Wl,--allow-shlib-undefined", "-o", exe],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    dims = (12, 10, 8)
    pkg.write_mesh(pkg.gen_params(*dims, ndomains=nranks), str(tmp_path / "dualgrid"), 2)
    r = subprocess.run([MPIEXEC, "-n", str(nranks), exe, "dualgrid", "2"] + [str(d) for d in dims], cwd=str(tmp_path),
                       capture_output=True, text=True, timeout=120)
    if r.returncode == 127:
        pytest.skip("MPI runtime libraries not resolvable on this box")
    assert r.returncode == 0 and "*** SUCCESS" in r.stdout and "mismatches 0" in r.stdout, r.stdout + r.stderr
