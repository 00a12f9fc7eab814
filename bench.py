#!/usr/bin/env python3
"""bench.py -- Green-Gauss gradient iterations/s on the F6-like dualgrid stand-ins.

    python bench.py --gpus N --steps K --warmup W [--config NAME]

N > 1: either launched by torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment,
one rank per GPU), or -- started plainly, without a rank environment -- this process only LAUNCHES: before it
touches the GPU it starts N fresh rank processes of itself (the reference is started the same way, `mpirun -np N`
around init_communication, reference src/comm_data.c:257-307, README.txt:92-93), forwards rank 0's JSON line, and
exits non-zero if any rank fails or the job times out (CFDP_BENCH_TIMEOUT seconds, default 570).

One "step" = one iteration of the hot path as the reference harness times it
(reference src/solver.c:48-54): Green-Gauss gradients over all faces (+ halo exchange of the
168-byte gradient rows when N>1) + the pseudo-flux face loop.  Inputs are resident in HBM
before the timed region starts.

Workloads = the BASELINE.json configs (the real f6/dualgrid.N files are stripped from the reference
checkout, so deterministic stand-ins with the same schema are generated):

  --config        mesh          domains      default for   scaling
  dualgrid.12     64^3  lvl 2   12           --gpus 1      (anchor of the strong series)
  dualgrid.24     64^3  lvl 2   24           --gpus 2      strong (the same level-2 mesh, 12 domains per GPU)
  dualgrid.48     64^3  lvl 2   48           --gpus 4      strong (BASELINE config 3)
  dualgrid.192    64^3  lvl 2   192          (--gpus 8)    strong (BASELINE config 4: ~33 k points per GPU)
  dualgrid.384    128^3 finest  384          --gpus 8      weak vs --gpus 1: 262,144 points per GPU (config 5)
  weak            262,144 owned points per GPU at every N (128x64x64 / 128x128x64 for 2 / 4 GPUs)

`value` is the whole-job rate in units of one level-2 mesh: iterations/s of the mesh x (mesh points /
262,144) -- so the strong series (64^3 on 1, 2, 4 GPUs) and the weak point (128^3 on 8) read on one
scale, and value(N) / (N x value(1)) is the scaling efficiency either way.  With --gpus 2 / 4 the line
also carries the weak-scaling measurement of the same run under "weak_scaling"; with --gpus 8 the
dualgrid.192 strong-scaling point (BASELINE config 4) under "strong_scaling".  "cpu_baseline" (the compiled
reference on rank 0's host cores, the whole mesh of the config as one domain) is carried at every N.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)
UNIT_POINTS = 262144   # one level-2 mesh (64^3 stand-in of dualgrid.* lvl 2)
# seconds of the same iterations run (untimed) right in front of every timed region, see condition(); 0 = none
CONDITION_S = float(os.environ.get("CFDP_BENCH_CONDITION_S", "0.25"))
PRIMER = os.environ.get("CFDP_BENCH_PRIMER", "1") != "0"  # one untimed replay of the timed graph in front of the barrier


def usable_cores() -> int:
    """host cores this process may really use: the cgroup CPU quota if there is one (a GPU box
    shows every CPU of the host but grants a share), else the affinity mask; CFDP_CPU_THREADS
    overrides"""
    if os.environ.get("CFDP_CPU_THREADS"):
        return max(1, int(os.environ["CFDP_CPU_THREADS"]))
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except Exception:
            pass
    return min(n, 64)


def cpu_model() -> str:
    """the host CPU as /proc/cpuinfo names it (SURVEY 8d asks for it beside the core count)"""
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def kernel_source_tag() -> str:
    """identifies the kernel build a committed PMC measurement belongs to"""
    h = hashlib.sha256()
    for f in ("cfd-proxy_amd/csrc/gg_device.h", "cfd-proxy_amd/csrc/gg_kernels.hip", "cfd-proxy_amd/host/tiling.c"):
        h.update(open(os.path.join(ROOT, f), "rb").read())
    return h.hexdigest()[:16]


def power_beside(run_chunk, seconds: float = 1.6) -> dict:
    """socket power and shader clock (rocm-smi, a child process polled from a thread) while run_chunk() -- a few
    hundred back-to-back launches of one kernel, synchronised -- repeats for `seconds`: the face loops of this path
    run the socket at its power cap and the governor answers with the shader clock (tools/power_watch.py),
    so a roofline figure without the two is half the story.  None-valued when rocm-smi is not there or cannot be
    read as an ordinary user."""
    import re, statistics, subprocess, threading
    def smi(*flags):
        return subprocess.run(["rocm-smi", *flags], capture_output=True, text=True, timeout=15).stdout
    def sample():
        out = smi("--showclocks", "--showpower", "--csv").strip().splitlines()
        d = dict(zip(out[0].split(","), out[1].split(",")))
        return (int(re.sub(r"\D", "", d["sclk clock speed:"])), float([v for k, v in d.items() if "Power" in k][0]))
    try:
        cap = re.search(r"Max Graphics Package Power \(W\): *([0-9.]+)", smi("--showmaxpower"))
        seen, stop, errs = [], [], []
        def watch():
            while not stop:
                try:
                    seen.append(sample())
                except Exception as e:
                    errs.append(repr(e)[:80]); return
                time.sleep(0.05)
        th = threading.Thread(target=watch)
        t0 = time.time()
        run_chunk()  # (the power ramps over the first tenths of a second: what is kept starts behind them)
        th.start()
        while time.time() - t0 < seconds:
            run_chunk()
        stop.append(1)
        th.join()
        if not seen:
            return {"socket_power_w": None, "note": (errs or ["no sample"])[0]}
        return {"socket_power_w": statistics.median(p for _, p in seen), "socket_power_max_w": max(p for _, p in seen),
                "shader_clock_mhz": statistics.median(c for c, _ in seen), "power_cap_w": float(cap.group(1)) if cap else None,
                "samples": len(seen), "how": f"rocm-smi polled while the kernel ran back to back for {seconds} s"}
    except Exception as e:
        return {"socket_power_w": None, "note": repr(e)[:120]}


def committed_traffic(workload_key: str):
    """HBM bytes per launch from the PMC counters (FETCH_SIZE / WRITE_SIZE, separate rocprofv3 --pmc passes
    with the gfx950 corrections of MI355X_MICROARCH.md: tools/measure_traffic.py).  PMC passes cannot run
    inside this process, so the figures are read from the newest committed profiles/rNN_traffic.json and
    labelled with where they come from and whether the kernels have changed since."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))
    if not files:
        return {}, None
    path = files[-1]
    try:
        tr = json.load(open(path))
    except Exception:
        return {}, None
    src = {"file": os.path.relpath(path, ROOT), "method": "rocprofv3 --pmc FETCH_SIZE (x2, gfx950) / WRITE_SIZE, separate passes",
           "kernel_source_tag": tr.get("kernel_source_tag"),
           "stale": tr.get("kernel_source_tag") != kernel_source_tag()}
    out = {}
    for k, v in tr.get(workload_key, {}).items():
        for name in ("gg_gradient", "gg_flux", "gg_fused"):
            if name in k:
                out[name] = v["traffic_bytes"]
    return out, src


def launch_ranks(n: int, argv: list, ndev: int, child_cmd=None, timeout: float = None) -> int:
    """`python bench.py --gpus N` without a rank environment: start N fresh processes of this script, one rank
    each (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT), forward rank 0's stdout (the JSON line)
    and every rank's stderr, and return the exit code for the job: non-zero if any rank failed or the job ran
    past `timeout` seconds -- the remaining ranks are then killed (by pid).  The caller has not touched the GPU
    (no exec from a GPU process either: children are spawned).  `child_cmd` replaces [python, bench.py] in tests."""
    import socket
    import subprocess
    if n > ndev and os.environ.get("CFDP_SHARED_GPU") != "1":
        print(f"bench.py: --gpus {n} but {ndev} GPU(s) are visible; one rank per GPU is the contract "
              f"(CFDP_SHARED_GPU=1 lets ranks share devices: rehearsals only, the timings mean nothing)", file=sys.stderr)
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    timeout = timeout if timeout is not None else float(os.environ.get("CFDP_BENCH_TIMEOUT", "570"))
    cmd = list(child_cmd) if child_cmd else [sys.executable, os.path.abspath(__file__)]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r % max(ndev, 1)), WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen(cmd + list(argv), env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL,
                                      stderr=None, text=True))
    t_end = time.time() + timeout
    rc, out0 = 0, ""
    try:
        pending = set(range(n))
        while pending:
            for r in sorted(pending):
                try:
                    if r == 0:
                        out0, _ = procs[0].communicate(timeout=0.2)
                    else:
                        procs[r].wait(timeout=0.2)
                except subprocess.TimeoutExpired:
                    continue
                pending.discard(r)
                if procs[r].returncode != 0:
                    print(f"bench.py: rank {r} exited with code {procs[r].returncode}", file=sys.stderr)
                    rc = rc or (procs[r].returncode if procs[r].returncode > 0 else 1)
            if rc:
                break
            if time.time() > t_end:
                print(f"bench.py: the {n}-rank job ran past {timeout:.0f} s", file=sys.stderr)
                rc = 124
                break
    finally:
        for p in procs:  # only on failure / timeout is anything still running here
            if p.poll() is None:
                p.kill()
        for p in procs:
            try:
                p.wait(timeout=10)
            except Exception:
                pass
    for line in out0.splitlines():  # the ONE JSON line goes to stdout; anything else rank 0 printed (library chatter) to stderr
        (sys.stdout if line.startswith("{") else sys.stderr).write(line + "\n")
    sys.stdout.flush()
    if rc == 0 and not any(l.startswith("{") for l in out0.splitlines()):
        print("bench.py: rank 0 printed no JSON line", file=sys.stderr)
        rc = 1
    return rc


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20000)
    ap.add_argument("--warmup", type=int, default=2000)
    ap.add_argument("--config", default="", help="dualgrid.12 | dualgrid.24 | dualgrid.48 | dualgrid.192 | dualgrid.384 | weak "
                                                 "(default by --gpus: 1 -> .12, 2 -> .24, 4 -> .48, 8 -> .384)")
    ap.add_argument("--tile-points", type=int, default=0)
    ap.add_argument("--grad-lanes", type=int, default=0)
    ap.add_argument("--flux-lanes", type=int, default=0)
    ap.add_argument("--no-fusion", action="store_true",
                    help="one kernel per face loop instead of the fused flux(i)+gradients(i+1) pass")
    ap.add_argument("--no-files", action="store_true", help="generate domains in memory (skip the loader)")
    ap.add_argument("--no-finest", action="store_true", help="skip the finest-level single-GPU roofline run")
    ap.add_argument("--no-irregular", action="store_true",
                    help="skip the irregular-mesh roofline run (the same kernels on an unstructured graph of the bench workload's size)")
    ap.add_argument("--no-power", action="store_true", help="skip the rocm-smi power / clock samples of the roofline blocks")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline")
    ap.add_argument("--no-loopback", action="store_true",
                    help="skip the loopback measurement of the exchange protocol's own cost (N = 1 only)")
    ap.add_argument("--no-weak", action="store_true", help="skip the extra weak-scaling measurement at --gpus 2 / 4")
    ap.add_argument("--cpu-samples", type=int, default=25,
                    help="samples of 25 iterations each whose median is the CPU baseline: N_MEDIAN of the reference harness "
                         "(src/solver.c:32); meshes larger than one level-2 mesh take at most 3")
    ap.add_argument("--transport", default="auto", choices=["auto", "ipc", "rccl", "torch", "staged"],
                    help="auto: set up ipc and rccl, time both briefly, keep the faster; ipc: xGMI write + notify "
                         "between the processes of a node (falls back to rccl if its check fails); rccl: "
                         "ncclSend/ncclRecv issued by the C library; torch: torch.distributed P2P ops; staged: "
                         "through the host (tests)")
    ap.add_argument("--no-strong", action="store_true", help="skip the extra dualgrid.192 strong-scaling measurement at --gpus 8")
    args = ap.parse_args()

    from __graft_entry__ import load_package
    pkg = load_package()  # (the C host library only: nothing here touches the GPU)
    # a run under an experiment switch (wrong values, ablated protocol, injected faults, test delays: host/experiments.c)
    # is not a measurement of the product: no line
    active = pkg.experiments_active()
    if active:
        print(f"bench.py: refusing to report a run made with experiment switches active: {' '.join(active)} "
              f"(unset them or CFDP_EXPERIMENTS)", file=sys.stderr)
        raise SystemExit(3)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started plainly (no launcher): this process only launches the ranks, BEFORE anything touches the GPU
        # (counting devices does not initialise HIP on this image)
        import torch
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:], torch.cuda.device_count()))

    import numpy as np
    import torch

    from cfd_proxy_amd import multigpu as mg

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but the launcher set WORLD_SIZE={world}; running {world} rank(s)", file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    ndev = torch.cuda.device_count()
    device = local_rank % ndev
    torch.cuda.set_device(device)
    dist = None
    if world > 1:
        import torch.distributed as dist
        # RCCL needs one device per rank; ranks told to share a device (CFDP_SHARED_GPU=1: tests on a
        # 1-GPU box) rendezvous over gloo
        backend = "gloo" if args.transport == "staged" or os.environ.get("CFDP_SHARED_GPU") == "1" else "nccl"
        dist.init_process_group(backend=backend, rank=rank, world_size=world,
                                device_id=torch.device("cuda", device) if backend == "nccl" else None)
    coll_device = torch.device("cuda", device) if dist is not None and dist.get_backend() == "nccl" else "cpu"

    walls = {}  # seconds of wall time per phase of this process (rank 0's view), reported as "wall_s"
    t_start = time.time()

    def lap(key: str, since: float) -> float:
        now = time.time()
        walls[key] = round(walls.get(key, 0.0) + now - since, 2)
        return now

    def measure(cfg: dict, keep_solver: bool = False):
        """set-up, W untimed + exactly K timed steps (max over ranks), exchange check, overlap report"""
        dims, ndom = cfg["dims"], cfg["ndomains"]
        gp = pkg.gen_params(*dims, ndomains=ndom)
        tag = cfg["name"]
        t0 = time.time()
        part, _ = mg.build_rank_partition(gp, ndom, world, rank, via_files=not args.no_files)
        mg.exchange_requests(part, rank, world, dist)
        nfaces_part, nown, nadd = part.nfaces, part.nown, part.nall - part.nown
        tl = lap(f"{tag}: mesh files, loader, merge", t0)
        solver = mg.RankSolver(part, rank, world, device, dist, transport=args.transport,
                               tile_points=args.tile_points, grad_lanes=args.grad_lanes, flux_lanes=args.flux_lanes,
                               fusion=not args.no_fusion)
        tl = lap(f"{tag}: plan, upload, transport set-up + validation", tl)
        if world > 1 and args.transport == "auto":
            solver.choose_transport()
            tl = lap(f"{tag}: transport probe", tl)
        # can rank 0's device reach the devices it sees (by bus id)?  A launcher that shows every rank one device leaves
        # only the device itself here; the IPC mappings of the write + notify exchange do not depend on it
        peer_row = {}
        try:
            for j in range(ndev):
                peer_row[pkg.device_bus_id(j)] = True if j == device else bool(torch.cuda.can_device_access_peer(device, j))
        except Exception as e:
            peer_row = {"error": repr(e)[:120]}
        t_setup = time.time() - t0

        def barrier():
            solver.synchronize()
            if dist is not None:
                dist.barrier()
            solver.synchronize()

        cond = {"steps": 0}

        def condition(**kw) -> None:
            """CONDITION_S seconds of the same iterations right in front of a timed region (untimed, like the warm-up):
            the chip reaches the clock it sustains only after ~0.2 s of continuous load, and a 20-step region behind 5
            warm-up steps runs 8 % below it (tools/clock_ramp_probe.py: 39.8 us/step after 5 steps, 38.3 after 500,
            36.4 after 5000, 35.6 in a long run) -- a solver runs in the sustained state.  Every rank runs the same
            count (whole 50-pass graphs: the graph of the timed run stays instantiated)."""
            if CONDITION_S <= 0:
                return
            run = ((lambda n: solver.gpu.run_iterations(n + 1, with_flux=True, use_graph=True)) if world == 1
                   else (lambda n: solver.run_steps(n, **kw)))
            if not cond["steps"]:
                run(100)
                solver.synchronize()
                t = time.perf_counter()
                run(200)
                solver.synchronize()
                dt200 = time.perf_counter() - t
                if dist is not None:
                    tt = torch.tensor([dt200], dtype=torch.float64, device=coll_device)
                    dist.all_reduce(tt, op=dist.ReduceOp.MIN)
                    dt200 = float(tt.item())
                cond["steps"] = 50 * max(2, min(2000, int(np.ceil(CONDITION_S / max(dt200, 1e-6) * 4))))
            run(cond["steps"])

        def timed(steps: int, conditioned: bool = True, **kw) -> float:
            """seconds for exactly `steps` steps, max over ranks.  The hipGraphs the run replays are
            built before the timed region: captured without executing (one partition), or by one untimed
            rehearsal of the same schedule (several ranks: a capture there includes the exchange kernels)"""
            if world == 1:
                solver.gpu.prepare_iterations(steps, with_flux=True)
            else:
                solver.run_steps(steps if steps <= 5000 else 100 + steps % 50, **kw)
            if conditioned:
                condition(**kw)
                # ... and every graph instantiated again: an executable graph that other work has gone through the
                # device behind starts 60-100 us late on an idle device (tools/clock_ramp_probe.py)
                try:
                    solver.gpu.refresh_graphs()
                except Exception as e:  # (a graph that could not be instantiated again is captured anew by the run)
                    print(f"bench.py: {e}", file=sys.stderr)
                # ... and the K steps themselves once more, untimed, straight in front of the barrier: the FIRST replay of a
                # freshly instantiated graph pays the runtime's set-up for it (tools/timed_region_probe.py, 24 regions each,
                # twice: median 36.9-37.0 -> 36.2 us/step, quartiles 36.6-37.5 -> 36.1-36.3; EXPERIMENTS E.12)
                if PRIMER:
                    if world == 1:
                        solver.gpu.run_iterations(steps, with_flux=True, use_graph=True, device_time=False)
                    else:
                        solver.run_steps(steps, **kw)
                    cond["primer"] = steps
            barrier()
            t = time.perf_counter()
            if world == 1:  # (device_time=False: no HIP event pair around a run the host clock times -- 0.2 us/step, E.12)
                solver.gpu.run_iterations(steps, with_flux=True, use_graph=True, device_time=False)
            else:
                solver.run_steps(steps, **kw)
            solver.synchronize()   # device sync (+ torch.cuda.synchronize): every rank's K steps are done ...
            if dist is not None:
                dist.barrier()     # ... on every rank (nothing is in flight any more: no second sync behind it)
            dt = time.perf_counter() - t
            if dist is not None:
                tt = torch.tensor([dt], dtype=torch.float64, device=coll_device)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                dt = float(tt.item())
            return dt

        def exchange_check() -> dict:
            # after the timed region: every sent row in its slot, no wait gave up, and -- 50 more steps in the scaled
            # field -- no flux phase read a ghost row of an earlier exchange (a constant field would hide that)
            return solver.exchange_check(stale_read_steps=50)

        # warmup (untimed), then EXACTLY K timed steps; a transport whose rows did not all arrive is dropped
        # and the measurement repeated on the next one (every rank sees the same gathered check)
        rejected, chk = [], None
        while True:
            if world == 1:
                solver.gpu.run_iterations(max(args.warmup, 1), with_flux=True, use_graph=True)
            else:
                solver.run_steps(max(args.warmup, 1), with_exchange=True, overlap=True)
            dt_cold = timed(args.steps, conditioned=False, with_exchange=True, overlap=True)
            dt = timed(args.steps, with_exchange=True, overlap=True)
            if world == 1:
                break
            chk = exchange_check()
            if os.environ.get("CFDP_BENCH_REJECT_FIRST") == "1" and not rejected:
                chk["ok"] = False  # test hook: exercises the fall-back below
            if chk["ok"]:
                break
            rejected.append(solver.transport)
            if not solver.fallback():
                break
        tl = lap(f"{tag}: warm-up, timed steps, exchange check", tl)
        its = args.steps / dt  # iterations/s of the whole mesh
        mesh_points = dims[0] * dims[1] * dims[2]
        res = {
            "value": its * mesh_points / UNIT_POINTS, "ms_per_step": dt / args.steps * 1e3, "scaling": cfg["scaling"],
            # the same K steps timed the same way straight behind the W warm-up steps (chip below its sustained clock)
            "clock_conditioning": {"seconds": CONDITION_S, "steps": cond["steps"],
                                   "untimed_steps_in_front_of_the_timed_region": max(args.warmup, 1) + cond["steps"] + cond.get("primer", 0),
                                   "replay_of_the_timed_graph_in_front_of_the_barrier": cond.get("primer", 0),
                                   "unconditioned_ms_per_step": dt_cold / args.steps * 1e3,
                                   "unconditioned_value": args.steps / dt_cold * mesh_points / UNIT_POINTS},
            "config": {
                "workload": cfg["workload"], "baseline_config": cfg["name"],
                "mesh_points": mesh_points, "domains": ndom, "domains_per_gpu": ndom // world,
                "points_per_gpu": nown, "faces_per_gpu": nfaces_part, "ghost_points_per_gpu": nadd,
                "iteration": "gradients + halo exchange + pseudo flux",
                "fused_iterations": not args.no_fusion,
                "transport": solver.transport if world > 1 else "none (one partition)",
                "transport_probe_us_per_iteration": solver.probe if world > 1 else {},
                "transport_probe_rows_arrived": solver.checks if world > 1 else {},
                "transport_probe_validation": solver.validation if world > 1 else {},
                "exchange_protocol": solver.gpu.ipc_mode() if world > 1 and solver.transport == "ipc" else {},
                "mesh_iterations_per_s": its, "via_dualgrid_files": not args.no_files,
                "tiles": solver.gpu.stats["ntiles"], "tile_points": solver.gpu.stats["tile_points"],
                "setup_s": round(t_setup, 2),
                # WHICH hardware the ranks ran on (PCI bus ids gathered from every rank; what the RCCL communicator itself
                # counts): a rehearsal on a shared device can never be read as a scaling point
                "device_of_rank": solver.census["device_of_rank"], "distinct_devices": solver.census["distinct_devices"],
                "ranks_per_device": solver.census["ranks_per_device"],
                "rccl_nranks": (solver.gpu.rccl_nranks() or None) if world > 1 else None,
                "process_group": {"backend": dist.get_backend(), "world_size": dist.get_world_size()} if dist is not None else None,
                "peer_access_of_rank0": peer_row,
            },
            "shared_gpu": bool(solver.census["shared_gpu"]),
        }
        if world > 1:
            res["exchange_check"] = chk
            if rejected:
                res["exchange_check"]["transports_rejected"] = rejected
            # overlap efficiency (reference's own normalisation: comm_free / with exchange)
            dt_free = timed(args.steps, with_exchange=False)
            dt_bulk = timed(args.steps, with_exchange=True, overlap=False)
            res["overlap"] = {"t_comm_free_ms": dt_free / args.steps * 1e3, "t_async_ms": res["ms_per_step"],
                              "t_bulk_sync_ms": dt_bulk / args.steps * 1e3,
                              "efficiency_async": dt_free / dt, "efficiency_bulk_sync": dt_free / dt_bulk}
            tl = lap(f"{tag}: overlap report (comm_free, bulk)", tl)
        if keep_solver:
            return res, solver, part
        solver.close()
        part.free()
        return res, None, None

    # (CFDP_BENCH_AS_GPUS=N: this run stands in for the N-GPU line -- its config, ride-along and CPU baseline -- with
    # fewer ranks: the 8-rank command cannot be rehearsed on a box that admits 6 GPU processes)
    role = mg.bench_role(world)
    cfg = mg.bench_config(args.config or mg.default_bench_config(role), world)
    res, solver, part = measure(cfg, keep_solver=True)
    nfaces_part, nown, nadd = part.nfaces, part.nown, part.nall - part.nown

    out = {
        "metric": "green_gauss_gradient_iterations_per_sec",
        "value": res["value"],
        "unit": "iterations/s in units of one level-2 mesh (262144 points): mesh iterations/s x mesh points / 262144, whole job",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": res["ms_per_step"],
        "higher_is_better": True, "scaling": res["scaling"], "vs_baseline": None, "dtype": "f64",
        "data": "synthetic (F6-like dualgrid stand-in; the f6/dualgrid.N files are not distributed)",
        "config": res["config"],
    }
    # every CFDP_* variable this run saw (experiment switches cannot be among them: refused above)
    out["config"]["env"] = {k: v for k, v in sorted(os.environ.items()) if k.startswith("CFDP_")}
    if role != world:
        out["config"]["rehearsal"] = f"{world} ranks standing in for the {role}-GPU line (CFDP_BENCH_AS_GPUS)"
    out["shared_gpu"] = res["shared_gpu"]
    if res["shared_gpu"]:
        out["shared_gpu_note"] = (f"{world} ranks on {out['config']['distinct_devices']} device(s): a rehearsal of the code path; "
                                  f"rates and scaling mean nothing")
    for k in ("clock_conditioning", "exchange_check", "overlap"):
        if k in res:
            out[k] = res[k]
    # (the driver's record keeps `config`: what stood in front of the timed region, and the same K steps without it)
    for k in ("unconditioned_value", "unconditioned_ms_per_step", "untimed_steps_in_front_of_the_timed_region",
              "replay_of_the_timed_graph_in_front_of_the_barrier"):
        out["config"][k] = res["clock_conditioning"][k]

    # ---- roofline of the dominant kernel, HIP events on the stream the kernels run on ----
    # three fractions of the 8 TB/s peak side by side:
    #   frac            SURVEY 8d algorithmic bytes (B_grad + B_flux for the fused pass: the reference streams the
    #                   face data once per face loop) / launch time
    #   frac_unique     the bytes one launch must move at least (the fused pass streams the face data ONCE:
    #                   B_grad + B_flux - 32 F) / launch time
    #   frac_traffic    HBM-side bytes from the PMC counters / launch time
    t_phase = time.time()
    bg = pkg.algo_bytes_grad(nfaces_part, nown, nadd)
    bf = pkg.algo_bytes_flux(nfaces_part, nown, nadd)
    ms_g, ms_f = solver.gpu.time_kernels(500)
    traffic, traffic_src = ({}, None)
    if world == 1 and cfg["name"] == "dualgrid.12":
        traffic, traffic_src = committed_traffic("dualgrid.12 lvl 2 stand-in (64^3)")

    def fracs(alg, uniq, tr, ms):
        gbs = lambda b: b / (ms * 1e-3) / 1e9
        # traffic / traffic_committed: HBM bytes per launch from the PMC counters, read from the newest COMMITTED
        # profiles/rNN_traffic.json (PMC passes cannot run inside this process), never measured in this run --
        # traffic_source names the file and says whether the kernel sources have changed since
        return {"achieved": gbs(alg), "frac": gbs(alg) / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": alg,
                "unique_bytes_per_launch": uniq, "frac_unique": gbs(uniq) / HBM_PEAK_GBS,
                "traffic": tr, "traffic_committed": tr, "frac_traffic": gbs(tr) / HBM_PEAK_GBS if tr else None,
                "us_per_launch": ms * 1e3}
    grad_k = {"kernel": "gg_gradient_dma_kernel", **fracs(bg, bg, traffic.get("gg_gradient"), ms_g)}
    flux_k = {"kernel": "gg_flux_dma_kernel", **fracs(bf, bf, traffic.get("gg_flux"), ms_f)}
    if args.no_fusion:
        out["roofline"] = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", **grad_k, "flux_kernel": flux_k}
    else:
        # the timed loop runs the fused pass (flux(i) + gradients(i+1), tile data streamed once):
        # one launch does the work of one gradient launch and one flux launch
        ms_fu = solver.gpu.time_fused(1000)
        out["roofline"] = {"bound": "hbm", "kernel": "gg_fused_split_kernel", "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           **fracs(bg + bf, bg + bf - 32.0 * nfaces_part, traffic.get("gg_fused"), ms_fu),
                           "gradient_kernel": grad_k, "flux_kernel": flux_k}
        # the floor: the same pass with neither face loop (every load and every store; a diagnostic instantiation of
        # the kernel) -- what the pass takes beyond it is arithmetic and latency the resident tiles do not hide
        try:
            mv = solver.gpu.time_fused_movement(1000)
            out["roofline"]["movement_only_us"] = mv * 1e3
            out["roofline"]["exposed_beyond_movement_us"] = (ms_fu - mv) * 1e3
        except Exception as e:
            out["roofline"]["movement_only_us"] = None
            out["roofline"]["movement_only_note"] = str(e)[:160]
        if world == 1 and not args.no_power:
            out["roofline"]["power"] = power_beside(lambda: solver.gpu.time_fused(2000))
    out["roofline"]["traffic_source"] = traffic_src
    t_phase = lap("roofline block", t_phase)

    # ---- what rides along: --gpus 2 / 4 the weak-scaling point of the same run (262,144 owned points per GPU);
    # --gpus 8 BASELINE config 4 (dualgrid.192 lvl 2, ~33 k points per GPU: the strong-scaling point) ----
    extra = mg.bench_extra(cfg["name"], role)
    if extra and not (args.no_weak if extra[0] == "weak_scaling" else args.no_strong):
        if solver is not None:
            solver.close()
            part.free()
            solver = part = None
        try:
            xres, _, _ = measure(mg.bench_config(extra[1], world))
            out[extra[0]] = {k: xres[k] for k in ("value", "ms_per_step", "scaling", "shared_gpu", "config", "clock_conditioning", "exchange_check", "overlap")
                             if k in xres}
        except Exception as e:  # the extra must never cost the line
            out[extra[0]] = {"error": repr(e)[:300]}
    t_phase = time.time()

    if world > 1:
        # the other ranks are done: the CPU baseline below runs on rank 0's host cores alone
        if solver is not None:
            solver.close()
            part.free()
            solver = part = None
        dist.barrier()
        dist.destroy_process_group()
        dist = None
        if rank != 0:
            return

    if rank == 0 and world == 1:
        # ---- the truly HBM-bound single-GPU case: finest level (2.1 M points, 0.95 GB per pass) ----
        if not args.no_finest:
            gp1 = pkg.gen_params(128, ndomains=384)
            d1, _ = mg.build_rank_partition(gp1, 384, 1, 0, via_files=not args.no_files)
            p1 = pkg.GpuPartition(d1, device=device, tile_points=args.tile_points, grad_lanes=args.grad_lanes,
                                  flux_lanes=args.flux_lanes)
            p1.time_kernels(10)  # first touches of 1.7 GB of device memory
            g1, f1 = p1.time_kernels(50)
            fu1 = None
            if not args.no_fusion:
                p1.set_fusion(True)
                p1.time_fused(10)
                fu1 = p1.time_fused(50)
            b1 = pkg.algo_bytes_grad(d1.nfaces, d1.nown, 0)
            b1f = pkg.algo_bytes_flux(d1.nfaces, d1.nown, 0)
            tr1, _ = committed_traffic("dualgrid.384 finest-level stand-in (128^3)")
            fl = {"workload": "dualgrid.384 finest-level stand-in (128^3): 384 domain files -> loader -> merged on 1 GPU, no halo exchange",
                  "points": d1.nown, "faces": d1.nfaces,
                  "iterations_per_s": 1e3 / (fu1 if fu1 else g1 + f1),
                  "gradient_kernel": fracs(b1, b1, tr1.get("gg_gradient"), g1), "flux_kernel": fracs(b1f, b1f, tr1.get("gg_flux"), f1)}
            if fu1:
                fl["fused"] = fracs(b1 + b1f, b1 + b1f - 32.0 * d1.nfaces, tr1.get("gg_fused"), fu1)
                try:
                    mv1 = p1.time_fused_movement(50)
                    fl["fused"]["movement_only_us"] = mv1 * 1e3
                    fl["fused"]["exposed_beyond_movement_us"] = (fu1 - mv1) * 1e3
                except Exception as e:
                    fl["fused"]["movement_only_us"] = None
                if not args.no_power:
                    fl["fused"]["power"] = power_beside(lambda: p1.time_fused(300))
            out["finest_level"] = fl
            p1.close()
            d1.free()
        # ---- the same kernels on an IRREGULAR mesh of the bench workload's size: every figure above is on a lattice whose tiles
        # are all alike -- the best case.  The generator's irregular option (host/dualgrid_gen.c: the edge graph of a random
        # tetrahedralisation, 8 to 24 incidences per point, hubs of 60+, scrambled file numbering), 12 domain files -> loader
        # -> merged, as the bench workload goes ----
        if not args.no_irregular:
            try:
                gpi = pkg.gen_params(64, ndomains=12, connectivity=pkg.CONN_IRREGULAR, numbering=1)
                di, _ = mg.build_rank_partition(gpi, 12, 1, 0, via_files=not args.no_files)
                pi = pkg.GpuPartition(di, device=device, tile_points=args.tile_points, grad_lanes=args.grad_lanes, flux_lanes=args.flux_lanes)
                deg = np.bincount(di.fpoint.ravel(), minlength=di.nall)[: di.nown]
                pi.time_kernels(20)
                pkg.kernel_forms()
                gi, fi = pi.time_kernels(500)
                forms = pkg.kernel_forms()
                bi, bif = pkg.algo_bytes_grad(di.nfaces, di.nown, 0), pkg.algo_bytes_flux(di.nfaces, di.nown, 0)
                tri, tri_src = committed_traffic("irregular stand-in of the dualgrid.12 lvl 2 size (64^3 points)")
                st = pi.stats
                im = {"workload": "irregular stand-in of the dualgrid.12 lvl 2 size (64^3 points, random tetrahedralisation + hubs, scrambled "
                                  "numbering): 12 domain files -> loader -> merged on 1 GPU, no halo exchange",
                      "points": di.nown, "faces": di.nfaces, "faces_per_point": di.nfaces / di.nown,
                      "incidences_per_point": {"mean": float(deg.mean()), "min": int(deg.min()), "max": int(deg.max()),
                                               "points_above_30": int((deg > 30).sum())},
                      "tiles": st["ntiles"], "points_per_tile": di.nown / st["ntiles"], "halo_rows_per_tile": st["nhalo"] / st["ntiles"],
                      "face_duplication": st["nfaces_dup"] / st["nfaces_used"], "blob_bytes": st["blob_bytes"],
                      "launch_groups": [{"tiles": [b, e], "class": ("small", "large", "generic")[c]} for b, e, c in st["groups"]],
                      "gradient_kernel": fracs(bi, bi, tri.get("gg_gradient"), gi), "flux_kernel": fracs(bif, bif, tri.get("gg_flux"), fi),
                      "traffic_source": tri_src}
                if not args.no_fusion:
                    pi.set_fusion(True)
                    pi.time_fused(50)
                    pkg.kernel_forms()
                    fui = pi.time_fused(1000)
                    forms += " " + pkg.kernel_forms()
                    pkg.kernel_forms_off()
                    im["iterations_per_s"] = 1e3 / fui
                    im["fused"] = fracs(bi + bif, bi + bif - 32.0 * di.nfaces, tri.get("gg_fused"), fui)
                    try:
                        mvi = pi.time_fused_movement(500)
                        im["fused"]["movement_only_us"] = mvi * 1e3
                        im["fused"]["exposed_beyond_movement_us"] = (fui - mvi) * 1e3
                    except Exception as e:
                        im["fused"]["movement_only_us"] = None
                        im["fused"]["movement_only_note"] = str(e)[:160]
                    # the lattice is the best case: by how much (same box, same process, same byte count per unit)
                    im["lattice_over_irregular"] = {"fused_frac": out["roofline"]["frac"] / im["fused"]["frac"],
                                                    "gradient_kernel_frac": grad_k["frac"] / im["gradient_kernel"]["frac"]}
                im["kernel_forms"] = list(dict.fromkeys(forms.split()))  # (each form once, in the order it first ran)
                pi.close()
                di.free()
                # where part of the distance to the lattice goes: the same mesh WITHOUT its hub points (one point in 1024 with
                # 60-75 incidences: a lane walks its point's whole list, so a hub's wave -- and the tile that waits for it --
                # takes five times as long as its neighbours)
                if not args.no_fusion:
                    gph = pkg.gen_params(64, ndomains=12, connectivity=pkg.CONN_IRREGULAR, numbering=1, hubs=-1)
                    dh, _ = mg.build_rank_partition(gph, 12, 1, 0, via_files=not args.no_files)
                    ph = pkg.GpuPartition(dh, device=device, tile_points=args.tile_points, grad_lanes=args.grad_lanes, flux_lanes=args.flux_lanes)
                    gh, fh = ph.time_kernels(300)
                    ph.set_fusion(True)
                    ph.time_fused(50)
                    fuh = ph.time_fused(1000)
                    bh, bhf = pkg.algo_bytes_grad(dh.nfaces, dh.nown, 0), pkg.algo_bytes_flux(dh.nfaces, dh.nown, 0)
                    im["without_hub_points"] = {"faces": dh.nfaces, "incidences_max": int(np.bincount(dh.fpoint.ravel(), minlength=dh.nall)[: dh.nown].max()),
                                                "fused_us_per_launch": fuh * 1e3, "fused_frac": (bh + bhf) / (fuh * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                                "gradient_kernel_us_per_launch": gh * 1e3, "gradient_kernel_frac": bh / (gh * 1e-3) / 1e9 / HBM_PEAK_GBS}
                    ph.close()
                    dh.free()
                out["irregular_mesh"] = im
            except Exception as e:  # the extra must never cost the line
                out["irregular_mesh"] = {"error": repr(e)[:300]}
    if rank == 0 and world == 1 and not args.no_loopback and not args.no_fusion:
        # ---- what the xGMI write + notify protocol itself costs per iteration when no partner is ever late: rank 0's partition
        # of the 8-GPU configs on THIS GPU, every partner slot looped back to the rank's own landing arenas and flag words
        # (cfdp_gpu_ipc_connect_loopback: wrong ghost values, right traffic, right protocol).  comm_free / with_exchange is an
        # UPPER bound of the overlap efficiency a rank with a GPU of its own can reach (DESIGN appendix C.4)
        t_lb = lap("finest level", t_phase)
        try:
            lb = {}
            for name in ("dualgrid.384", "dualgrid.192"):
                c8 = mg.bench_config(name, 8)
                gp8 = pkg.gen_params(*c8["dims"], ndomains=c8["ndomains"])
                parts8 = [mg.build_rank_partition(gp8, c8["ndomains"], 8, r, via_files=False)[0] for r in range(8)]
                reqs = [{int(k): (v[0], v[1]) for k, v in pkg.merge_requests(p).items()} for p in parts8]
                mg.exchange_requests(parts8[0], 0, 8, None, all_requests=reqs)

                def lb_time(run, sync, steps, reps, **kw):
                    run(200 if steps >= 200 else 3 * steps, **kw)
                    sync()
                    best = float("inf")
                    for _ in range(reps):
                        t_ = time.perf_counter()
                        run(steps, **kw)
                        sync()
                        best = min(best, (time.perf_counter() - t_) / steps)
                    return best * 1e6

                def loopback(notify, push_inkernel=None):
                    """one row of the table: the memory mode the hosts try first, the given notification form"""
                    g8 = pkg.GpuPartition(parts8[0], device=device)
                    g8.set_fusion(True)
                    g8.ipc_configure(memory_mode=mg.ipc_mode_attempts()[0], notify=notify, push_inkernel=push_inkernel)
                    g8.ipc_export()
                    for s_ in range(len(g8.partners())):
                        g8._ck(g8.lib.cfdp_gpu_ipc_connect_loopback(g8.h, s_))
                    g8.ipc_ready()
                    run = lambda n, **kw: g8.run_steps_ipc(n, use_graph=2, **kw)
                    free = lb_time(run, g8.sync, 1000, 3, with_exchange=False, overlap=True)
                    exch = lb_time(run, g8.sync, 1000, 3, with_exchange=True, overlap=True)
                    free20 = lb_time(run, g8.sync, 20, 7, with_exchange=False, overlap=True)
                    exch20 = lb_time(run, g8.sync, 20, 7, with_exchange=True, overlap=True)
                    row = {"partition": f"rank 0 of 8: {parts8[0].nown} points, {len(g8.partners())} partners, "
                                        f"{g8.stats['nbtiles']} boundary tiles of {g8.stats['ntiles']}",
                           "us_per_iteration_comm_free": free, "us_per_iteration_with_exchange": exch,
                           "efficiency_bound": free / exch, "steps20_comm_free": free20, "steps20_with_exchange": exch20,
                           "steps20_ratio": free20 / exch20, "wait_timeouts": int(g8.ipc_error() != 0),
                           "graph_replay": g8.ipc_graph_stats(), "protocol": g8.ipc_mode()}
                    g8.ipc_disconnect()
                    g8.close()
                    return row
                lb[name] = loopback("counter")
                flag = loopback("flag")
                keys = ("us_per_iteration_with_exchange", "efficiency_bound", "steps20_with_exchange", "steps20_ratio", "wait_timeouts")
                lb[name]["flag_notification"] = {k: flag[k] for k in keys}
                # the conservative rung on the same mappings (the hosts try it after every in-kernel rung, before RCCL):
                # push, notify and wait as kernels of their own, flags, release / acquire at kernel boundaries
                sep = loopback("flag", push_inkernel=False)
                lb[name]["push_notify_wait_kernels"] = {k: sep[k] for k in keys}
                # ... and MPI_Put's pattern between them (src/exchange_data_mpidma.c:93-127): the send arena packed by a kernel,
                # one hipMemcpyAsync per partner slice into its landing slice, the notify kernel behind the copies
                try:
                    put = loopback("flag", push_inkernel="put")
                    lb[name]["copy_engine_put"] = dict({k: put[k] for k in keys}, protocol=put["protocol"], graph_replay=put["graph_replay"])
                except Exception as e:
                    lb[name]["copy_engine_put"] = {"error": repr(e)[:200]}
                # the fall-back branch priced on the same partition: grouped ncclSend / ncclRecv issued by the C library from
                # the streams (RCCL cannot be captured into a hipGraph in this ROCm: replay hangs), a communicator of ONE rank
                # exchanging with itself -- pack kernel + RCCL kernel + stream launches per iteration, no link crossed
                try:
                    g8 = pkg.GpuPartition(parts8[0], device=device)
                    g8.set_fusion(True)
                    rlib = mg.RankSolver.torch_rccl_path()
                    g8.rccl_init(pkg.GpuPartition.rccl_unique_id(rlib), 1, 0, rank_of_partner=[0] * len(g8.partners()), libpath=rlib, self_exchange=True)
                    run = lambda n, **kw: g8.run_steps_rccl(n, **kw)
                    rfree = lb_time(run, g8.sync, 500, 2, with_exchange=False, overlap=True)
                    rexch = lb_time(run, g8.sync, 500, 2, with_exchange=True, overlap=True)
                    rfree20 = lb_time(run, g8.sync, 20, 5, with_exchange=False, overlap=True)
                    rexch20 = lb_time(run, g8.sync, 20, 5, with_exchange=True, overlap=True)
                    lb[name]["rccl_self_sendrecv"] = {
                        "us_per_iteration_comm_free_stream_launched": rfree, "us_per_iteration_with_exchange": rexch,
                        "steps20_comm_free_stream_launched": rfree20, "steps20_with_exchange": rexch20,
                        "efficiency_bound_vs_graph_replayed_comm_free": lb[name]["us_per_iteration_comm_free"] / rexch,
                        "steps20_ratio_vs_graph_replayed_comm_free": lb[name]["steps20_comm_free"] / rexch20,
                        "rccl_nranks": g8.rccl_nranks(),
                        "note": "communicator of one rank, every partner mapped to itself; steps launched from the streams"}
                    g8.close()
                except Exception as e:
                    lb[name]["rccl_self_sendrecv"] = {"error": repr(e)[:200]}
                for p8 in parts8:
                    p8.free()
            lb["note"] = ("ONE GPU, every partner slot looped back to the rank's own arenas and flags: the protocol's own cost with a "
                          "partner that is never late; an upper bound of the overlap efficiency, not a measurement of it")
            out["exchange_protocol_loopback"] = lb
        except Exception as e:  # never costs the line
            out["exchange_protocol_loopback"] = {"error": repr(e)[:300]}
        t_phase = lap("exchange protocol in loopback", t_lb)
    # ---- CPU baseline on rank 0's host cores, at every N: the COMPILED REFERENCE (oracle/_ref/ref_dump_raw: the
    # reference's own OpenMP path, src/solver.c:42-58 comm_free loop + flux) when the binary is there,
    # with the oracle's port of the same algorithm class beside it ----
    if rank == 0 and not args.no_cpu:
        try:  # (a checker that cannot be loaded or timed must not cost the line)
            from __graft_entry__ import load_oracle
            orc = load_oracle()
            cores = usable_cores()
            dims = cfg["dims"]
            mesh_name = f"{dims[0]}x{dims[1]}x{dims[2]}" if len(set(dims)) > 1 else f"{dims[0]}^3"
            nsamp = args.cpu_samples if dims[0] * dims[1] * dims[2] <= UNIT_POINTS else min(args.cpu_samples, 3)
            cpu_dom = None
            if part is None:
                # N > 1: the whole mesh of this config as ONE domain, on rank 0's host cores (the other ranks are done)
                cpu_dom = pkg.gen_domain(pkg.gen_params(*dims, ndomains=1), 0)
                pkg.fill_var(cpu_dom, None, pkg.VAR_HASH)
                part = cpu_dom
                mesh_what = f"the whole {mesh_name} mesh of this config as one domain"
            else:
                mesh_what = f"same {mesh_name} merged mesh"
            in_units = dims[0] * dims[1] * dims[2] / UNIT_POINTS
            ref = orc.CpuRef(part.fpoint, part.fnormal, part.pvolume, part.nown, nthreads=cores)
            samples = sorted(ref.timed(part.var, niter=25, with_flux=True) for _ in range(nsamp))
            gsamples = sorted(ref.timed(part.var, niter=25, with_flux=False) for _ in range(3))
            ref.close()
            med = samples[len(samples) // 2]
            host = {"cpu_model": cpu_model(), "threads": cores,
                    # a GPU box grants a share of a bigger host: `cores` is what this process may run on, not the socket
                    "cores_note": f"cgroup / affinity share of a host that shows {os.cpu_count()} CPUs (not every physical core of it)",
                    "n_median": nsamp, "niter": 25,
                    "protocol": "N_MEDIAN samples of NITER iterations each, the median sample (src/solver.c:32,40,298; src/hybrid.f6.c:72)"}
            port = {"value": 25.0 / med, "unit": "iterations/s", "cores": cores, "kind": "port", **host,
                    "thread_binding": "none (OMP_PROC_BIND unset)",
                    "sample": f"{mesh_what}, {nsamp} samples x 25 iterations "
                              f"(gradients+flux), median; oracle/cpu_ref.c OpenMP, threads not bound",
                    "gradient_only_iterations_per_s": 25.0 / gsamples[len(gsamples) // 2],
                    "value_in_units_of_the_headline": 25.0 / med * in_units}
            out["cpu_baseline"] = port
            ref_bin = orc.ref_dump_path()
            if os.path.exists(ref_bin):
                import re
                import subprocess
                import tempfile
                try:
                    best = None
                    with tempfile.TemporaryDirectory() as tmp:
                        raw = os.path.join(tmp, "merged")
                        orc.write_raw_domain(raw, 0, part.fpoint, part.fnormal, part.pvolume, part.nown, var=part.var)
                        # the reference spin-waits between its threads: where they are pinned matters on a box
                        # that grants a share of a bigger host, so both placements are timed and the faster kept
                        def ref_rate(bind, samples, wf):
                            env = dict(os.environ, OMP_NUM_THREADS=str(cores), OMP_PROC_BIND=bind)
                            r = subprocess.run([ref_bin, "time", raw, str(samples), str(wf)],
                                               env=env, capture_output=True, text=True, timeout=300)
                            m = re.search(r"median_s=([0-9.]+)", r.stdout)
                            return 25.0 / float(m.group(1)) if r.returncode == 0 and m else None
                        quick = {b: ref_rate(b, min(nsamp, 5), 1) for b in ("false", "true")}  # which placement is faster here
                        quick = {b: v for b, v in quick.items() if v}
                        if quick:
                            bind = max(quick, key=quick.get)
                            v = ref_rate(bind, nsamp, 1)  # ... and THAT one by the reference's own protocol
                            if v:
                                best = {"value": v, "omp_proc_bind": bind,
                                        "gradient_only_iterations_per_s": ref_rate(bind, min(nsamp, 5), 0)}
                    if best:
                        out["cpu_baseline"] = {
                            "value": best["value"], "unit": "iterations/s", "cores": cores, "kind": "reference", **host,
                            "thread_binding": f"OMP_PROC_BIND={best['omp_proc_bind']}, OMP_NUM_THREADS={cores} (the faster of "
                                              f"false / true on this host, picked from 5 samples each)",
                            "sample": f"compiled reference (oracle/_ref/ref_dump_raw: src/solver.c:42-58 comm_free loop + "
                                      f"compute_psd_flux), {mesh_what}{'' if cpu_dom else ' as one domain'}, {nsamp} samples x 25 "
                                      f"iterations, median; OMP_PROC_BIND={best['omp_proc_bind']} (faster of false/true)",
                            "gradient_only_iterations_per_s": best.get("gradient_only_iterations_per_s"),
                            "value_in_units_of_the_headline": best["value"] * in_units,
                            "port": port}
                except Exception as e:  # the baseline is optional; the bench line must still print
                    out["cpu_baseline"]["reference_binary_error"] = str(e)[:200]
            if cpu_dom is not None:
                cpu_dom.free()
                part = None
        except Exception as e:
            out.setdefault("cpu_baseline", {})["error"] = repr(e)[:300]
    if rank == 0:
        lap("cpu baseline" if "finest level" in walls else "finest level + cpu baseline", t_phase)
        walls["total (this process, after imports)"] = round(time.time() - t_start, 2)
        out["wall_s"] = walls
        print(json.dumps(out))
    if solver is not None:
        solver.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
