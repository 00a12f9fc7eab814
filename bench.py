#!/usr/bin/env python3
"""bench.py -- Green-Gauss gradient iterations/s on the F6-like dualgrid stand-ins.

    python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run)

One "step" = one iteration of the hot path as the reference harness times it
(reference src/solver.c:48-54): Green-Gauss gradients over all faces (+ halo exchange of the
168-byte gradient rows when N>1) + the pseudo-flux face loop.  Inputs are resident in HBM
before the timed region starts.

Workload (BASELINE.json configs; the real f6/dualgrid.N files are stripped from the reference
checkout, so deterministic stand-ins with the same schema are generated):
  N=1: dualgrid.12 level-2 stand-in: 64^3 lattice, 12 domain files -> loader -> merged on one GPU
  N=8: dualgrid.384 finest-level stand-in: 128^3 lattice, 384 domains, 48 per GPU
  N=2,4: the same 262,144 owned points per GPU (128x64x64 / 128x128x64, 12 domains per GPU)
so per-GPU work is fixed ("scaling": "weak") and `value` counts 262,144-point partition
iterations per second summed over GPUs (= iterations/s of the whole mesh x N).

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)


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


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20000)
    ap.add_argument("--warmup", type=int, default=2000)
    ap.add_argument("--tile-points", type=int, default=0)
    ap.add_argument("--grad-lanes", type=int, default=0)
    ap.add_argument("--flux-lanes", type=int, default=0)
    ap.add_argument("--no-fusion", action="store_true",
                    help="one kernel per face loop instead of the fused flux(i)+gradients(i+1) pass")
    ap.add_argument("--no-files", action="store_true", help="generate domains in memory (skip the loader)")
    ap.add_argument("--no-finest", action="store_true", help="skip the finest-level single-GPU roofline run")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline")
    ap.add_argument("--cpu-samples", type=int, default=7)
    ap.add_argument("--transport", default="auto", choices=["auto", "ipc", "rccl", "torch", "staged"],
                    help="auto: set up ipc and rccl, time both briefly, keep the faster; ipc: xGMI write + notify "
                         "between the processes of a node (falls back to rccl if its check fails); rccl: "
                         "ncclSend/ncclRecv issued by the C library; torch: torch.distributed P2P ops; staged: "
                         "through the host (tests)")
    args = ap.parse_args()

    import numpy as np
    import torch

    from __graft_entry__ import load_package
    pkg = load_package()
    from cfd_proxy_amd import multigpu as mg

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
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

    dims, ndom = mg.bench_mesh(world)
    gp = pkg.gen_params(*dims, ndomains=ndom)
    t0 = time.time()
    part, st = mg.build_rank_partition(gp, ndom, world, rank, via_files=not args.no_files)
    mg.exchange_requests(part, rank, world, dist)
    nfaces_part, nown, nadd = part.nfaces, part.nown, part.nall - part.nown
    solver = mg.RankSolver(part, rank, world, device, dist, transport=args.transport,
                           tile_points=args.tile_points, grad_lanes=args.grad_lanes, flux_lanes=args.flux_lanes,
                           fusion=not args.no_fusion)
    if world > 1 and args.transport == "auto":
        solver.choose_transport()
    t_setup = time.time() - t0
    coll_device = solver.device if dist is not None and dist.get_backend() == "nccl" else "cpu"

    def barrier():
        solver.synchronize()
        if dist is not None:
            dist.barrier()
        solver.synchronize()

    def timed(steps: int, **kw) -> float:
        """seconds for exactly `steps` steps, max over ranks"""
        if world > 1:
            solver.run_steps(56, **kw)  # untimed: the hipGraph of this schedule is captured here, not in the timed region
        barrier()
        t = time.perf_counter()
        if world == 1:
            solver.gpu.run_iterations(steps, with_flux=True, use_graph=True)  # hipGraph replay of 25-step chunks
        else:
            solver.run_steps(steps, **kw)
        solver.synchronize()
        barrier()
        dt = time.perf_counter() - t
        if dist is not None:
            tt = torch.tensor([dt], dtype=torch.float64, device=coll_device)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt

    def exchange_check() -> dict:
        """every row that was sent must have arrived: sum over ranks of the ghost rows == sum over ranks
        of the packed send rows (same doubles, so equal up to the order of the additions)"""
        solver.synchronize()
        g = solver.grad_host()
        sidx = [part.sendindex(k) for k in part.partners]  # a point sent to two partners counts twice
        sent = float(np.abs(g[np.concatenate(sidx)]).sum()) if sidx else 0.0
        got = float(np.abs(g[part.nown:]).sum())
        tt = torch.tensor([sent, got], dtype=torch.float64, device=coll_device)
        dist.all_reduce(tt)
        chk = {"sum_abs_sent_rows": float(tt[0]), "sum_abs_ghost_rows": float(tt[1]),
               "ok": bool(abs(float(tt[0]) - float(tt[1])) <= 1e-9 * max(float(tt[0]), 1e-300))}
        if solver.transport == "ipc":
            et = torch.tensor([float(solver.gpu.ipc_error() != 0)], dtype=torch.float64, device=coll_device)
            dist.all_reduce(et)
            chk["wait_timeouts"] = int(et.item())
            chk["ok"] = chk["ok"] and int(et.item()) == 0
        return chk

    # warmup (untimed), then EXACTLY K timed steps; a transport whose rows did not all arrive is dropped
    # and the measurement repeated on the next one (every rank sees the same all-reduced check)
    rejected, chk = [], None
    while True:
        if world == 1:
            solver.gpu.run_iterations(max(args.warmup, 1), with_flux=True, use_graph=True)
        else:
            solver.run_steps(max(args.warmup, 1), with_exchange=True, overlap=True)
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
    ms_per_step = dt / args.steps * 1e3
    its = args.steps / dt  # iterations/s of the whole mesh

    out = {
        "metric": "green_gauss_gradient_iterations_per_sec",
        "value": its * world,
        "unit": "iterations/s of one 262144-point (dualgrid.12 lvl-2 sized) partition, summed over GPUs",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic (F6-like dualgrid stand-in; the f6/dualgrid.N files are not distributed)",
        "config": {
            "workload": {1: "dualgrid.12 lvl 2 stand-in (64^3, 12 domains merged on 1 GPU, no halo exchange)",
                         8: "dualgrid.384 finest-level stand-in (128^3, 384 domains, 48 per GPU, halo exchange over xGMI)"}.get(
                world, f"{dims[0]}x{dims[1]}x{dims[2]} lattice, {ndom} domains, {ndom // world} per GPU, halo exchange over xGMI"),
            "mesh_points": dims[0] * dims[1] * dims[2], "points_per_gpu": nown, "faces_per_gpu": nfaces_part,
            "ghost_points_per_gpu": nadd, "iteration": "gradients + halo exchange + pseudo flux",
            "fused_iterations": not args.no_fusion,
            "transport": solver.transport if world > 1 else "none (one partition)",
            "transport_probe_us_per_iteration": solver.probe if world > 1 else {},
            "mesh_iterations_per_s": its, "via_dualgrid_files": not args.no_files,
            "tiles": solver.gpu.stats["ntiles"], "tile_points": solver.gpu.stats["tile_points"],
            "setup_s": round(t_setup, 2),
        },
    }

    if world > 1:
        out["exchange_check"] = chk
        if rejected:
            out["exchange_check"]["transports_rejected"] = rejected

    # ---- overlap efficiency (reference's own normalisation: comm_free / with exchange) ----
    if world > 1:
        dt_free = timed(args.steps, with_exchange=False)
        dt_bulk = timed(args.steps, with_exchange=True, overlap=False)
        out["overlap"] = {"t_comm_free_ms": dt_free / args.steps * 1e3, "t_async_ms": ms_per_step,
                          "t_bulk_sync_ms": dt_bulk / args.steps * 1e3,
                          "efficiency_async": dt_free / dt, "efficiency_bulk_sync": dt_free / dt_bulk}

    # ---- roofline of the dominant kernel (gradient face loop), HIP events on its own stream ----
    bg = pkg.algo_bytes_grad(nfaces_part, nown, nadd)
    bf = pkg.algo_bytes_flux(nfaces_part, nown, nadd)
    ms_g, ms_f = solver.gpu.time_kernels(500)
    # HBM bytes per launch from the PMC counters (FETCH_SIZE / WRITE_SIZE, separate rocprofv3 passes,
    # gfx950 corrections: tools/measure_traffic.py) of the same workload, committed under profiles/
    traffic = None
    if world == 1:
        try:
            tr = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))
            for k, v in tr["dualgrid.12 lvl 2 stand-in (64^3)"].items():
                if "gg_gradient" in k:
                    traffic = v["traffic_bytes"]
        except Exception:
            traffic = None
    grad_k = {"kernel": "gg_gradient_dma_kernel", "achieved": bg / (ms_g * 1e-3) / 1e9,
              "frac": bg / (ms_g * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": traffic,
              "algorithmic_bytes_per_launch": bg, "us_per_launch": ms_g * 1e3}
    flux_k = {"kernel": "gg_flux_dma_kernel", "achieved": bf / (ms_f * 1e-3) / 1e9, "us_per_launch": ms_f * 1e3,
              "algorithmic_bytes_per_launch": bf}
    if args.no_fusion:
        out["roofline"] = {"bound": "hbm", "kernel": grad_k["kernel"], "achieved": grad_k["achieved"],
                           "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": grad_k["frac"], "traffic": traffic,
                           "algorithmic_bytes_per_launch": bg, "us_per_launch": ms_g * 1e3, "flux_kernel": flux_k}
    else:
        # the timed loop runs the fused pass (flux(i) + gradients(i+1), tile data streamed once):
        # one launch does the work of one gradient launch and one flux launch
        ms_fu = solver.gpu.time_fused(1000)
        ftraffic = None
        if world == 1:
            try:
                for k, v in tr["dualgrid.12 lvl 2 stand-in (64^3)"].items():
                    if "gg_fused" in k:
                        ftraffic = v["traffic_bytes"]
            except Exception:
                ftraffic = None
        out["roofline"] = {"bound": "hbm", "kernel": "gg_fused_split_kernel", "achieved": (bg + bf) / (ms_fu * 1e-3) / 1e9,
                           "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": (bg + bf) / (ms_fu * 1e-3) / 1e9 / HBM_PEAK_GBS,
                           "traffic": ftraffic, "algorithmic_bytes_per_launch": bg + bf, "us_per_launch": ms_fu * 1e3,
                           "gradient_kernel": grad_k, "flux_kernel": flux_k}

    if rank == 0 and world == 1:
        # ---- the truly HBM-bound single-GPU case: finest level (2.1 M points, 0.95 GB per pass) ----
        if not args.no_finest:
            gp1 = pkg.gen_params(128, ndomains=1)
            d1 = pkg.gen_domain(gp1, 0)
            pkg.fill_var(d1, None, pkg.VAR_HASH)
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
            out["finest_level"] = {"workload": "dualgrid.384 finest-level stand-in merged on 1 GPU (128^3)",
                                   "points": d1.nown, "faces": d1.nfaces, "algorithmic_bytes_per_launch": b1,
                                   "us_per_launch": g1 * 1e3, "achieved": b1 / (g1 * 1e-3) / 1e9,
                                   "frac": b1 / (g1 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                   "iterations_per_s": 1e3 / (fu1 if fu1 else g1 + f1), "flux_us_per_launch": f1 * 1e3,
                                   "flux_achieved": b1f / (f1 * 1e-3) / 1e9}
            if fu1:
                out["finest_level"]["fused"] = {"us_per_launch": fu1 * 1e3, "algorithmic_bytes_per_launch": b1 + b1f,
                                                "achieved": (b1 + b1f) / (fu1 * 1e-3) / 1e9,
                                                "frac": (b1 + b1f) / (fu1 * 1e-3) / 1e9 / HBM_PEAK_GBS}
            p1.close()
            d1.free()
        # ---- CPU baseline: the oracle (a port of the reference's algorithm class) on the host cores ----
        if not args.no_cpu:
            from __graft_entry__ import load_oracle
            orc = load_oracle()
            cores = usable_cores()
            ref = orc.CpuRef(part.fpoint, part.fnormal, part.pvolume, part.nown, nthreads=cores)
            samples = sorted(ref.timed(part.var, niter=25, with_flux=True) for _ in range(args.cpu_samples))
            gsamples = sorted(ref.timed(part.var, niter=25, with_flux=False) for _ in range(3))
            ref.close()
            med = samples[len(samples) // 2]
            out["cpu_baseline"] = {"value": 25.0 / med, "unit": "iterations/s", "cores": cores, "kind": "port",
                                   "sample": f"same 64^3 merged mesh, {args.cpu_samples} samples x 25 iterations "
                                             f"(gradients+flux), median; oracle/cpu_ref.c OpenMP",
                                   "gradient_only_iterations_per_s": 25.0 / gsamples[len(gsamples) // 2]}
            # the compiled reference itself (oracle/_ref/ref_dump, built in the build container from the
            # reference's own sources, see oracle/Makefile), timed on the same mesh written as ONE
            # dualgrid file; reported next to the port so that the two can be compared
            ref_bin = os.path.join(ROOT, "oracle", "_ref", "ref_dump")
            if os.path.exists(ref_bin):
                import re
                import subprocess
                import tempfile
                try:
                    with tempfile.TemporaryDirectory() as tmp:
                        part.write(os.path.join(tmp, "merged_domain_0_lvl_2"))
                        env = dict(os.environ, OMP_NUM_THREADS=str(cores), OMP_PROC_BIND="true")
                        r = subprocess.run([ref_bin, "time", os.path.join(tmp, "merged"), "2", str(args.cpu_samples), "1"],
                                           env=env, capture_output=True, text=True, timeout=300)
                    m = re.search(r"median_s=([0-9.]+)", r.stdout)
                    if r.returncode == 0 and m:
                        out["cpu_baseline"]["reference_binary"] = {
                            "value": 25.0 / float(m.group(1)), "unit": "iterations/s", "cores": cores, "kind": "reference",
                            "sample": f"compiled reference (comm_free, gradients+flux), {args.cpu_samples} samples x 25 "
                                      f"iterations, median, same mesh as one dualgrid file"}
                except Exception as e:  # the baseline is optional; the bench line must still print
                    out["cpu_baseline"]["reference_binary_error"] = str(e)[:200]
    if rank == 0:
        print(json.dumps(out))
    solver.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
