"""One rank of a multi-process halo-exchange test (launched by test_multirank.py).

CPU mode (gloo, no GPU): host logic only -- domain merge, request exchange over the process
group (the create_recvsend_index analogue), message order -- with the ORACLE standing in for the
device kernels (tests may use the oracle; the product never does).
GPU mode (--gpu): the real RankSolver with the "staged" transport, all ranks on cuda:0.
Each rank checks its owned AND ghost gradient rows against the un-partitioned mesh."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_oracle, load_package  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpu", action="store_true")
    ap.add_argument("--dims", default="16,12,10")
    ap.add_argument("--ndomains", type=int, default=12)
    ap.add_argument("--files", action="store_true")
    ap.add_argument("--transport", default="staged")
    ap.add_argument("--per-device", action="store_true",
                    help="one device per rank (LOCAL_RANK) and the nccl backend: the real multi-GPU set-up")
    ap.add_argument("--soak", type=int, default=0, help="extra iterations before the values are checked")
    ap.add_argument("--repeat-check", type=int, default=0,
                    help="development: only the scaled-field check of the fused schedule, this many times; the evidence of the "
                         "first failure is printed")
    ap.add_argument("--mode-may-be-rejected", action="store_true",
                    help="the memory mode named by CFDP_IPC_MODE may be REJECTED by the set-up validation (a question only "
                         "hardware answers): then print the evidence and exit with code 77 instead of failing")
    ap.add_argument("--inject-early-read", action="store_true",
                    help="CFDP_IPC_FAULT=skip_wait is set: the comparison of final states must still pass, the scaled-field "
                         "check must see the ghost rows that were read one exchange early")
    ap.add_argument("--irregular", action="store_true",
                    help="the generator's irregular option (random tetrahedralisation + hub points, scrambled numbering): tiles of "
                         "the large image, long incidence lists in chunks -- also in boundary tiles that push from registers")
    ap.add_argument("--notify-by-rank", default="",
                    help="comma-separated counter / flag, one per rank: neighbours that resolved to DIFFERENT forms of "
                         "notification (the per-partner protocol depends on a rank's own partition) must understand each other")
    ap.add_argument("--fail-first-validation", action="store_true",
                    help="the first exchange validation reports failure: the set-up must be torn down and retried")
    args = ap.parse_args()
    import torch
    import torch.distributed as dist

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    device = 0
    if args.per_device:
        device = int(os.environ.get("LOCAL_RANK", rank)) % torch.cuda.device_count()
        torch.cuda.set_device(device)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg, orc = load_package(), load_oracle()
    if args.notify_by_rank:  # (read when multigpu is imported: a form named in the environment is the only one tried)
        os.environ["CFDP_IPC_NOTIFY"] = args.notify_by_rank.split(",")[int(os.environ["RANK"])]
    from cfd_proxy_amd import multigpu as mg

    dims = tuple(int(x) for x in args.dims.split(","))
    mesh_kw = dict(connectivity=pkg.CONN_IRREGULAR, numbering=1) if args.irregular else {}
    gp = pkg.gen_params(*dims, ndomains=args.ndomains, **mesh_kw)
    part, st = mg.build_rank_partition(gp, args.ndomains, world, rank, via_files=args.files)
    mg.exchange_requests(part, rank, world, dist)

    # global truth from the un-partitioned mesh
    g1 = pkg.gen_params(*dims, ndomains=1, **({"connectivity": pkg.CONN_IRREGULAR} if args.irregular else {}))
    whole = pkg.gen_domain(g1, 0)
    pkg.fill_var(whole, None, pkg.VAR_HASH, *dims)
    truth = orc.np_gradients(whole.fpoint, whole.fnormal, whole.pvolume, whole.var, whole.nown)
    # lattice id of every merged point: var = 1 + 0.01*((7*gid + 13*eq) % 101) is not injective, so
    # rebuild gid from the generator instead
    first, count = pkg.rank_domains(rank, args.ndomains, world)
    gid = np.full(part.nall, -1, np.int64)
    mi = part.merge_info.contents
    for dl in range(count):
        dom = pkg.gen_domain(gp, first + dl)
        l2m = np.ctypeslib.as_array(mi.local2merged[dl], shape=(dom.nall,))
        gid[l2m] = pkg.gen_global_ids(gp, first + dl, dom.nall)
        dom.free()
    assert (gid >= 0).all()
    assert np.array_equal(part.var, whole.var[gid])

    if args.gpu:
        ftruth = orc.np_flux(whole.fpoint, whole.fnormal, truth, whole.nown, mode=0)
        if args.fail_first_validation:
            real, calls = mg.RankSolver.validate_exchange, []

            def flaky(self):
                calls.append(self._validating)
                return real(self) and len(calls) > 2
            mg.RankSolver.validate_exchange = flaky
            env0 = {k: v for k, v in os.environ.items() if k.startswith("CFDP_IPC")}
            solver = mg.RankSolver(part, rank, world, 0, dist, transport="ipc", tile_points=32)
            # the rungs in order: counters then flags on a fine-grained block, then the next memory mode
            assert solver.transport == "ipc" and calls == ["ipc / fine-grained block, notification by counters",
                                                           "ipc / fine-grained block, notification by flags",
                                                           "ipc / coarse-grained block, notification by counters"], (solver.transport, calls)
            m = solver.gpu.ipc_mode()
            assert m["memory"].startswith("coarse") and m["notify_by"].startswith("counters"), m
            # the attempts were configured by argument: the process environment is as the user left it
            assert env0 == {k: v for k, v in os.environ.items() if k.startswith("CFDP_IPC")}
            solver.run_steps(60, with_exchange=True, overlap=True)
            g = solver.grad_host()
            assert np.abs(g - truth[gid]).max() / np.abs(truth).max() <= 1e-12
            assert solver.gpu.ipc_error() == 0
            solver.close()
            mg.RankSolver.validate_exchange = real
            print(f"RANK_OK {rank}", flush=True)
            dist.destroy_process_group()
            return
        if args.inject_early_read:
            import time
            assert os.environ.get("CFDP_IPC_FAULT") == "skip_wait"
            real = mg.RankSolver.validate_exchange
            mg.RankSolver.validate_exchange = lambda self: True  # the set-up validation would (rightly) reject the faulty path
            solver = mg.RankSolver(part, rank, world, 0, dist, transport="ipc", tile_points=32)
            mg.RankSolver.validate_exchange = real
            assert solver.transport == "ipc"
            # (1) what rounds 1-3 checked: states after the run.  Blind: the field is constant in time
            solver.run_steps(60, with_exchange=True, overlap=True)
            old = solver.exchange_check()
            g = solver.grad_host()
            assert old["ok"], old
            assert np.abs(g - truth[gid]).max() / np.abs(truth).max() <= 1e-12
            assert np.abs(part.psd_flux[: part.nown] - ftruth[gid[: part.nown]]).max() / np.abs(ftruth[: whole.nown]).max() <= 1e-12
            # (2) the scaled field: rank 1 starts late, so every other rank's boundary tiles (which no longer wait)
            # read rows of earlier exchanges -- each such read is off by 2x or 4x and the device check counts it
            ev = solver.stale_read_check(batches=(40,), before_steps=(lambda: time.sleep(1.5)) if rank == 1 else None)
            if rank == 0:
                import json
                print("STALE_READ_EVIDENCE " + json.dumps(ev), flush=True)
            assert not ev["ok"] and ev["stale_reads"] > 0 and ev["first"] is not None, ev
            assert ev["wait_timeouts"] == 0, ev
            solver.close()
            print(f"RANK_OK {rank}", flush=True)
            dist.destroy_process_group()
            return
        if args.repeat_check:
            import json
            real = mg.RankSolver.validate_exchange
            mg.RankSolver.validate_exchange = lambda self: True
            solver = mg.RankSolver(part, rank, world, device, dist, transport=args.transport, tile_points=32, fusion=True)
            mg.RankSolver.validate_exchange = real
            bad = 0
            for i in range(args.repeat_check):
                ev = solver.stale_read_check(batches=(1, 2, 3, 5, 8, 57))
                if not ev["ok"]:
                    bad += 1
                    if rank == 0:
                        print(f"CHECK {i} FAILED " + json.dumps(ev), flush=True)
            if rank == 0:
                print(f"REPEAT_CHECK {args.repeat_check} runs, {bad} failed", flush=True)
            solver.close()
            print(f"RANK_OK {rank}", flush=True)
            dist.destroy_process_group()
            return
        for fusion in (False, True):
            solver = mg.RankSolver(part, rank, world, device, dist, transport=args.transport, tile_points=32, fusion=fusion)
            if rank == 0:  # what the set-up validation saw, and which check a rejected transport failed
                import json
                print("VALIDATION " + json.dumps(solver.validation), flush=True)
            if args.mode_may_be_rejected and solver.transport != args.transport:
                bad = {k: v for k, v in solver.validation.items() if not v.get("ok")}
                assert bad and all(v.get("failed") for v in bad.values()), solver.validation  # rejected BY the validation, with a reason
                print("MODE_REJECTED " + __import__("json").dumps(bad), flush=True)
                solver.close()
                dist.destroy_process_group()
                sys.exit(77)
            assert solver.transport == args.transport, solver.transport  # no silent fallback in the tests
            if args.notify_by_rank:  # this rank really runs the form it was given; its neighbours run another
                want = args.notify_by_rank.split(",")[rank]
                assert solver.gpu.ipc_mode()["notify_by"].startswith("counters" if want == "counter" else "flags"), solver.gpu.ipc_mode()
            if args.transport == "ipc" and not args.mode_may_be_rejected:
                # ... and no silent fall to a later RUNG either (round 5: a conservative rung behind the in-kernel ones
                # masked a broken in-kernel push in every multi-rank test): the first rung attempted is the one accepted
                first_rung, first_ev = next(iter(solver.validation.items()))
                assert first_ev.get("ok"), (first_rung, solver.validation)
            if args.soak:  # a long run in the scaled field: no flux phase of any step may have read a row of an earlier exchange
                ev = solver.stale_read_check(batches=(args.soak,))
                assert ev["ok"] and ev["stale_reads"] == 0, ev
            for overlap in (True, False):
                part.grad[:] = 1.0
                part.psd_flux[:] = 2.0
                solver.gpu.push_fields()
                for _ in range(3):  # repeated: buffer reuse hazards, both grad buffers of the fused mode
                    solver.step(with_exchange=True, overlap=overlap, with_flux=True)
                # the scaled field through this schedule (stream launches, and for ipc hipGraph replays): every step's flux
                ev = solver.stale_read_check(batches=(1, 2, 3, 5, 8, 57), overlap=overlap)
                assert ev["ok"] and ev["stale_reads"] == 0 and ev["flux_fields_compared_per_rank"] >= 75, (rank, fusion, overlap, ev)
                if args.transport == "ipc":  # batches: lead-in steps + hipGraph replays of 50 + remainder
                    solver.run_steps(107, with_exchange=True, overlap=overlap)
                    solver.gpu.refresh_graphs()  # the cached graph sets instantiated again (bench.py does, in front of a timed run)
                    solver.run_steps(52, with_exchange=True, overlap=overlap)
                    solver.run_steps(3, with_exchange=False, overlap=overlap)
                    assert solver.gpu.ipc_error() == 0
                    # EVERY schedule replays from hipGraphs (in-kernel wait, wait kernel, push / notify kernels of their own,
                    # un-fused steps): no capture abandoned, the chunks of 50 and the remainders really replayed
                    gs = solver.gpu.ipc_graph_stats()
                    assert gs["captures_failed"] == 0 and gs["steps_replayed"] >= 150, (rank, fusion, overlap, gs, solver.gpu.ipc_mode())
                    # a replayed graph has the landing arena baked in: capture one without exchange, flip the
                    # arena parity with ONE exchange step on doubled data, run without exchange again -- the
                    # flux must see the new ghost rows (everything is linear in var)
                    solver.run_steps(53, with_exchange=False, overlap=overlap)
                    part.var[:] *= 2.0
                    solver.gpu._ck(solver.gpu.lib.cfdp_gpu_set_var(solver.gpu.h, part.sd.var))
                    solver.run_steps(1, with_exchange=True, overlap=overlap)
                    solver.run_steps(53, with_exchange=False, overlap=overlap)
                    g2 = solver.grad_host().copy()
                    assert np.abs(g2 - 2.0 * truth[gid]).max() / np.abs(truth).max() <= 1e-12, (rank, fusion, overlap)
                    f2 = part.psd_flux[: part.nown]
                    assert np.abs(f2 - 2.0 * ftruth[gid[: part.nown]]).max() / np.abs(ftruth[: whole.nown]).max() <= 1e-12
                    part.var[:] *= 0.5
                    solver.gpu._ck(solver.gpu.lib.cfdp_gpu_set_var(solver.gpu.h, part.sd.var))
                    solver.run_steps(2, with_exchange=True, overlap=overlap)
                g = solver.grad_host().copy()
                err = np.abs(g - truth[gid]).max() / np.abs(truth).max()
                assert err <= 1e-12, (rank, fusion, overlap, err)
                f = part.psd_flux[: part.nown]
                ferr = np.abs(f - ftruth[gid[: part.nown]]).max() / np.abs(ftruth[: whole.nown]).max()
                assert ferr <= 1e-12, (rank, fusion, overlap, ferr)
            solver.close()
    else:
        ref = orc.CpuRef(part.fpoint, part.fnormal, part.pvolume, part.nown, nthreads=2, sendpoints=part.send_points())
        g = ref.gradients(part.var)
        ref.close()
        assert np.all(g[part.nown:] == 1.0)
        reqs, bufs = [], {}
        for s in part.partners:
            msg = torch.from_numpy(orc.pack(part.sendindex(s), g).ravel().copy())
            bufs[s] = torch.empty(len(part.recvindex(s)) * 21, dtype=torch.float64)
            reqs.append(dist.isend(msg, s))
            reqs.append(dist.irecv(bufs[s], s))
        for r in reqs:
            r.wait()
        for s in part.partners:
            orc.unpack(part.recvindex(s), g, bufs[s].numpy().reshape(-1, 21))
        err = np.abs(g - truth[gid]).max() / np.abs(truth).max()
        assert err <= 1e-12, (rank, err)
    dist.barrier()
    print(f"RANK_OK {rank} own={part.nown} ghost={part.nall - part.nown} partners={part.partners}")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
