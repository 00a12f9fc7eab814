"""One process per GPU: the multi-rank host of the hot path (torch.distributed = plumbing).

The reference runs one MPI rank per dualgrid domain and exchanges ghost gradients with
MPI/GASPI point-to-point messages (reference src/exchange_data_mpi.c:96-166,199-543,
src/exchange_data_gaspi.c:105-151).  Here each GPU rank owns N/G whole domains
(host/domain_merge.c) and exchanges 168-byte gradient rows with its neighbour ranks:

  * transport "ipc": xGMI write + notify -- the analogue of the reference's best variant,
    gaspi_write_notify / notify_waitsome (src/exchange_data_gaspi.c:105-151,190-305).  Every rank
    exports a block [flag words | landing arena 0 | landing arena 1] through a HIP IPC handle; the
    packing kernel writes a partner's rows straight into that partner's arena and a second kernel
    raises the iteration counter in the partner's flag word; the receiver polls its own flags on
    the device.  No communication library and no host work beyond kernel launches in the
    iteration, so runs of iterations are replayed from a hipGraph.  One node only.  The setup
    checks an exchange against the owners' values (twice, with var rescaled in between, so that a
    stale copy of a ghost row could not pass) and falls back to "rccl" if it does not hold.
  * transport "rccl": grouped ncclSend/ncclRecv over xGMI issued by the C library itself
    (cfdp_gpu_step_rccl / cfdp_gpu_run_steps_rccl: one host call per iteration or per batch of
    iterations), on a communicator it creates from a ncclUniqueId broadcast over the process
    group; RCCL is the library PyTorch already loaded.  The send side is the packed send arena,
    the receive side is the ghost block of `grad` itself (ghost rows are numbered in message
    order), so there is no unpack pass and no staging copy.  Used across nodes, and as the
    fallback of "ipc".
  * transport "torch": the same messages as torch.distributed batch_isend_irecv ("nccl" backend)
    on the context's comm stream -- the fallback when the library cannot set up its own
    communicator (several Python-level calls per iteration).
  * transport "staged": device -> host -> gloo -> host -> device.  Only for tests on
    machines without one GPU per rank (several ranks may share cuda:0).

Overlap (reference: "trigger the communication as early as possible", README.txt:118-130):
tiles holding sent points run first; pack + exchange go to a second stream while the
interior tiles run on the main stream; the pseudo-flux kernel waits for both.

Setup traffic mirrors create_recvsend_index (reference src/comm_data.c:203-249): every rank
tells each partner which of the partner's points it needs, as (domain, idx) pairs.
"""
from __future__ import annotations

import os
import shutil
import tempfile
from typing import Dict, List, Optional, Tuple

import numpy as np

from . import (FLUX_CONSISTENT, TILES_ALL, TILES_BOUNDARY, TILES_INTERIOR, VAR_HASH, Domain,
               GpuPartition, fill_var, gen_domain, gen_global_ids, load_domain, merge_domains,
               merge_requests, merge_set_send, rank_domain_list)

ROWLEN = 21  # NGRAD * 3 doubles per halo point (reference dim2, src/gradients.c:176-177)


def bench_mesh(n_ranks: int) -> Tuple[Tuple[int, int, int], int]:
    """Stand-in meshes for the BASELINE.json configs, 262,144 owned points per GPU:
    1 GPU = dualgrid.12 level 2 (64^3), 8 GPUs = dualgrid.384 finest level (128^3)."""
    table = {1: ((64, 64, 64), 12), 2: ((128, 64, 64), 24), 4: ((128, 128, 64), 48),
             8: ((128, 128, 128), 384)}
    if n_ranks in table:
        return table[n_ranks]
    return (64 * n_ranks, 64, 64), 12 * n_ranks


# the BASELINE.json workloads (stand-ins: the F6 files are not distributed).  dualgrid.12/.24/.48/.192 are the
# SAME level-2 mesh cut into more domains -- the reference's strong-scaling regime, where a rank's share
# shrinks to ~1 k points per core (reference README.txt:39-40) -- dualgrid.384 is the finest level.
_BENCH = {
    "dualgrid.12": ((64, 64, 64), 12, "lvl 2", "strong"),
    "dualgrid.24": ((64, 64, 64), 24, "lvl 2", "strong"),
    "dualgrid.48": ((64, 64, 64), 48, "lvl 2", "strong"),
    "dualgrid.192": ((64, 64, 64), 192, "lvl 2", "strong"),
    "dualgrid.384": ((128, 128, 128), 384, "finest level", "weak"),
}


def bench_role(n_ranks: int) -> int:
    """the GPU count whose BASELINE workload a run takes: n_ranks itself, or CFDP_BENCH_AS_GPUS -- a rehearsal of the
    N-GPU line (its config, its ride-along, its CPU baseline) by fewer ranks on a box that cannot hold N processes"""
    return int(os.environ.get("CFDP_BENCH_AS_GPUS", "0") or 0) or n_ranks


def mesh_divisor() -> int:
    """CFDP_BENCH_MESH_DIVISOR=d: every bench lattice is d times smaller per axis (contract tests of the multi-rank code
    paths on a tiny mesh; the line says so)"""
    return max(1, int(os.environ.get("CFDP_BENCH_MESH_DIVISOR", "1") or 1))


def default_bench_config(n_ranks: int) -> str:
    """BASELINE.json: config 2 on 1 GPU, config 3 on 4, config 5 on 8; 2 GPUs continue the strong series"""
    return {1: "dualgrid.12", 2: "dualgrid.24", 4: "dualgrid.48", 8: "dualgrid.384"}.get(n_ranks, "weak")


def bench_extra(name: str, n_ranks: int):
    """the measurement that rides along with a bench run of config `name` on n_ranks GPUs, as (key of the JSON line,
    config): the weak-scaling point at 2 / 4 GPUs, BASELINE config 4 (dualgrid.192, the strong-scaling point at ~33 k
    points per GPU) at 8; None otherwise"""
    if n_ranks in (2, 4) and name != "weak":
        return "weak_scaling", "weak"
    if n_ranks == 8 and name != "dualgrid.192":
        return "strong_scaling", "dualgrid.192"
    return None


def bench_config(name: str, n_ranks: int) -> dict:
    """one bench workload: lattice, number of domains (whole domains per GPU), the text of
    config.workload and the scaling label of the series the run belongs to"""
    dv = mesh_divisor()
    if name == "weak":
        dims, ndom = bench_mesh(n_ranks)
        dims = tuple(max(4, x // dv) for x in dims)
        return dict(name="weak", dims=dims, ndomains=ndom, scaling="weak",
                    workload=f"weak-scaling stand-in ({dims[0]}x{dims[1]}x{dims[2]}, {ndom} domains, {ndom // n_ranks} per GPU, "
                             f"262144 owned points per GPU" + (", halo exchange over xGMI)" if n_ranks > 1 else ")"))
    if name not in _BENCH:
        raise ValueError(f"unknown bench config {name!r}: one of {sorted(_BENCH)} or 'weak'")
    dims, ndom, level, scaling = _BENCH[name]
    if ndom % n_ranks:
        raise ValueError(f"{name}: {ndom} domains do not divide over {n_ranks} GPUs")
    dims = tuple(max(4, x // dv) for x in dims)
    what = f"{dims[0]}^3, {ndom} domains, {ndom // n_ranks} per GPU" + (f"; TINY MESH: lattice / {dv} per axis" if dv > 1 else "")
    how = "merged on 1 GPU, no halo exchange" if n_ranks == 1 else "halo exchange over xGMI with compute overlap"
    return dict(name=name, dims=dims, ndomains=ndom, scaling=scaling,
                workload=f"{name} {level} stand-in ({what}; {how})")


IPC_MODE_LABEL = {"coarse": "coarse-grained block", "split": "fine-grained flags, coarse-grained arenas, explicit invalidate",
                  "fine": "fine-grained block"}
IPC_NOTIFY_LABEL = {"counter": "counter notification", "flag": "flag notification"}


# what the environment asked for when the process started (RankSolver sets CFDP_IPC_MODE itself while it tries the modes)
_IPC_MODE_PRESET = os.environ.get("CFDP_IPC_MODE", "") or ("fine" if os.environ.get("CFDP_IPC_FINEGRAINED", "0") not in ("", "0") else "")


def ipc_mode_attempts() -> List[str]:
    """memory modes of the xGMI landing block in the order the set-up tries them (cfdproxy_hip.h, CFDP_IPC_MODE): the
    one the environment named (a comma-separated list is an order), else fine -> coarse -> split.  Fine-grained first: it
    is coherent between devices by definition, and in loopback it costs nothing (an iteration with exchange 41.8 us
    against 41.3 with a coarse-grained block and 43.4 with the split form, dualgrid.384 rank; DESIGN appendix C.4) --
    every ghost-row load bypasses the caches in every mode anyway"""
    if _IPC_MODE_PRESET:
        return [x for x in _IPC_MODE_PRESET.split(",") if x in IPC_MODE_LABEL] or ["fine"]
    return ["fine", "coarse", "split"]


_IPC_NOTIFY_PRESET = os.environ.get("CFDP_IPC_NOTIFY", "")


def ipc_notify_attempts() -> List[str]:
    """notification forms in the order the set-up tries them (cfdproxy_hip.h, CFDP_IPC_NOTIFY): counters raised by
    fire-and-forget atomic adds first (nothing on the boundary tile's critical path; relies on atomics to a peer's memory
    over xGMI, which only a validation on the machine itself can confirm), then flags stored by the tile that completes a
    partner's rows.  A form named in the environment is the only one tried"""
    if _IPC_NOTIFY_PRESET in IPC_NOTIFY_LABEL:
        return [_IPC_NOTIFY_PRESET]
    return ["counter", "flag"]


def device_census(device: int, rank: int, world: int, dist=None) -> dict:
    """collective: which PHYSICAL device every rank runs on (PCI bus ids, gathered) -- ordinals say nothing under a
    launcher that shows every rank one device (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES per rank)"""
    from . import device_bus_id
    mine = device_bus_id(device)
    ids: List[str] = [mine]
    if dist is not None and world > 1:
        ids = [None] * world  # type: ignore[list-item]
        dist.all_gather_object(ids, mine)
    per: Dict[str, int] = {}
    for b in ids:
        per[b] = per.get(b, 0) + 1
    return {"device_of_rank": list(ids), "distinct_devices": len(per), "ranks_per_device": max(per.values()),
            "shared_gpu": max(per.values()) > 1}


def exchange_requests(part: Domain, rank: int, world: int, dist=None, all_requests=None) -> None:
    """Fill part's send lists from what the partners request (comm_data.c:203-249 analogue).
    `dist`: an initialised torch.distributed module, or None with `all_requests` given
    (list over ranks of merge_requests() dicts) for in-process use."""
    mine = {int(k): (v[0], v[1]) for k, v in merge_requests(part).items()}
    if dist is not None and world > 1:
        gathered: List[Optional[dict]] = [None] * world
        dist.all_gather_object(gathered, mine)
    else:
        gathered = all_requests if all_requests is not None else [mine]
    for s, req in enumerate(gathered):
        if s == rank or not req:
            continue
        if rank in req:
            dom, idx = req[rank]
            merge_set_send(part, s, dom, idx)


def build_rank_partition(gp, n_domains: int, world: int, rank: int, via_files: bool = True,
                         var_kind: int = VAR_HASH, workdir: Optional[str] = None) -> Tuple[Domain, dict]:
    """Generate (optionally through dualgrid files + the drop-in loader) and merge the domains
    of `rank`.  Returns the merged partition (send lists not yet linked) and setup stats."""
    ids = rank_domain_list(rank, n_domains, world)  # blocks of ids, or the installed domain map
    count = len(ids)
    tmp = None
    doms = []
    if via_files:
        tmp = tempfile.mkdtemp(prefix=f"cfdp_r{rank}_", dir=workdir)
        prefix = os.path.join(tmp, "dualgrid")
        from . import write_mesh
        write_mesh(gp, prefix, 2, ids)
        doms = [load_domain(prefix, d, 2) for d in ids]
        shutil.rmtree(tmp, ignore_errors=True)
    else:
        doms = [gen_domain(gp, d) for d in ids]
    for d, dom in zip(ids, doms):
        fill_var(dom, gen_global_ids(gp, d, dom.nall), var_kind, gp.nx, gp.ny, gp.nz)
    part = merge_domains(doms, ids, n_domains, world, rank)
    mi = part.merge_info.contents
    for dl, dom in enumerate(doms):
        l2m = np.ctypeslib.as_array(mi.local2merged[dl], shape=(dom.nall,))
        part.var[l2m] = dom.var
    stats = dict(domains=count, nown=part.nown, nghost=part.nall - part.nown, nfaces=part.nfaces,
                 faces_dropped=int(mi.nfaces_dropped))
    for dom in doms:
        dom.free()
    return part, stats


class RankSolver:
    """The per-rank iteration: gradients (+ halo exchange) + pseudo flux on one GPU."""

    def __init__(self, part: Domain, rank: int, world: int, device: int, dist=None,
                 transport: str = "auto", tile_points: int = 0, grad_lanes: int = 0, flux_lanes: int = 0,
                 fusion: bool = True):
        import torch

        self.torch = torch
        self.dist = dist
        self.rank, self.world = rank, world
        self.transport = transport
        torch.cuda.set_device(device)
        self.device = torch.device("cuda", device)
        self.gpu = GpuPartition(part, device=device, tile_points=tile_points, grad_lanes=grad_lanes,
                                flux_lanes=flux_lanes)
        c = self.gpu.counts()
        self.nown, self.nall, self.nsend, self.nrecv = c["nown"], c["nall"], c["nsend"], c["nrecv"]
        self.partners = self.gpu.partners()
        # torch owns the buffers RCCL touches: grad (its ghost block = receive side) and the send arena
        self.grad_t = torch.empty(self.nall * ROWLEN, dtype=torch.float64, device=self.device)
        self.send_t = torch.empty(max(self.nsend, 1) * ROWLEN, dtype=torch.float64, device=self.device)
        self.gpu.bind_grad(self.grad_t.data_ptr())
        self.gpu.bind_sendbuf(self.send_t.data_ptr())
        # fused iterations (flux(i) rides with gradients(i+1)): grad is double-buffered and the two
        # buffers swap roles every iteration; the exchange delivers into the current one
        self.grad_bufs = [self.grad_t]
        if fusion:
            self.grad_bufs.append(torch.empty_like(self.grad_t))
            self.gpu.bind_grad_alt(self.grad_bufs[1].data_ptr())
            self.gpu.set_fusion(True)
        # the context's own HIP streams, seen from torch (the transport enqueues on the comm stream;
        # all stream ordering of an iteration lives in cfdp_gpu_step_pre/_post)
        self.s_main = torch.cuda.ExternalStream(self.gpu.stream(0), device=self.device)
        self.s_comm = torch.cuda.ExternalStream(self.gpu.stream(1), device=self.device)
        self.send_views, self.recv_views = [], []
        for s in range(len(self.partners)):
            sp, sb = self.gpu.send_slice(s)
            rp, rb = self.gpu.recv_slice(s)
            so, ro = (sp - self.send_t.data_ptr()) // 8, (rp - self.grad_t.data_ptr()) // 8
            self.send_views.append(self.send_t[so:so + sb // 8])
            # whole rows inside the ghost block, one view per grad buffer
            self.recv_views.append([b[ro:ro + rb // 8] for b in self.grad_bufs])
        # "auto": set up both device-side transports and let choose_transport() time them
        self.available: List[str] = []
        self.probe: Dict[str, float] = {}
        self.checks: Dict[str, bool] = {}   # choose_transport: did every row arrive in its slot, per transport
        # what the set-up validation of every attempted transport saw (all ranks hold the same all-reduced evidence):
        # {"ipc / as configured": {"ok": False, "failed": "stale rows", "wait_timeouts": 0, "worst_mismatch": 0.31}, ...}
        # -- the datum that says WHICH check a transport failed on a machine nobody could test on before
        self.validation: Dict[str, dict] = {}
        # which physical device every rank runs on (PCI bus ids): ranks that SHARE a device -- found by comparing the ids,
        # not ordinals or device counts, which say nothing when a launcher shows every rank one device -- or
        # CFDP_SHARED_GPU=1 as the explicit override
        self.census = device_census(device, rank, world, dist)
        self.shared_gpu = self.census["shared_gpu"] or os.environ.get("CFDP_SHARED_GPU") == "1"
        if transport in ("ipc", "auto") and world > 1:
            import sys
            # ranks that SHARE a device must not wait inside the fused pass: the boundary tiles of every rank would sit in
            # the device's workgroup slots, spinning, while the pass whose flags they wait for cannot get a slot (measured:
            # 4 ranks at 128^3 on one MI355X starve each other until the bounded waits give up).  One waiting workgroup
            # per rank (the wait kernel) cannot exhaust the device.  (A CFDP_IPC_WAIT_INKERNEL in the environment wins.)
            wait_in = None if "CFDP_IPC_WAIT_INKERNEL" in os.environ else (False if self.shared_gpu else None)
            if rank == 0:
                print(f"[cfdp] {world} ranks on {self.census['distinct_devices']} device(s), at most "
                      f"{self.census['ranks_per_device']} per device: the wait for an exchange runs "
                      + ("as a kernel of its own (ranks share a device)" if wait_in is False else
                         "inside the fused pass" if wait_in is None else "as the environment says"), file=sys.stderr)
            # the rungs of the exchange, in the order they are tried (every rank fails or passes alike: _init_ipc raises
            # from all-reduced evidence only): memory modes of the landing block, and per mode the notification forms.
            # Each attempt is configured BY ARGUMENT (cfdp_gpu_ipc_configure): the process environment is left alone, so
            # a later solver of this process starts from the user's presets again, not from the last rung tried here
            # Last rung on the IPC mappings, before any other transport: push, notify and wait as KERNELS OF THEIR OWN
            # with flags (plain release stores / acquire loads at kernel boundaries, nothing handed over inside a kernel).
            # Why it stands in front of RCCL: priced on one GPU, same partitions, steps between two syncs (bench.py,
            # exchange_protocol_loopback; profiles/r05_bench_n1_steps20.json) an iteration takes 53 / 25 us this way on the
            # dualgrid.384 / dualgrid.192 partitions (38 / 10 us without exchange) but 73 / 65 us with stream-launched RCCL
            # send/recv -- RCCL cannot be captured into a hipGraph in this ROCm, so every step pays its launches
            rungs = [(m, n, None) for m in ipc_mode_attempts() for n in ipc_notify_attempts()]
            if "CFDP_IPC_INKERNEL" not in os.environ:
                rungs += [(m, "flag", False) for m in ipc_mode_attempts()]
                # ... and behind it MPI_Put's pattern (src/exchange_data_mpidma.c:93-127): the send arena packed by a kernel, one
                # copy per partner slice into its landing slice, the notify kernel.  Priced in the same table: 0.52 / 0.16 of
                # comm-free -- a shade above stream-launched RCCL (0.49 / 0.15), graph-replayed, and the one rung that does not
                # depend on a KERNEL's stores reaching a peer's memory
                rungs += [(m, "flag", "put") for m in ipc_mode_attempts()]
            for mode, notify, push_in in rungs:
                self.gpu.ipc_configure(memory_mode=mode, wait_inkernel=wait_in, notify=notify, push_inkernel=push_in)
                self._ipc_mode_now = mode
                self._validating = (f"ipc / {IPC_MODE_LABEL[mode]}, {IPC_NOTIFY_LABEL[notify]}"
                                    + (", push / notify / wait as kernels of their own" if push_in is False else "")
                                    + (", copy-engine put" if push_in == "put" else ""))
                try:
                    self._init_ipc()
                    self.available.append("ipc")
                    break
                except Exception as e:
                    self.validation.setdefault(self._validating, {"ok": False, "failed": f"setup: {e}"[:160]})
                    print(f"[rank {rank}] xGMI write+notify setup failed ({self._validating}: {e})", file=sys.stderr)
            if "ipc" not in self.available and transport == "ipc":
                transport = "rccl"
        elif transport in ("ipc", "auto"):
            transport = "rccl"  # one rank: no exchange at all
        if transport == "rccl" and world > 1 and dist.get_backend() != "nccl":
            raise ValueError("the rccl transport needs one device per rank (process group backend nccl)")
        if transport in ("rccl", "auto") and world > 1 and dist.get_backend() == "nccl":
            try:
                self._init_own_communicator()
                self.available.append("rccl")
                self.validation["rccl"] = {"ok": True, "failed": None, "check": "communicator created (values: exchange_check after the run)"}
            except Exception as e:  # keep going on the torch.distributed transport
                self.validation["rccl"] = {"ok": False, "failed": f"setup: {e}"[:160]}
                import sys
                print(f"[rank {rank}] own RCCL communicator failed ({e}); using torch.distributed P2P", file=sys.stderr)
                if transport == "rccl":
                    transport = "torch"
        if transport == "auto":
            transport = self.available[0] if self.available else "torch"
        if "ipc" in self.available and transport != "ipc":
            self.gpu.ipc_enable(False)
        self.transport = transport
        if transport == "staged":
            self.h_send = [torch.empty(v.numel(), dtype=torch.float64).pin_memory() for v in self.send_views]
            self.h_recv = [torch.empty(v[0].numel(), dtype=torch.float64).pin_memory() for v in self.recv_views]

    @staticmethod
    def torch_rccl_path() -> str:
        import torch
        p = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        return p if os.path.exists(p) else ""

    def _coll_device(self):
        """where tensors of the setup collectives live: the GPU with the nccl backend, else the host"""
        return self.device if self.dist.get_backend() == "nccl" else "cpu"

    def _all_ok(self, ok: bool) -> bool:
        t = self.torch.tensor([1 if ok else 0], device=self._coll_device())
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
        return bool(int(t.item()))

    def _init_ipc(self) -> None:
        """collective: export / exchange handles / connect / validate; raises on every rank alike"""
        dist = self.dist
        ok, info = True, None
        try:
            handle, land = self.gpu.ipc_export()
            roff = [0]
            for s in range(len(self.partners)):
                roff.append(roff[-1] + self.gpu.recv_slice(s)[1] // (8 * ROWLEN))
            info = dict(handle=handle, fhandle=self.gpu.ipc_export_flags(), land=land, partners=list(self.partners), recv_off=roff)
        except Exception as e:
            ok, self._ipc_why = False, str(e)
        gathered = [None] * self.world
        dist.all_gather_object(gathered, info if ok else None)
        if any(g is None for g in gathered):
            raise RuntimeError("a rank could not export its IPC block (%s)" % getattr(self, "_ipc_why", "other rank"))
        try:
            for s, p in enumerate(self.partners):
                pi = gathered[p]
                t = pi["partners"].index(self.rank)
                base = self.gpu.lib.cfdp_gpu_ipc_header_bytes() + pi["recv_off"][t] * 8 * ROWLEN
                foff = self.gpu.lib.cfdp_gpu_ipc_flag_offset(t)  # (the word of my slot there: a cache line of its own)
                self.gpu.ipc_connect(s, pi["handle"], base, base + pi["land"], foff)
                self.gpu.ipc_connect_flags(s, pi["fhandle"], foff)  # (a block of its own in "split" mode)
            self.gpu.ipc_ready()
        except Exception as e:
            ok, self._ipc_why = False, str(e)
        if not self._all_ok(ok):  # (also the barrier: nobody pushes before everybody is ready)
            self._ipc_off()
            raise RuntimeError("a rank could not map its partners' IPC blocks (%s)" % getattr(self, "_ipc_why", "other rank"))
        # what the library RESOLVED on every rank (the per-partner protocol -- and with it notification by counters -- depends on
        # a rank's own partition; neighbours of different forms understand each other: the sender states what its word
        # advances by, csrc/gg_kernels.h): the rung is recorded under what really runs, and a rung whose resolved
        # configuration has been tried already is not validated twice
        forms = [None] * self.world
        dist.all_gather_object(forms, self.gpu.ipc_mode().get("notify_by", "?").split(" ")[0])
        resolved = forms[0] if len(set(forms)) == 1 else "mixed: " + ", ".join(f"{f} on rank {r}" for r, f in enumerate(forms))
        asked = self._validating
        self._validating = (asked.split(",")[0] + f", notification by {resolved}"
                            + (", push / notify / wait as kernels of their own" if "kernels of their own" in asked else "")
                            + (", copy-engine put" if "copy-engine put" in asked else ""))
        self.resolved_rungs = getattr(self, "resolved_rungs", [])
        if self._validating in self.resolved_rungs:
            self._ipc_off()
            raise RuntimeError(f"resolves to a configuration already tried ({self._validating})")
        self.resolved_rungs.append(self._validating)
        self.transport = "ipc"
        if not self.validate_exchange():
            self._ipc_off()
            raise RuntimeError("the exchange check failed")

    def use_transport(self, name: str) -> None:
        """switch between the transports set up by transport="auto" (collective: all ranks alike)"""
        if name not in self.available:
            raise ValueError(f"transport {name} is not available ({self.available})")
        self.synchronize()
        if "ipc" in self.available:
            self.gpu.ipc_enable(name == "ipc")
        self.transport = name
        self.dist.barrier()

    def choose_transport(self, steps: int = 200) -> str:
        """collective: time `steps` overlapped iterations on every available transport (max over
        ranks) and keep the fastest; self.probe holds the microseconds per iteration"""
        import time
        torch, dist = self.torch, self.dist
        for name in list(self.available):
            self.use_transport(name)
            self.run_steps(56, with_exchange=True, overlap=True)  # warm: the step graph is captured here
            self.synchronize()
            dist.barrier()
            t = time.perf_counter()
            self.run_steps(steps, with_exchange=True, overlap=True)
            self.synchronize()
            dt = torch.tensor([time.perf_counter() - t], dtype=torch.float64, device=self._coll_device())
            dist.all_reduce(dt, op=dist.ReduceOp.MAX)
            self.probe[name] = float(dt.item()) / steps * 1e6
            # a transport whose rows did not all arrive in their slots (or, ipc: a wait that gave up) is
            # out, however fast it looked (all ranks see the same gathered evidence)
            self.checks[name] = bool(self.exchange_check(stale_read_steps=56)["ok"])
            if not self.checks[name]:
                del self.probe[name]
                self.available.remove(name)
                if name == "ipc":
                    self.gpu.ipc_enable(False)
        if self.probe:
            self.use_transport(min(self.probe, key=self.probe.get))
        elif self.transport not in self.available:
            self.transport = "torch"  # nothing device-side is left: torch.distributed P2P ops
        return self.transport

    def exchange_check(self, stale_read_steps: int = 0) -> dict:
        """collective: every row that was sent must have arrived in ITS slot.  Per partner slice a
        position-weighted checksum of the rows this rank sent must equal the partner's checksum of the
        ghost rows it received from this rank (row j of a message weighs j+1, so permuted or mis-slotted
        rows do not cancel); something must have been sent at all; no device-side wait may have given up"""
        torch, dist, part = self.torch, self.dist, self.gpu.dom
        stale = self.stale_read_check(batches=(stale_read_steps,)) if stale_read_steps > 0 else None
        g = self.grad_host()
        mine = {}
        for k in part.partners:
            sidx, ridx = part.sendindex(k), part.recvindex(k)
            ws, wr = np.arange(1, len(sidx) + 1.0), np.arange(1, len(ridx) + 1.0)
            mine[int(k)] = (float((np.abs(g[sidx]).sum(axis=(1, 2)) * ws).sum()), float(np.abs(g[sidx]).sum()),
                            float((np.abs(g[ridx]).sum(axis=(1, 2)) * wr).sum()))
        allc = [None] * self.world
        dist.all_gather_object(allc, mine)
        ok, sent_total, worst = True, 0.0, 0.0
        for a in range(self.world):
            for b, (ws_ab, s_ab, _) in allc[a].items():
                got = allc[b].get(a, (0.0, 0.0, float("nan")))[2]  # what b received from a
                sent_total += s_ab
                rel = abs(ws_ab - got) / max(abs(ws_ab), 1e-300)
                worst = max(worst, rel if rel == rel else float("inf"))
                ok = ok and rel <= 1e-9
        chk = {"sum_abs_sent_rows": sent_total, "worst_partner_slice_mismatch": worst,
               "check": "position-weighted |row| sums per partner slice, sender vs receiver",
               "ok": bool(ok and sent_total > 0.0)}
        if self.transport == "ipc":
            et = torch.tensor([float(self.gpu.ipc_error() != 0)], dtype=torch.float64, device=self._coll_device())
            dist.all_reduce(et)
            chk["wait_timeouts"] = int(et.item())
            chk["ok"] = chk["ok"] and int(et.item()) == 0
        if stale is not None:  # did any flux phase of `stale_read_steps` more steps read a row of an earlier exchange?
            chk["stale_read_check"] = stale
            chk["ok"] = chk["ok"] and stale["ok"]
        return chk

    def fallback(self) -> bool:
        """collective (every rank takes the same decision from the same all-reduced evidence): give up
        the current transport for the next best one; False when nothing is left to fall back to"""
        cur = self.transport
        last = "torch" if self.dist.get_backend() == "nccl" else "staged"
        if cur == last or self.world == 1:
            return False
        self.synchronize()
        if cur in self.available:
            self.available.remove(cur)
        self.probe.pop(cur, None)
        if cur == "ipc":
            self.gpu.ipc_enable(False)
        if self.available:
            self.use_transport(min(self.available, key=lambda n: self.probe.get(n, float("inf"))))
            return True
        if last == "staged" and not hasattr(self, "h_send"):
            torch = self.torch
            self.h_send = [torch.empty(v.numel(), dtype=torch.float64).pin_memory() for v in self.send_views]
            self.h_recv = [torch.empty(v[0].numel(), dtype=torch.float64).pin_memory() for v in self.recv_views]
        self.transport = last
        self.dist.barrier()
        return True

    def _ipc_off(self) -> None:
        try:
            self.gpu.sync()
        finally:
            self.dist.barrier()
            self.gpu.ipc_disconnect()
            self.dist.barrier()

    def stale_read_check(self, batches=(1, 2, 3, 5, 8, 60, 107), overlap: bool = True, before_steps=None) -> dict:
        """collective: can any flux phase have READ a ghost row before the rows of its exchange had landed?  The
        field is constant in time and the landing arenas alternate, so a row read one exchange early has the value
        of the right one and no comparison of final states can tell (the reference asserts stage / flag lock-step
        at every receive instead, src/exchange_data_mpi.c:189,439, src/exchange_data_gaspi.c:389-416).  Here the
        library scales var by 2, 2, 1/4, ... after every iteration (cfdp_gpu_scaled_check_begin): iteration k's
        gradients, ghost rows and flux are the first iteration's times 2^((k-1) mod 3) exactly, and a device kernel
        compares the flux EVERY step produced -- in the schedule that is timed: stream launches, hipGraph replays,
        in-kernel wait and push -- with reference * 2^e bit for bit.  The reference flux comes from an iteration
        whose exchange is complete beyond doubt: device sync on every rank, barrier, one step without exchange.
        `before_steps` (tests): called on every rank between the set-up and the scaled steps."""
        torch, dist = self.torch, self.dist
        self.run_steps(2, with_exchange=True, overlap=overlap)  # both grad buffers / arenas hold delivered rows
        self.synchronize()
        dist.barrier()
        self.run_steps(1, with_exchange=False, overlap=overlap)
        self.synchronize()
        self.gpu.scaled_check_begin()
        dist.barrier()
        if before_steps is not None:
            before_steps()
        try:
            for k in batches:
                self.run_steps(int(k), with_exchange=True, overlap=overlap)
        finally:
            ev = self.gpu.scaled_check_end()
        err = float(self.gpu.ipc_error() != 0) if self.transport == "ipc" else 0.0
        t = torch.tensor([float(ev["mismatches"]), float(ev["mismatches"] > 0), err, float(ev["var_mismatches"])], dtype=torch.float64,
                         device=self._coll_device())
        dist.all_reduce(t)
        tmin = torch.tensor([float(ev["flux_checks"])], dtype=torch.float64, device=self._coll_device())
        dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
        first = None
        if ev["mismatches"]:
            first = {"rank": self.rank, "iteration": ev["first_iteration"], "point": ev["first_point"],
                     "component": ev["first_component"], "seen": ev["seen"], "expected": ev["expected"]}
        firsts = [None] * self.world
        dist.all_gather_object(firsts, first)
        out = {"steps": int(sum(batches)), "flux_fields_compared_per_rank": int(tmin.item()),
               "stale_reads": int(t[0].item()), "ranks_with_stale_reads": int(t[1].item()),
               "wait_timeouts": int(t[2].item()), "first": next((f for f in firsts if f), None),
               "var_bookkeeping_errors": int(t[3].item()),
               "check": "var scaled by 2, 2, 1/4, ... after every iteration; the flux of every step == reference x 2^e, "
                        "bit for bit, compared on the device",
               "ok": bool(t[0].item() == 0 and t[2].item() == 0 and t[3].item() == 0 and tmin.item() >= sum(batches) - 1)}
        dist.barrier()
        return out

    def validate_exchange(self) -> bool:
        """collective, before a transport is used: (1) no flux phase may read a ghost row early (stale_read_check over
        stream-launched steps and hipGraph replays), (2) after it every ghost row must equal its owner's row
        (position-weighted sums of |rows| sent and received over all ranks), (3) no device-side wait may give up.
        WHICH of them failed is recorded: "stale read" (a flux saw a row of an earlier exchange), "stale rows" (rows
        that never arrived although every flag did), "wait timeout" (a partner's flag never arrived)"""
        torch, dist, part = self.torch, self.dist, self.gpu.dom
        lib = self.gpu.lib
        # a broken mapping must not cost half a minute per iteration here; ranks that time-slice ONE device (rehearsals)
        # wait for each other's turn on it, which takes seconds with 6 of them
        lib.cfdp_ipc_set_wait_seconds(float(os.environ.get("CFDP_IPC_VALIDATE_WAIT_SECONDS") or
                                            (20.0 if getattr(self, "shared_gpu", False) else 2.0)))
        try:
            ev = self.stale_read_check()
            g = self.grad_host()
            sent = got = 0.0
            for k in part.partners:
                sidx, ridx = part.sendindex(k), part.recvindex(k)
                sent += float((np.abs(g[sidx]).sum(axis=(1, 2)) * np.arange(1, len(sidx) + 1.0)).sum())
                got += float((np.abs(g[ridx]).sum(axis=(1, 2)) * np.arange(1, len(ridx) + 1.0)).sum())
            t = torch.tensor([sent, got, float(self.gpu.ipc_error() != 0) if self.transport == "ipc" else 0.0],
                             dtype=torch.float64, device=self._coll_device())
            dist.all_reduce(t)
            sent, got, err = (float(x) for x in t)
        finally:
            lib.cfdp_ipc_set_wait_seconds(float(os.environ.get("CFDP_IPC_WAIT_SECONDS", "10")))
        timeouts = int(err) + ev["wait_timeouts"]
        worst = abs(sent - got) / sent if sent > 0 else float("inf")
        rows_ok = sent > 0 and abs(sent - got) <= 1e-9 * sent
        good = ev["ok"] and rows_ok and timeouts == 0
        failed = None if good else ("wait timeout" if timeouts else "nothing sent" if not sent > 0 else
                                    "stale read" if ev["stale_reads"] else "stale rows" if not rows_ok else "too few flux checks")
        self.validation[getattr(self, "_validating", self.transport)] = {
            "ok": bool(good), "failed": failed, "wait_timeouts": timeouts, "stale_reads": ev["stale_reads"],
            "first_stale_read": ev["first"], "steps": ev["steps"], "worst_sum_mismatch": worst,
            "check": "scaled field: flux of every one of %d steps (stream launches + hipGraph replays) == reference x 2^e "
                     "on the device; then position-weighted sum |sent rows| vs |ghost rows| over all ranks" % ev["steps"]}
        return bool(good)

    def _init_own_communicator(self) -> None:
        """collective over the process group; raises on every rank alike when a step fails"""
        torch, dist = self.torch, self.dist
        lib = self.torch_rccl_path()

        all_ok = self._all_ok
        ok = True
        try:
            self.gpu._ck(self.gpu.lib.cfdp_rccl_load(lib.encode()))
        except Exception:
            ok = False
        if not all_ok(ok):
            raise RuntimeError("RCCL could not be resolved on every rank")
        box = [None]
        if self.rank == 0:
            try:
                box[0] = GpuPartition.rccl_unique_id(lib)
            except Exception:
                box[0] = None
        dist.broadcast_object_list(box, src=0)
        if box[0] is None:
            raise RuntimeError("ncclGetUniqueId failed on rank 0")
        self.gpu.rccl_init(box[0], self.world, self.rank, None, lib)

    # ------------------------------------------------------------------------------ pieces
    def _exchange(self) -> None:
        """halo exchange, enqueued on the context's comm stream (step_pre has made that stream wait
        for the pack and for the previous flux, step_post makes the flux wait for it)"""
        torch, dist = self.torch, self.dist
        cur = 0
        if len(self.grad_bufs) > 1 and self.gpu.grad_ptr() == self.grad_bufs[1].data_ptr():
            cur = 1  # the buffer this iteration's gradients went to
        with torch.cuda.stream(self.s_comm):
            if self.transport == "torch":
                ops = []
                for s, peer in enumerate(self.partners):
                    if self.send_views[s].numel():
                        ops.append(dist.P2POp(dist.isend, self.send_views[s], peer))
                    if self.recv_views[s][cur].numel():
                        ops.append(dist.P2POp(dist.irecv, self.recv_views[s][cur], peer))
                for w in dist.batch_isend_irecv(ops):
                    w.wait()  # orders s_comm after the RCCL stream; does not block the host
            else:  # staged through the host (tests only)
                for s in range(len(self.partners)):
                    self.h_send[s].copy_(self.send_views[s], non_blocking=True)
                self.s_comm.synchronize()
                reqs = []
                for s, peer in enumerate(self.partners):
                    if self.h_send[s].numel():
                        reqs.append(dist.isend(self.h_send[s], peer))
                    if self.h_recv[s].numel():
                        reqs.append(dist.irecv(self.h_recv[s], peer))
                for r in reqs:
                    r.wait()
                for s in range(len(self.partners)):
                    self.recv_views[s][cur].copy_(self.h_recv[s], non_blocking=True)

    def step(self, with_exchange: bool = True, overlap: bool = True, with_flux: bool = True,
             flux_mode: int = FLUX_CONSISTENT) -> None:
        """one iteration = what test_solver times (reference src/solver.c:48-54): two ABI calls
        around one communication call"""
        comm = with_exchange and self.world > 1 and bool(self.partners)
        if self.transport == "ipc":
            self.gpu.step_ipc(comm, overlap, with_flux, flux_mode)
            return
        if self.transport == "rccl":
            self.gpu.step_rccl(comm, overlap, with_flux, flux_mode)  # one call: brackets + ncclGroup
            return
        self.gpu.step_pre(comm, overlap)
        if comm:
            self._exchange()
        self.gpu.step_post(with_flux, flux_mode)

    def run_steps(self, steps: int, with_exchange: bool = True, overlap: bool = True, with_flux: bool = True,
                  flux_mode: int = FLUX_CONSISTENT) -> None:
        """`steps` iterations (one library call with the library's own communicator)"""
        comm = with_exchange and self.world > 1 and bool(self.partners)
        if self.transport == "ipc":
            # use_graph = 2: a batch of up to 64 steps is ONE graph, the flux of its last iteration and the wait for its last
            # exchange included (what the caller's synchronize() would enqueue behind it anyway)
            self.gpu.run_steps_ipc(steps, comm, overlap, with_flux, flux_mode, use_graph=2)
            return
        if self.transport == "rccl":
            self.gpu.run_steps_rccl(steps, comm, overlap, with_flux, flux_mode)
            return
        for _ in range(steps):
            self.step(with_exchange, overlap, with_flux, flux_mode)

    def synchronize(self) -> None:
        self.gpu.sync()  # also runs a flux deferred by the fused mode
        self.torch.cuda.synchronize(self.device)

    def grad_host(self) -> np.ndarray:
        """grad in FILE (merged-partition) numbering, ghost rows included"""
        self.synchronize()
        self.gpu.pull_fields()
        return self.gpu.dom.grad

    def close(self) -> None:
        if "ipc" in self.available and self.world > 1:
            self._ipc_off()  # nobody unmaps or frees a block a partner may still write to
        self.gpu.close()
