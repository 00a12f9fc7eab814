"""cfd-proxy_amd -- Python host side of the MI355X-native CFD-Proxy hot path.

The product is the C-ABI library ``lib/libcfdproxy_hip.so`` (C host code + hand-written
gfx950 HIP kernels; see ``include/*.h``).  This module is a thin ctypes binding of that
ABI for tests, ``bench.py`` and the multi-process (one rank per GPU, torch.distributed /
RCCL) driver.  PyTorch is only plumbing here: device buffers handed to RCCL and the
process group.  Nothing in this module computes on the CPU: without the HIP library and
a GPU the solver classes raise.

The directory name contains a hyphen, so import it with ``importlib`` (see
``__graft_entry__.py``) -- it registers itself as ``cfd_proxy_amd``.

Reference interfaces mirrored (file:line into /root/reference/src):
  Domain            <- solver_data + comm_data              solver_data.h:66-81, comm_data.h:15-55
  load_domain       <- main()'s open/read sequence           hybrid.f6.c:56-79
  Plan              <- init_threads()                        threads.c:730-788
  GpuPartition      <- compute_gradients_gg_* / compute_psd_flux / exchange_dbl_copy_in/out
                                                             gradients.c:150-336, flux.c:194-200,
                                                             threads.c:791-869
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import List, Optional, Sequence

import numpy as np

NGRAD = 7
NFLUX = 3
_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)

FLUX_CONSISTENT = 0
FLUX_REFERENCE = 1
TILES_ALL, TILES_BOUNDARY, TILES_INTERIOR = 0, 1, 2
VAR_ONE, VAR_HASH, VAR_LINEAR = 0, 1, 2


# --------------------------------------------------------------------------- build / load
def build(verbose: bool = False) -> None:
    """Compile the host library, the HIP library (gfx950) and the driver in-tree."""
    r = subprocess.run(["make", "-C", _HERE, "all"], capture_output=True, text=True)
    if verbose or r.returncode:
        print(r.stdout[-4000:])
        print(r.stderr[-4000:])
    if r.returncode:
        raise RuntimeError("building cfd-proxy_amd failed")


def _load(name: str) -> C.CDLL:
    # CFDP_LIBDIR: load a differently built library (kernel experiments); default = in-tree lib/
    path = os.path.join(os.environ.get("CFDP_LIBDIR") or os.path.join(_HERE, "lib"), name)
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing: run `make -C cfd-proxy_amd` (or __graft_entry__.build())")
    return C.CDLL(path, mode=C.RTLD_GLOBAL)


_host_lib: Optional[C.CDLL] = None
_hip_lib: Optional[C.CDLL] = None


class SolverData(C.Structure):
    _fields_ = [
        ("nfaces", C.c_int), ("nallfaces", C.c_int), ("nownpoints", C.c_int),
        ("nallpoints", C.c_int), ("ncolors", C.c_int),
        ("fpoint", C.POINTER(C.c_int)), ("fnormal", C.POINTER(C.c_double)),
        ("pvolume", C.POINTER(C.c_double)), ("var", C.POINTER(C.c_double)),
        ("grad", C.POINTER(C.c_double)), ("psd_flux", C.POINTER(C.c_double)),
        ("fcolor", C.c_void_p), ("niter", C.c_int), ("gpu", C.c_void_p),
    ]


class CommData(C.Structure):
    _fields_ = [
        ("nProc", C.c_int), ("iProc", C.c_int), ("ndomains", C.c_int), ("ncommdomains", C.c_int),
        ("nownpoints", C.c_int), ("naddpoints", C.c_int),
        ("addpoint_owner", C.POINTER(C.c_int)), ("addpoint_id", C.POINTER(C.c_int)),
        ("commpartner", C.POINTER(C.c_int)), ("sendcount", C.POINTER(C.c_int)),
        ("recvcount", C.POINTER(C.c_int)),
        ("recvindex", C.POINTER(C.POINTER(C.c_int))), ("sendindex", C.POINTER(C.POINTER(C.c_int))),
        ("nreq", C.c_int), ("req", C.c_void_p), ("stat", C.c_void_p),
        ("recvbuf", C.c_void_p), ("sendbuf", C.c_void_p),
        ("remote_recv_offset", C.c_void_p), ("local_recv_offset", C.c_void_p),
        ("local_send_offset", C.c_void_p), ("notification", C.c_void_p),
        ("recv_flag", C.c_void_p), ("send_flag", C.c_void_p),
        ("recv_stage", C.c_int), ("send_stage", C.c_int), ("comm_stage", C.c_int),
        ("group", C.c_void_p),
    ]


class GenParams(C.Structure):
    _fields_ = [
        ("nx", C.c_int), ("ny", C.c_int), ("nz", C.c_int), ("ndomains", C.c_int),
        ("connectivity", C.c_int), ("normals", C.c_int), ("volumes", C.c_int),
        ("ghost_faces", C.c_int), ("cdf_version", C.c_int), ("seed", C.c_uint64),
        ("numbering", C.c_int), ("hubs", C.c_int),
    ]


class MergeInfo(C.Structure):
    _fields_ = [
        ("G", C.c_int), ("r", C.c_int), ("ndomains_total", C.c_int), ("ndom_local", C.c_int),
        ("domain_ids", C.POINTER(C.c_int)), ("own_offset", C.POINTER(C.c_int)),
        ("local2merged", C.POINTER(C.POINTER(C.c_int))),
        ("nghost", C.c_int), ("ghost_domain", C.POINTER(C.c_int)), ("ghost_idx", C.POINTER(C.c_int)),
        ("npartners", C.c_int), ("partner", C.POINTER(C.c_int)), ("want_off", C.POINTER(C.c_int)),
        ("nfaces_in", C.c_long), ("nfaces_dropped", C.c_long),
    ]


class TileDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in
                ("pstart", "npts", "nhalo", "nfaces", "ninc", "halo_off", "blob_off", "blob_qw")]


class PlanOpts(C.Structure):
    _fields_ = [("tile_points", C.c_int), ("boundary_first", C.c_int), ("supertile", C.c_int)]


class PlanStruct(C.Structure):
    _fields_ = [
        ("nown", C.c_int), ("nall", C.c_int), ("nfaces_used", C.c_long),
        ("ntiles", C.c_int), ("nbtiles", C.c_int), ("tile_points", C.c_int),
        ("new2old", C.POINTER(C.c_int)), ("old2new", C.POINTER(C.c_int)),
        ("tiles", C.POINTER(TileDesc)),
        ("halo_idx", C.POINTER(C.c_int)), ("nhalo_total", C.c_long),
        ("blob", C.POINTER(C.c_ubyte)), ("blob_bytes", C.c_long),
        ("vol", C.POINTER(C.c_double)), ("degree", C.POINTER(C.c_int)),
        ("lds_grad", C.c_long), ("lds_flux", C.c_long),
        ("lds_grad_cls", C.c_long * 2), ("lds_flux_cls", C.c_long * 2),
        ("nfaces_dup", C.c_long), ("ninc_total", C.c_long),
        ("npartners", C.c_int), ("partner", C.POINTER(C.c_int)),
        ("send_off", C.POINTER(C.c_int)), ("send_idx", C.POINTER(C.c_int)),
        ("recv_off", C.POINTER(C.c_int)),
        ("ngroups", C.c_int), ("group_begin", C.c_int * 5), ("group_class", C.c_int * 4),
    ]


def _declare_host(lib: C.CDLL) -> None:
    P = C.POINTER
    lib.cfdp_tile_class_of.argtypes = [C.c_int, C.c_int, C.c_long]
    lib.cfdp_gen_domain.argtypes = [P(GenParams), C.c_int, P(SolverData), P(CommData)]
    lib.cfdp_gen_write_domain.argtypes = [P(GenParams), C.c_int, C.c_char_p, C.c_int]
    lib.cfdp_write_domain_file.argtypes = [C.c_char_p, P(SolverData), P(CommData), C.c_int]
    lib.cfdp_gen_global_ids.argtypes = [P(GenParams), C.c_int, P(C.c_int)]
    lib.cfdp_free_solver_data.argtypes = [P(SolverData)]
    lib.cfdp_free_solver_data.restype = None
    lib.cfdp_free_comm_data.argtypes = [P(CommData)]
    lib.cfdp_free_comm_data.restype = None
    lib.cfdp_load_domain.argtypes = [C.c_char_p, C.c_int, C.c_int, P(SolverData), P(CommData)]
    lib.cfdp_domain_rank.argtypes = [C.c_int, C.c_int, C.c_int]
    lib.cfdp_rank_domains.argtypes = [C.c_int, C.c_int, C.c_int, P(C.c_int), P(C.c_int)]
    lib.cfdp_rank_domains.restype = None
    lib.cfdp_set_domain_map.argtypes = [P(C.c_int), C.c_int, C.c_int]
    lib.cfdp_set_domain_map.restype = None
    lib.cfdp_rank_domain_list.argtypes = [C.c_int, C.c_int, C.c_int, P(C.c_int)]
    lib.cfdp_cluster_domains.argtypes = [C.c_int, C.c_int, P(C.c_int), P(C.c_int), P(C.c_int), P(C.c_int)]
    lib.cfdp_cluster_domains.restype = C.c_long
    lib.cfdp_domain_graph.argtypes = [C.c_char_p, C.c_int, C.c_int, P(P(C.c_int)), P(P(C.c_int)), P(P(C.c_int))]
    lib.cfdp_merge_domains.argtypes = [C.c_int, P(C.c_int), P(SolverData), P(CommData), C.c_int,
                                       C.c_int, C.c_int, P(SolverData), P(CommData),
                                       P(P(MergeInfo))]
    lib.cfdp_merge_set_send.argtypes = [P(CommData), P(MergeInfo), C.c_int, C.c_int, P(C.c_int),
                                        P(C.c_int)]
    lib.cfdp_merge_link_group.argtypes = [C.c_int, P(P(CommData)), P(P(MergeInfo))]
    lib.cfdp_merge_link_group.restype = None
    lib.cfdp_merge_scatter.argtypes = [P(MergeInfo), C.c_int, C.c_int, C.c_int, P(C.c_double),
                                       P(C.c_double)]
    lib.cfdp_merge_scatter.restype = None
    lib.cfdp_merge_info_free.argtypes = [P(MergeInfo)]
    lib.cfdp_merge_info_free.restype = None
    lib.cfdp_plan_default_opts.argtypes = [P(PlanOpts)]
    lib.cfdp_plan_default_opts.restype = None
    lib.cfdp_plan_build.argtypes = [P(SolverData), P(CommData), P(PlanOpts)]
    lib.cfdp_plan_build.restype = P(PlanStruct)
    lib.cfdp_plan_free.argtypes = [P(PlanStruct)]
    lib.cfdp_plan_free.restype = None
    lib.cfdp_algo_bytes_grad.argtypes = [C.c_long, C.c_long, C.c_long]
    lib.cfdp_algo_bytes_grad.restype = C.c_double
    lib.cfdp_algo_bytes_flux.argtypes = [C.c_long, C.c_long, C.c_long]
    lib.cfdp_algo_bytes_flux.restype = C.c_double
    lib.cfdp_fill_var.argtypes = [P(C.c_double), P(C.c_int), C.c_int, C.c_int, C.c_int, C.c_int,
                                  C.c_int]
    lib.cfdp_fill_var.restype = None
    lib.cfdp_nc_open.argtypes = [C.c_char_p]
    lib.cfdp_nc_close.argtypes = [C.c_int]
    lib.cfdp_nc_close.restype = None
    lib.get_nc_val.argtypes = [C.c_int, C.c_char_p]
    lib.get_nc_int.argtypes = [C.c_int, C.c_char_p, P(C.c_int)]
    lib.get_nc_int.restype = None
    lib.get_nc_double.argtypes = [C.c_int, C.c_char_p, P(C.c_double)]
    lib.get_nc_double.restype = None
    lib.compute_communication_tables.argtypes = [P(CommData)]
    lib.compute_communication_tables.restype = None
    lib.cfdp_group_link_raw.argtypes = [C.c_int, P(P(CommData))]
    lib.cfdp_group_link_raw.restype = None
    lib.cfdp_host_version.restype = C.c_char_p
    lib.cfdp_experiment_getenv.argtypes = [C.c_char_p]
    lib.cfdp_experiment_getenv.restype = C.c_char_p
    lib.cfdp_experiment_switches.restype = C.c_char_p
    lib.cfdp_experiments_active.argtypes = [C.c_char_p, C.c_size_t]


class ScaledCheck(C.Structure):
    """cfdp_scaled_check (cfdproxy_hip.h): the evidence of a scaled-field validation run"""
    _fields_ = [("iterations", C.c_int), ("flux_checks", C.c_int), ("mismatches", C.c_int), ("first_iteration", C.c_int),
                ("first_point", C.c_int), ("first_component", C.c_int), ("seen", C.c_double), ("expected", C.c_double),
                ("var_mismatches", C.c_int)]


def _declare_hip(lib: C.CDLL) -> None:
    P = C.POINTER
    vp = C.c_void_p
    lib.cfdp_gpu_device_count.restype = C.c_int
    lib.cfdp_gpu_last_error.restype = C.c_char_p
    lib.cfdp_gpu_create.argtypes = [C.c_int, P(vp)]
    lib.cfdp_gpu_destroy.argtypes = [vp]
    lib.cfdp_gpu_destroy.restype = None
    lib.cfdp_gpu_upload_plan.argtypes = [vp, P(PlanStruct)]
    lib.cfdp_gpu_bind_grad.argtypes = [vp, vp]
    lib.cfdp_gpu_bind_sendbuf.argtypes = [vp, vp]
    lib.cfdp_gpu_set_fusion.argtypes = [vp, C.c_int]
    lib.cfdp_gpu_bind_grad_alt.argtypes = [vp, vp]
    lib.cfdp_gpu_time_fused.argtypes = [vp, C.c_int, C.c_int, P(C.c_float)]
    lib.cfdp_gpu_time_fused_movement.argtypes = [vp, C.c_int, P(C.c_float)]
    for n in ("set_var", "set_grad", "set_flux", "get_grad", "get_flux"):
        getattr(lib, "cfdp_gpu_" + n).argtypes = [vp, P(C.c_double)]
    lib.cfdp_gpu_set_variant.argtypes = [vp, C.c_int, C.c_int]
    lib.cfdp_gpu_gradients.argtypes = [vp, C.c_int, vp]
    lib.cfdp_gpu_flux.argtypes = [vp, C.c_int, vp]
    lib.cfdp_gpu_pack.argtypes = [vp, vp]
    lib.cfdp_gpu_unpack.argtypes = [vp, vp, vp]
    lib.cfdp_gpu_sync.argtypes = [vp]
    lib.cfdp_gpu_stream.argtypes = [vp, C.c_int]
    lib.cfdp_gpu_stream.restype = vp
    lib.cfdp_gpu_npartners.argtypes = [vp]
    lib.cfdp_gpu_partner_rank.argtypes = [vp, C.c_int]
    lib.cfdp_gpu_send_ptr.argtypes = [vp, C.c_int, P(C.c_size_t)]
    lib.cfdp_gpu_send_ptr.restype = vp
    lib.cfdp_gpu_recv_ptr.argtypes = [vp, C.c_int, P(C.c_size_t)]
    lib.cfdp_gpu_recv_ptr.restype = vp
    lib.cfdp_gpu_grad_ptr.argtypes = [vp]
    lib.cfdp_gpu_grad_ptr.restype = vp
    lib.cfdp_gpu_var_ptr.argtypes = [vp]
    lib.cfdp_gpu_var_ptr.restype = vp
    lib.cfdp_gpu_iteration_group.argtypes = [P(vp), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.cfdp_gpu_rank_gradients.argtypes = [P(vp), C.c_int, C.c_int, C.c_int, C.c_int]
    lib.cfdp_gpu_rank_flux.argtypes = [P(vp), C.c_int, C.c_int, C.c_int, C.c_int]
    lib.cfdp_gpu_sync_group.argtypes = [P(vp), C.c_int]
    lib.cfdp_gpu_rank_gradients_launch.argtypes = [P(vp), C.c_int, C.c_int, C.c_int, C.c_int]
    lib.cfdp_gpu_rank_gradients_send.argtypes = [P(vp), C.c_int, C.c_int]
    lib.cfdp_gpu_enable_peer_access.argtypes = [P(vp), C.c_int, P(C.c_int)]
    lib.cfdp_gpu_step_pre.argtypes = [vp, C.c_int, C.c_int]
    lib.cfdp_gpu_step_post.argtypes = [vp, C.c_int, C.c_int]
    lib.cfdp_gpu_time_kernels.argtypes = [vp, C.c_int, C.c_int, P(C.c_float), P(C.c_float)]
    lib.cfdp_gpu_run_iterations.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, P(C.c_float)]
    lib.cfdp_gpu_prepare_iterations.argtypes = [vp, C.c_int, C.c_int, C.c_int]
    lib.cfdp_gpu_refresh_graphs.argtypes = [vp]
    lib.cfdp_gpu_scaled_check_begin.argtypes = [vp]
    lib.cfdp_gpu_scaled_check_end.argtypes = [vp, P(ScaledCheck)]
    lib.cfdp_rccl_load.argtypes = [C.c_char_p]
    lib.cfdp_rccl_unique_id.argtypes = [vp]
    lib.cfdp_gpu_rccl_init.argtypes = [vp, vp, C.c_int, C.c_int, P(C.c_int)]
    lib.cfdp_gpu_rccl_allow_self_exchange.argtypes = [vp, C.c_int]
    lib.cfdp_gpu_rccl_finalize.argtypes = [vp]
    lib.cfdp_gpu_step_rccl.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.cfdp_gpu_run_steps_rccl.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.cfdp_gpu_ipc_export.argtypes = [vp, vp, P(C.c_size_t)]
    lib.cfdp_gpu_ipc_connect.argtypes = [vp, C.c_int, vp, C.c_size_t, C.c_size_t, C.c_size_t]
    lib.cfdp_gpu_ipc_ready.argtypes = [vp]
    lib.cfdp_gpu_ipc_export_flags.argtypes = [vp, vp]
    lib.cfdp_gpu_ipc_connect_flags.argtypes = [vp, C.c_int, vp, C.c_size_t]
    lib.cfdp_gpu_ipc_mode.argtypes = [vp]
    lib.cfdp_gpu_ipc_graph_stats.argtypes = [vp, P(C.c_long), P(C.c_long), P(C.c_long)]
    lib.cfdp_gpu_ipc_configure.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.cfdp_gpu_rccl_nranks.argtypes = [vp]
    lib.cfdp_gpu_ipc_flag_offset.argtypes = [C.c_int]
    lib.cfdp_gpu_ipc_flag_offset.restype = C.c_size_t
    lib.cfdp_gpu_device.argtypes = [vp]
    lib.cfdp_gpu_device_bus_id.argtypes = [C.c_int, C.c_char_p, C.c_int]
    lib.cfdp_gpu_ipc_connect_loopback.argtypes = [vp, C.c_int]
    lib.cfdp_gpu_ipc_enable.argtypes = [vp, C.c_int]
    lib.cfdp_gpu_ipc_disconnect.argtypes = [vp]
    lib.cfdp_gpu_ipc_error.argtypes = [vp]
    lib.cfdp_ipc_set_wait_seconds.argtypes = [C.c_double]
    lib.cfdp_gpu_step_ipc.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.cfdp_gpu_run_steps_ipc.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.cfdp_gpu_time_schedule.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, P(C.c_float)]
    lib.cfdp_gpu_vcycle.argtypes = [P(vp), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, P(C.c_float)]
    lib.cfdp_gpu_counts.argtypes = [vp, P(C.c_int), P(C.c_int), P(C.c_int), P(C.c_int)]


def host_lib() -> C.CDLL:
    """C host side only (loader, generator, merger, tiler); usable without a GPU."""
    global _host_lib
    if _host_lib is None:
        _host_lib = _load("libcfdproxy_host.so")
        _declare_host(_host_lib)
    return _host_lib


def hip_lib() -> C.CDLL:
    """The C-ABI library with the HIP kernels.  Loading needs ROCm, running needs a GPU."""
    global _hip_lib
    if _hip_lib is None:
        _hip_lib = _load("libcfdproxy_hip.so")
        _declare_hip(_hip_lib)
    return _hip_lib


def _np_view(ptr, shape, dtype):
    n = int(np.prod(shape))
    if n == 0 or not ptr:
        return np.zeros(shape, dtype=dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,)).reshape(shape)


# --------------------------------------------------------------------------------- Domain
class Domain:
    """One partition: the reference's (solver_data, comm_data) pair with numpy views."""

    def __init__(self, owns: bool = True):
        self.sd = SolverData()
        self.cd = CommData()
        self._owns = owns
        self.merge_info = None  # set for merged partitions

    # numpy views onto the C arrays (no copies)
    @property
    def nown(self): return self.sd.nownpoints
    @property
    def nall(self): return self.sd.nallpoints
    @property
    def nfaces(self): return self.sd.nfaces
    @property
    def fpoint(self): return _np_view(self.sd.fpoint, (self.nfaces, 2), np.int32)
    @property
    def fnormal(self): return _np_view(self.sd.fnormal, (self.nfaces, 3), np.float64)
    @property
    def pvolume(self): return _np_view(self.sd.pvolume, (self.nall,), np.float64)
    @property
    def var(self): return _np_view(self.sd.var, (self.nall, NGRAD), np.float64)
    @property
    def grad(self): return _np_view(self.sd.grad, (self.nall, NGRAD, 3), np.float64)
    @property
    def psd_flux(self): return _np_view(self.sd.psd_flux, (self.nall, NFLUX), np.float64)

    @property
    def partners(self) -> List[int]:
        return [self.cd.commpartner[i] for i in range(self.cd.ncommdomains)] if self.cd.ndomains > 1 else []

    def sendindex(self, k: int) -> np.ndarray:
        n = self.cd.sendcount[k]
        if n == 0 or not self.cd.sendindex or not self.cd.sendindex[k]:
            return np.zeros(0, np.int32)
        return np.ctypeslib.as_array(self.cd.sendindex[k], shape=(n,)).copy()

    def recvindex(self, k: int) -> np.ndarray:
        n = self.cd.recvcount[k]
        if n == 0 or not self.cd.recvindex or not self.cd.recvindex[k]:
            return np.zeros(0, np.int32)
        return np.ctypeslib.as_array(self.cd.recvindex[k], shape=(n,)).copy()

    def send_points(self) -> np.ndarray:
        idx = [self.sendindex(k) for k in self.partners]
        return np.unique(np.concatenate(idx)) if idx else np.zeros(0, np.int32)

    def addpoint_owner(self): return _np_view(self.cd.addpoint_owner, (self.cd.naddpoints,), np.int32)
    def addpoint_id(self): return _np_view(self.cd.addpoint_id, (self.cd.naddpoints,), np.int32)

    def write(self, path: str, cdf_version: int = 1) -> None:
        rc = host_lib().cfdp_write_domain_file(path.encode(), C.byref(self.sd), C.byref(self.cd), cdf_version)
        if rc:
            raise IOError(f"writing {path} failed ({rc})")

    def free(self) -> None:
        if self._owns:
            lib = host_lib()
            lib.cfdp_free_solver_data(C.byref(self.sd))
            lib.cfdp_free_comm_data(C.byref(self.cd))
            if self.merge_info is not None:
                lib.cfdp_merge_info_free(self.merge_info)
                self.merge_info = None
            self._owns = False


_libc = C.CDLL(None)
_libc.malloc.restype = C.c_void_p
_libc.malloc.argtypes = [C.c_size_t]


def _c_copy(arr: np.ndarray, dtype, ctype):
    """copy a numpy array into malloc()ed memory (owned by the C struct, freed with free())"""
    a = np.ascontiguousarray(arr, dtype)
    p = _libc.malloc(max(a.nbytes, 1))
    C.memmove(p, a.ctypes.data, a.nbytes)
    return C.cast(p, C.POINTER(ctype))


def domain_from_arrays(fpoint, fnormal, pvolume, nown, var=None, ndomains=1, iproc=0,
                       addpoint_owner=None, addpoint_idx=None, commpartner=None, sendcount=None,
                       recvcount=None) -> Domain:
    """Build a (solver_data, comm_data) pair from arrays -- what read_solver_data() /
    read_communication_data() produce from a file (fields initialised to 1.0)."""
    d = Domain()
    nall, nf = len(pvolume), len(fpoint)
    d.sd.nfaces = d.sd.nallfaces = nf
    d.sd.nownpoints, d.sd.nallpoints, d.sd.ncolors, d.sd.niter = int(nown), nall, 1, 25
    d.sd.fpoint = _c_copy(fpoint, np.int32, C.c_int)
    d.sd.fnormal = _c_copy(fnormal, np.float64, C.c_double)
    d.sd.pvolume = _c_copy(pvolume, np.float64, C.c_double)
    d.sd.var = _c_copy(np.ones((nall, NGRAD)) if var is None else var, np.float64, C.c_double)
    d.sd.grad = _c_copy(np.ones((nall, NGRAD, 3)), np.float64, C.c_double)
    d.sd.psd_flux = _c_copy(np.ones((nall, NFLUX)), np.float64, C.c_double)
    d.cd.nProc, d.cd.iProc, d.cd.ndomains, d.cd.nownpoints = ndomains, iproc, ndomains, int(nown)
    if ndomains > 1:
        d.cd.naddpoints = nall - int(nown)
        d.cd.ncommdomains = len(commpartner)
        d.cd.addpoint_owner = _c_copy(addpoint_owner, np.int32, C.c_int)
        d.cd.addpoint_id = _c_copy(addpoint_idx, np.int32, C.c_int)
        d.cd.commpartner = _c_copy(commpartner, np.int32, C.c_int)
        d.cd.sendcount = _c_copy(sendcount, np.int32, C.c_int)
        d.cd.recvcount = _c_copy(recvcount, np.int32, C.c_int)
    return d


CONN_IRREGULAR = 62  # cfdproxy_host.h CFDP_CONN_IRREGULAR


def gen_params(nx, ny=None, nz=None, ndomains=1, connectivity=7, normals=1, volumes=1,
               ghost_faces=0, cdf_version=1, seed=20241, numbering=0, hubs=0) -> GenParams:
    ny = nx if ny is None else ny
    nz = nx if nz is None else nz
    return GenParams(nx, ny, nz, ndomains, connectivity, normals, volumes, ghost_faces, cdf_version, seed, numbering, hubs)


def gen_domain(gp: GenParams, domain: int) -> Domain:
    d = Domain()
    rc = host_lib().cfdp_gen_domain(C.byref(gp), domain, C.byref(d.sd), C.byref(d.cd))
    if rc:
        raise RuntimeError("generator failed")
    return d


def gen_global_ids(gp: GenParams, domain: int, nall: int) -> np.ndarray:
    gid = np.zeros(nall, np.int32)
    n = host_lib().cfdp_gen_global_ids(C.byref(gp), domain, gid.ctypes.data_as(C.POINTER(C.c_int)))
    assert n == nall, (n, nall)
    return gid


def write_mesh(gp: GenParams, prefix: str, lvl: int, domains: Optional[Sequence[int]] = None) -> None:
    """Write "<prefix>_domain_<d>_lvl_<lvl>" (reference file naming, hybrid.f6.c:58-62)."""
    lib = host_lib()
    for d in (range(gp.ndomains) if domains is None else domains):
        rc = lib.cfdp_gen_write_domain(C.byref(gp), d, prefix.encode(), lvl)
        if rc:
            raise IOError(f"writing domain {d} failed ({rc})")


def load_domain(prefix: str, domain: int, lvl: int) -> Domain:
    d = Domain()
    host_lib().cfdp_load_domain(prefix.encode(), domain, lvl, C.byref(d.sd), C.byref(d.cd))
    return d


def fill_var(dom: Domain, gid: Optional[np.ndarray], kind: int, nx=1, ny=1, nz=1) -> None:
    g = None if gid is None else np.ascontiguousarray(gid, np.int32)
    host_lib().cfdp_fill_var(dom.sd.var, None if g is None else g.ctypes.data_as(C.POINTER(C.c_int)),
                             dom.nall, kind, nx, ny, nz)


def rank_domains(r: int, N: int, G: int):
    f, c = C.c_int(), C.c_int()
    host_lib().cfdp_rank_domains(r, N, G, C.byref(f), C.byref(c))
    return f.value, c.value


def rank_domain_list(r: int, N: int, G: int) -> List[int]:
    """the domains of rank r under the active domain -> rank mapping (blocks, or set_domain_map)"""
    ids = (C.c_int * N)()
    n = host_lib().cfdp_rank_domain_list(r, N, G, ids)
    return list(ids[:n])


def set_domain_map(rank_of_domain, G: int = 0) -> None:
    """install (or, with None, remove) an explicit domain -> rank map for the merger"""
    if rank_of_domain is None:
        host_lib().cfdp_set_domain_map(None, 0, 0)
        return
    m = np.ascontiguousarray(rank_of_domain, np.int32)
    host_lib().cfdp_set_domain_map(m.ctypes.data_as(C.POINTER(C.c_int)), len(m), G or int(m.max()) + 1)


def cluster_domains(xadj, adj, wgt, G: int):
    """greedy graph growing on the domain graph -> (rank_of_domain, weight of the cut edges)"""
    xa, ad = np.ascontiguousarray(xadj, np.int32), np.ascontiguousarray(adj, np.int32)
    wg = None if wgt is None else np.ascontiguousarray(wgt, np.int32)
    ip = C.POINTER(C.c_int)
    out = np.empty(len(xa) - 1, np.int32)
    cut = host_lib().cfdp_cluster_domains(len(xa) - 1, G, xa.ctypes.data_as(ip), ad.ctypes.data_as(ip),
                                          None if wg is None else wg.ctypes.data_as(ip), out.ctypes.data_as(ip))
    return out, int(cut)


def domain_graph(prefix: str, lvl: int, N: int):
    """(xadj, adj, wgt) of the commpartner graph of N dualgrid files"""
    ip = C.POINTER(C.c_int)
    xa, ad, wg = ip(), ip(), ip()
    host_lib().cfdp_domain_graph(prefix.encode(), lvl, N, C.byref(xa), C.byref(ad), C.byref(wg))
    xadj = np.ctypeslib.as_array(xa, shape=(N + 1,)).copy()
    n = int(xadj[N])
    adj = np.ctypeslib.as_array(ad, shape=(max(n, 1),))[:n].copy()
    wgt = np.ctypeslib.as_array(wg, shape=(max(n, 1),))[:n].copy()
    for q in (xa, ad, wg):
        _libc.free(q)
    return xadj, adj, wgt


def merge_domains(doms: Sequence[Domain], domain_ids: Sequence[int], N: int, G: int, r: int) -> Domain:
    """N/G loaded domains -> the partition of GPU rank r (domain_merge.c)."""
    lib = host_lib()
    n = len(doms)
    sds = (SolverData * n)(*[d.sd for d in doms])
    cds = (CommData * n)(*[d.cd for d in doms])
    ids = (C.c_int * n)(*domain_ids)
    out = Domain()
    info = C.POINTER(MergeInfo)()
    rc = lib.cfdp_merge_domains(n, ids, sds, cds, N, G, r, C.byref(out.sd), C.byref(out.cd), C.byref(info))
    if rc:
        raise RuntimeError("merge failed")
    out.merge_info = info
    return out


def merge_requests(part: Domain):
    """{partner rank: (domains, idx)} -- which points this rank wants, in message order."""
    mi = part.merge_info.contents
    out = {}
    for i in range(mi.npartners):
        a, b = mi.want_off[i], mi.want_off[i + 1]
        out[mi.partner[i]] = (np.array(mi.ghost_domain[a:b], np.int32), np.array(mi.ghost_idx[a:b], np.int32))
    return out


def merge_set_send(part: Domain, s: int, domains: np.ndarray, idx: np.ndarray) -> None:
    d = np.ascontiguousarray(domains, np.int32)
    i = np.ascontiguousarray(idx, np.int32)
    ip = C.POINTER(C.c_int)
    host_lib().cfdp_merge_set_send(C.byref(part.cd), part.merge_info, s, len(d), d.ctypes.data_as(ip),
                                   i.ctypes.data_as(ip))


def merge_link_group(parts: Sequence[Domain]) -> None:
    """All ranks in this process: wire every rank's send lists (no communication needed)."""
    for s, ps in enumerate(parts):
        for r, (dom, idx) in merge_requests(ps).items():
            merge_set_send(parts[r], s, dom, idx)


def link_raw_group(doms: Sequence[Domain]) -> None:
    """Un-merged partitions, one per rank, all in this process: build recvindex
    (compute_communication_tables, comm_data.c:161-174) and sendindex (the MPI index exchange
    of comm_data.c:203-249, read directly from the partners' tables)."""
    lib = host_lib()
    for d in doms:
        lib.compute_communication_tables(C.byref(d.cd))
    arr = (C.POINTER(CommData) * len(doms))(*[C.pointer(d.cd) for d in doms])
    lib.cfdp_group_link_raw(len(doms), arr)


def merge_scatter(part: Domain, dl: int, npoints_d: int, field: np.ndarray) -> np.ndarray:
    """merged per-point field -> file numbering of local domain dl (ghost rows included)."""
    f = np.ascontiguousarray(field, np.float64)
    rowlen = int(np.prod(f.shape[1:])) if f.ndim > 1 else 1
    out = np.zeros((npoints_d,) + f.shape[1:], np.float64)
    dp = C.POINTER(C.c_double)
    host_lib().cfdp_merge_scatter(part.merge_info, dl, npoints_d, rowlen, f.ctypes.data_as(dp),
                                  out.ctypes.data_as(dp))
    return out


def stored_rows(rows: np.ndarray) -> np.ndarray:
    """gradient rows [n][21] (or [n][7][3]) as the DEVICE keeps and sends them: the first 10 doubles of a row -- the 3x3
    velocity-gradient block g0..g8 and g9 -- as [g0 g4 g8 | g1+g3 g2+g6 g5+g7 | g3 g6 g7 | g9] (gg_a_encode,
    csrc/gg_kernels.h: the first 48 bytes are all the flux loop stages); the other 11 as they are"""
    r = np.ascontiguousarray(rows, dtype=np.float64).reshape(-1, 21)
    e = r.copy()
    e[:, 0], e[:, 1], e[:, 2] = r[:, 0], r[:, 4], r[:, 8]
    e[:, 3], e[:, 4], e[:, 5] = r[:, 1] + r[:, 3], r[:, 2] + r[:, 6], r[:, 5] + r[:, 7]
    e[:, 6], e[:, 7], e[:, 8] = r[:, 3], r[:, 6], r[:, 7]
    return e


def handed_out_rows(stored: np.ndarray) -> np.ndarray:
    """the inverse at the host boundary (gg_a_decode): the three upper off-diagonals come back as (sum) - (lower one),
    within one rounding of the sum of the two"""
    e = np.ascontiguousarray(stored, dtype=np.float64).reshape(-1, 21)
    r = e.copy()
    r[:, 0], r[:, 4], r[:, 8] = e[:, 0], e[:, 1], e[:, 2]
    r[:, 3], r[:, 6], r[:, 7] = e[:, 6], e[:, 7], e[:, 8]
    r[:, 1], r[:, 2], r[:, 5] = e[:, 3] - e[:, 6], e[:, 4] - e[:, 7], e[:, 5] - e[:, 8]
    return r


def kernel_forms() -> str:
    """the face-loop kernel forms this thread has launched since the last call (cfdp_gpu_kernel_forms); the first call
    switches the log on and returns ''"""
    lib = hip_lib()
    lib.cfdp_gpu_kernel_forms.argtypes = [C.c_char_p, C.c_size_t]
    buf = C.create_string_buffer(1024)
    lib.cfdp_gpu_kernel_forms(buf, 1024)
    return buf.value.decode()


def kernel_forms_off() -> None:
    """switch the log of kernel_forms() off again"""
    lib = hip_lib()
    lib.cfdp_gpu_kernel_forms.argtypes = [C.c_char_p, C.c_size_t]
    lib.cfdp_gpu_kernel_forms(None, 0)


def device_bus_id(device: int) -> str:
    """PCI bus id of a visible device: the same string in every process that sees the same physical device"""
    buf = C.create_string_buffer(64)
    if hip_lib().cfdp_gpu_device_bus_id(int(device), buf, len(buf)):
        raise RuntimeError(hip_lib().cfdp_gpu_last_error().decode())
    return buf.value.decode()


def experiment_switches() -> list:
    """the environment variables host/experiments.c gates behind CFDP_EXPERIMENTS=1 (not the product path)"""
    return host_lib().cfdp_experiment_switches().decode().split()


def experiments_active() -> list:
    """["NAME=value", ...] of the experiment switches that are set WITH the master key: a benchmark must not report
    a run made under any of them (bench.py refuses)"""
    buf = C.create_string_buffer(1024)
    n = host_lib().cfdp_experiments_active(buf, len(buf))
    return buf.value.decode().split() if n else []


def algo_bytes_grad(nfaces: int, nown: int, nadd: int) -> float:
    return host_lib().cfdp_algo_bytes_grad(nfaces, nown, nadd)


def algo_bytes_flux(nfaces: int, nown: int, nadd: int) -> float:
    return host_lib().cfdp_algo_bytes_flux(nfaces, nown, nadd)


# ----------------------------------------------------------------------------------- Plan
class Plan:
    """The GPU tiling of one partition (host/tiling.c) -- the init_threads() analogue."""

    def __init__(self, dom: Domain, tile_points: int = 0, boundary_first: bool = True, device_stages: int = 0,
                 device: int = 0):
        """device_stages: 0 = all on the host; bit 0 = the point->face CSR, bit 1 = the tile blobs built by HIP
        kernels on `device` (cfdp_plan_build_gpu; needs a GPU, no fallback); stage_seconds then holds their times"""
        lib = host_lib()
        o = PlanOpts()
        lib.cfdp_plan_default_opts(C.byref(o))
        if tile_points:
            o.tile_points = tile_points
        o.boundary_first = 1 if boundary_first else 0
        self.stage_seconds = None
        if device_stages:
            hl = hip_lib()
            hl.cfdp_plan_build_gpu.argtypes = [C.POINTER(SolverData), C.POINTER(CommData), C.POINTER(PlanOpts), C.c_int, C.c_int,
                                               C.POINTER(C.POINTER(PlanStruct)), C.POINTER(C.c_double)]
            ptr = C.POINTER(PlanStruct)()
            secs = (C.c_double * 2)()
            if hl.cfdp_plan_build_gpu(C.byref(dom.sd), C.byref(dom.cd), C.byref(o), device, device_stages, C.byref(ptr), secs):
                raise GpuError(hl.cfdp_gpu_last_error().decode())
            self.ptr = ptr
            self.stage_seconds = (secs[0], secs[1])
        else:
            self.ptr = lib.cfdp_plan_build(C.byref(dom.sd), C.byref(dom.cd), C.byref(o))
        if not self.ptr:
            raise RuntimeError("plan build failed")
        self.p = self.ptr.contents

    def __getattr__(self, name):
        return getattr(self.p, name)

    @property
    def new2old(self): return np.ctypeslib.as_array(self.p.new2old, shape=(self.p.nall,))
    @property
    def old2new(self): return np.ctypeslib.as_array(self.p.old2new, shape=(self.p.nall,))

    def tile(self, t: int) -> TileDesc:
        return self.p.tiles[t]

    def tile_arrays(self, t: int):
        """(normals[E,3], inc[I], ioff[T+1], halo[H]) of tile t, decoded from the blob."""
        td = self.p.tiles[t]
        blob = np.ctypeslib.as_array(self.p.blob, shape=(self.p.blob_bytes,))
        b0 = td.blob_off * 16
        plane = (td.nfaces * 8 + 15) & ~15   # one padded plane per normal component (SoA)
        fnb = 3 * plane
        incb = (td.ninc * 4 + 15) & ~15
        fn = np.stack([blob[b0 + c * plane: b0 + c * plane + td.nfaces * 8].view(np.float64) for c in range(3)], 1)
        inc = blob[b0 + fnb:b0 + fnb + td.ninc * 4].view(np.uint32)
        ioff = blob[b0 + fnb + incb:b0 + fnb + incb + (td.npts + 1) * 4].view(np.uint32) & np.uint32(0xFFFFFF)  # (bits 24-31: chunks of a long list - 1)
        halo = np.ctypeslib.as_array(self.p.halo_idx, shape=(max(self.p.nhalo_total, 1),))[
            td.halo_off:td.halo_off + td.nhalo]
        return fn, inc, ioff, halo

    def free(self):
        if self.ptr:
            host_lib().cfdp_plan_free(self.ptr)
            self.ptr = None


# --------------------------------------------------------------------------- GpuPartition
class GpuError(RuntimeError):
    pass


class GpuPartition:
    """One partition resident on one MI355X, driven through the C ABI (cfdproxy_hip.h)."""

    def __init__(self, dom: Domain, device: int = 0, tile_points: int = 0, boundary_first: bool = True,
                 grad_lanes: int = 0, flux_lanes: int = 0):
        self.lib = hip_lib()
        if self.lib.cfdp_gpu_device_count() <= 0:
            raise GpuError("no HIP device: the CFD-Proxy hot path has no CPU fallback")
        self.dom = dom
        self.h = C.c_void_p()
        self._ck(self.lib.cfdp_gpu_create(device, C.byref(self.h)))
        # which heavy stages of the plan run as HIP kernels (bit 0 CSR, bit 1 tile blobs): both by default -- the
        # plans are bit-identical and the device builds them 6-12x faster; CFDP_PLAN_DEVICE=0 keeps the host stages
        plan = Plan(dom, tile_points, boundary_first, device_stages=int(os.environ.get("CFDP_PLAN_DEVICE", "3")), device=device)
        self.stats = dict(ntiles=plan.ntiles, nbtiles=plan.nbtiles, nfaces_used=plan.nfaces_used,
                          nfaces_dup=plan.nfaces_dup, ninc=plan.ninc_total, lds_grad=plan.lds_grad,
                          lds_flux=plan.lds_flux, blob_bytes=plan.blob_bytes, nhalo=plan.nhalo_total,
                          tile_points=plan.tile_points, plan_stage_seconds=plan.stage_seconds,
                          groups=[(plan.group_begin[k], plan.group_begin[k + 1], plan.group_class[k]) for k in range(plan.ngroups)])
        try:
            # (lanes first: the upload refuses tiles whose points x lanes exceed a workgroup)
            self._ck(self.lib.cfdp_gpu_set_variant(self.h, grad_lanes, flux_lanes))
            self._ck(self.lib.cfdp_gpu_upload_plan(self.h, plan.ptr))
        finally:
            plan.free()
        self.push_fields()

    def _ck(self, rc: int) -> None:
        if rc:
            raise GpuError(self.lib.cfdp_gpu_last_error().decode())

    @staticmethod
    def _dp(a: np.ndarray):
        return a.ctypes.data_as(C.POINTER(C.c_double))

    def push_fields(self) -> None:
        self._ck(self.lib.cfdp_gpu_set_var(self.h, self.dom.sd.var))
        self._ck(self.lib.cfdp_gpu_set_grad(self.h, self.dom.sd.grad))
        self._ck(self.lib.cfdp_gpu_set_flux(self.h, self.dom.sd.psd_flux))

    def pull_fields(self) -> None:
        self._ck(self.lib.cfdp_gpu_get_grad(self.h, self.dom.sd.grad))
        self._ck(self.lib.cfdp_gpu_get_flux(self.h, self.dom.sd.psd_flux))

    def set_variant(self, grad_lanes: int, flux_lanes: int = 0) -> None:
        self._ck(self.lib.cfdp_gpu_set_variant(self.h, grad_lanes, flux_lanes))


    def gradients(self, which: int = TILES_ALL, stream: int = 0) -> None:
        self._ck(self.lib.cfdp_gpu_gradients(self.h, which, C.c_void_p(stream)))

    def flux(self, mode: int = FLUX_CONSISTENT, stream: int = 0) -> None:
        self._ck(self.lib.cfdp_gpu_flux(self.h, mode, C.c_void_p(stream)))

    def pack(self, stream: int = 0) -> None:
        self._ck(self.lib.cfdp_gpu_pack(self.h, C.c_void_p(stream)))

    def unpack(self, dev_ptr: int, stream: int = 0) -> None:
        self._ck(self.lib.cfdp_gpu_unpack(self.h, C.c_void_p(dev_ptr), C.c_void_p(stream)))

    def step_pre(self, with_exchange: bool, overlap: bool) -> None:
        self._ck(self.lib.cfdp_gpu_step_pre(self.h, int(with_exchange), int(overlap)))

    def step_post(self, with_flux: bool = True, flux_mode: int = FLUX_CONSISTENT) -> None:
        self._ck(self.lib.cfdp_gpu_step_post(self.h, int(with_flux), flux_mode))

    def sync(self) -> None:
        self._ck(self.lib.cfdp_gpu_sync(self.h))

    def stream(self, which: int = 0) -> int:
        return self.lib.cfdp_gpu_stream(self.h, which) or 0

    def counts(self):
        a, b, c, d = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        self._ck(self.lib.cfdp_gpu_counts(self.h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        return dict(nown=a.value, nall=b.value, nsend=c.value, nrecv=d.value)

    def partners(self) -> List[int]:
        return [self.lib.cfdp_gpu_partner_rank(self.h, s) for s in range(self.lib.cfdp_gpu_npartners(self.h))]

    def send_slice(self, s: int):
        n = C.c_size_t()
        p = self.lib.cfdp_gpu_send_ptr(self.h, s, C.byref(n))
        return p or 0, n.value

    def recv_slice(self, s: int):
        n = C.c_size_t()
        p = self.lib.cfdp_gpu_recv_ptr(self.h, s, C.byref(n))
        return p or 0, n.value

    def bind_grad(self, dev_ptr: int) -> None:
        self._ck(self.lib.cfdp_gpu_bind_grad(self.h, C.c_void_p(dev_ptr)))

    def bind_sendbuf(self, dev_ptr: int) -> None:
        self._ck(self.lib.cfdp_gpu_bind_sendbuf(self.h, C.c_void_p(dev_ptr)))

    # ---- the exchange issued from the C library (RCCL resolved at run time)
    @staticmethod
    def rccl_unique_id(libpath: str = "") -> bytes:
        lib = hip_lib()
        if lib.cfdp_rccl_load(libpath.encode()):
            raise GpuError(lib.cfdp_gpu_last_error().decode())
        buf = C.create_string_buffer(128)
        if lib.cfdp_rccl_unique_id(buf):
            raise GpuError(lib.cfdp_gpu_last_error().decode())
        return buf.raw

    def rccl_init(self, unique_id: bytes, nranks: int, rank: int, rank_of_partner=None, libpath: str = "",
                  self_exchange: bool = False) -> None:
        """self_exchange: measurements / plumbing tests only -- a communicator of ONE rank exchanges with itself
        (cfdp_gpu_rccl_allow_self_exchange); refused otherwise"""
        self._ck(self.lib.cfdp_rccl_load(libpath.encode()))
        self._ck(self.lib.cfdp_gpu_rccl_allow_self_exchange(self.h, 1 if self_exchange else 0))
        rp = None
        if rank_of_partner is not None:
            rp = (C.c_int * len(rank_of_partner))(*rank_of_partner)
        self._ck(self.lib.cfdp_gpu_rccl_init(self.h, C.create_string_buffer(unique_id, 128), nranks, rank, rp))

    def rccl_nranks(self) -> int:
        """what ncclCommCount says about this context's communicator (0: none)"""
        return int(self.lib.cfdp_gpu_rccl_nranks(self.h))

    def step_rccl(self, with_exchange=True, overlap=True, with_flux=True, flux_mode: int = FLUX_CONSISTENT) -> None:
        self._ck(self.lib.cfdp_gpu_step_rccl(self.h, int(with_exchange), int(overlap), int(with_flux), flux_mode))

    def run_steps_rccl(self, steps: int, with_exchange=True, overlap=True, with_flux=True,
                       flux_mode: int = FLUX_CONSISTENT) -> None:
        self._ck(self.lib.cfdp_gpu_run_steps_rccl(self.h, steps, int(with_exchange), int(overlap), int(with_flux),
                                                  flux_mode))

    # ---- xGMI write + notify between processes (HIP IPC)
    def ipc_export(self):
        """(64-byte IPC handle of this rank's block, bytes of one landing arena)"""
        buf, n = C.create_string_buffer(64), C.c_size_t()
        self._ck(self.lib.cfdp_gpu_ipc_export(self.h, buf, C.byref(n)))
        return buf.raw, n.value

    def ipc_connect(self, slot: int, handle: bytes, land_off0: int, land_off1: int, flag_off: int) -> None:
        self._ck(self.lib.cfdp_gpu_ipc_connect(self.h, slot, C.create_string_buffer(handle, 64), land_off0, land_off1,
                                               flag_off))

    def ipc_export_flags(self) -> bytes:
        """handle of the block holding this rank's flag words (the main block's again unless CFDP_IPC_MODE=split)"""
        buf = C.create_string_buffer(64)
        self._ck(self.lib.cfdp_gpu_ipc_export_flags(self.h, buf))
        return buf.raw

    def ipc_connect_flags(self, slot: int, flags_handle: bytes, flag_off: int) -> None:
        self._ck(self.lib.cfdp_gpu_ipc_connect_flags(self.h, slot, C.create_string_buffer(flags_handle, 64), flag_off))

    def ipc_mode(self) -> dict:
        m = self.lib.cfdp_gpu_ipc_mode(self.h)
        if m < 0:
            return {}
        return {"push": "in the fused pass" if m & 1 else ("pack kernel + one copy per partner slice (copy-engine put)" if m & 64 else "push kernel"), "wait": "in the fused pass" if m & 2 else "wait kernel",
                "notify": "per partner" if m & 4 else "all partners by the last boundary tile",
                "notify_by": "counters (fire-and-forget atomic adds)" if m & 8 else "flags",
                "memory": ("coarse-grained", "fine-grained", "split: fine-grained flags, coarse-grained arenas")[(m >> 4) & 3]}

    IPC_MODES = {"coarse": 0, "fine": 1, "split": 2}

    def ipc_configure(self, memory_mode=None, wait_inkernel=None, notify=None, push_inkernel=None) -> None:
        """cfdp_gpu_ipc_configure: memory_mode "coarse" | "fine" | "split", wait_inkernel bool, notify "counter" | "flag",
        push_inkernel bool (False: push / notify / wait as kernels of their own); None = what the environment says, else
        the library default.  Takes effect at the next ipc_export / ipc_ready"""
        self._ck(self.lib.cfdp_gpu_ipc_configure(
            self.h, -1 if memory_mode is None else self.IPC_MODES[memory_mode],
            -1 if wait_inkernel is None else int(bool(wait_inkernel)),
            -1 if notify is None else {"counter": 1, "flag": 0}[notify],
            -1 if push_inkernel is None else (2 if push_inkernel == "put" else int(bool(push_inkernel)))))

    def ipc_graph_stats(self) -> dict:
        """steps of run_steps_ipc replayed from hipGraphs / launched from the streams, captures abandoned"""
        a, b, c = C.c_long(), C.c_long(), C.c_long()
        self._ck(self.lib.cfdp_gpu_ipc_graph_stats(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return {"steps_replayed": a.value, "steps_streamed": b.value, "captures_failed": c.value}

    def ipc_ready(self) -> None:
        self._ck(self.lib.cfdp_gpu_ipc_ready(self.h))

    def ipc_enable(self, on: bool) -> None:
        self._ck(self.lib.cfdp_gpu_ipc_enable(self.h, int(on)))

    def ipc_disconnect(self) -> None:
        self._ck(self.lib.cfdp_gpu_ipc_disconnect(self.h))

    def ipc_error(self) -> int:
        return self.lib.cfdp_gpu_ipc_error(self.h)

    def step_ipc(self, with_exchange=True, overlap=True, with_flux=True, flux_mode: int = FLUX_CONSISTENT) -> None:
        self._ck(self.lib.cfdp_gpu_step_ipc(self.h, int(with_exchange), int(overlap), int(with_flux), flux_mode))

    def run_steps_ipc(self, steps: int, with_exchange=True, overlap=True, with_flux=True,
                      flux_mode: int = FLUX_CONSISTENT, use_graph: bool = True) -> None:
        self._ck(self.lib.cfdp_gpu_run_steps_ipc(self.h, steps, int(with_exchange), int(overlap), int(with_flux),
                                                 flux_mode, int(use_graph)))

    def time_schedule(self, steps: int, with_exchange: bool, overlap: bool, use_graph: bool) -> float:
        ms = C.c_float()
        self._ck(self.lib.cfdp_gpu_time_schedule(self.h, steps, int(with_exchange), int(overlap), int(use_graph), C.byref(ms)))
        return ms.value

    def set_fusion(self, on: bool) -> None:
        """defer each iteration's flux into the pass that computes the next gradients"""
        self._ck(self.lib.cfdp_gpu_set_fusion(self.h, int(on)))

    def bind_grad_alt(self, dev_ptr: int) -> None:
        self._ck(self.lib.cfdp_gpu_bind_grad_alt(self.h, C.c_void_p(dev_ptr)))

    def grad_ptr(self) -> int:
        return self.lib.cfdp_gpu_grad_ptr(self.h) or 0

    def time_fused(self, iters: int, flux_mode: int = FLUX_CONSISTENT) -> float:
        ms = C.c_float()
        self._ck(self.lib.cfdp_gpu_time_fused(self.h, iters, flux_mode, C.byref(ms)))
        return ms.value

    def time_fused_movement(self, iters: int) -> float:
        """milliseconds per pass of the fused pass's data movement alone (every load, every store, no face loop)"""
        ms = C.c_float()
        self._ck(self.lib.cfdp_gpu_time_fused_movement(self.h, iters, C.byref(ms)))
        return ms.value

    def time_kernels(self, iters: int, flux_mode: int = FLUX_CONSISTENT):
        g, f = C.c_float(), C.c_float()
        self._ck(self.lib.cfdp_gpu_time_kernels(self.h, iters, flux_mode, C.byref(g), C.byref(f)))
        return g.value, f.value

    def run_iterations(self, iters: int, with_flux: bool = True, flux_mode: int = FLUX_CONSISTENT,
                       use_graph: bool = True, device_time: bool = True) -> float:
        """returns the device time of the run in ms (HIP events around it); device_time = False: no event pair -- for a
        caller that times the run itself -- and 0.0 comes back"""
        if not device_time:
            self._ck(self.lib.cfdp_gpu_run_iterations(self.h, iters, int(with_flux), flux_mode, int(use_graph), None))
            return 0.0
        ms = C.c_float()
        self._ck(self.lib.cfdp_gpu_run_iterations(self.h, iters, int(with_flux), flux_mode, int(use_graph),
                                                  C.byref(ms)))
        return ms.value

    def prepare_iterations(self, iters: int, with_flux: bool = True, flux_mode: int = FLUX_CONSISTENT) -> None:
        """capture the hipGraphs run_iterations(iters) replays, without executing anything"""
        self._ck(self.lib.cfdp_gpu_prepare_iterations(self.h, iters, int(with_flux), flux_mode))

    def refresh_graphs(self) -> None:
        """instantiate every cached hipGraph again (in front of a short timed run on an idle device): cfdp_gpu_refresh_graphs"""
        self._ck(self.lib.cfdp_gpu_refresh_graphs(self.h))

    def scaled_check_begin(self) -> None:
        """the flux held now becomes the reference; from here every step ends with the validation kernel (compare the
        flux with reference * 2^e, then var *= 2, 2, 1/4, ...): cfdp_gpu_scaled_check_begin"""
        self._ck(self.lib.cfdp_gpu_scaled_check_begin(self.h))

    def scaled_check_end(self) -> dict:
        r = ScaledCheck()
        self._ck(self.lib.cfdp_gpu_scaled_check_end(self.h, C.byref(r)))
        return {k: getattr(r, k) for k, _ in ScaledCheck._fields_}

    def close(self) -> None:
        if self.h:
            self.lib.cfdp_gpu_destroy(self.h)
            self.h = C.c_void_p()


def group_iteration(parts: Sequence[GpuPartition], with_exchange=True, overlap=True, with_flux=True,
                    flux_mode=FLUX_CONSISTENT) -> None:
    """One iteration of G in-process ranks (peer copies between their devices)."""
    lib = hip_lib()
    arr = (C.c_void_p * len(parts))(*[p.h for p in parts])
    rc = lib.cfdp_gpu_iteration_group(arr, len(parts), int(with_exchange), int(overlap), int(with_flux), flux_mode)
    if rc:
        raise GpuError(lib.cfdp_gpu_last_error().decode())


def group_iterations_threaded(parts: Sequence[GpuPartition], iters: int, with_exchange=True, overlap=True, with_flux=True,
                              flux_mode=FLUX_CONSISTENT) -> None:
    """`iters` iterations of G in-process ranks driven by G host threads, thread r enqueuing rank r's work ("thread t
    drives device t"): phase 1 in its two parts with a barrier behind each (cfdp_gpu_rank_gradients_launch /
    _send), then the flux -- what the drop-in layer's test_solver does with pthreads"""
    import threading
    lib = hip_lib()
    G = len(parts)
    arr = (C.c_void_p * G)(*[p.h for p in parts])
    bar = threading.Barrier(G)
    errors: List[str] = []

    def body(r: int) -> None:
        try:
            for _ in range(iters):
                rc = lib.cfdp_gpu_rank_gradients_launch(arr, G, r, int(with_exchange), int(overlap))
                if rc:
                    errors.append(lib.cfdp_gpu_last_error().decode())
                bar.wait()
                if lib.cfdp_gpu_rank_gradients_send(arr, G, r):
                    errors.append(lib.cfdp_gpu_last_error().decode())
                bar.wait()
                if lib.cfdp_gpu_rank_flux(arr, G, r, int(with_flux), flux_mode):
                    errors.append(lib.cfdp_gpu_last_error().decode())
        except threading.BrokenBarrierError:
            pass
        except Exception as e:  # never leave the other threads at the barrier
            errors.append(repr(e))
            bar.abort()

    threads = [threading.Thread(target=body, args=(r,)) for r in range(G)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise GpuError("; ".join(errors[:3]))


def enable_peer_access(parts: Sequence[GpuPartition]) -> int:
    """hipDeviceEnablePeerAccess between the devices of the ranks; returns the ordered device pairs enabled"""
    lib = hip_lib()
    arr = (C.c_void_p * len(parts))(*[p.h for p in parts])
    n = C.c_int()
    if lib.cfdp_gpu_enable_peer_access(arr, len(parts), C.byref(n)):
        raise GpuError(lib.cfdp_gpu_last_error().decode())
    return n.value


def vcycle(levels: Sequence[GpuPartition], sweeps: int = 3, cycles: int = 1, flux_mode: int = FLUX_CONSISTENT,
           use_graph: bool = True) -> float:
    """`cycles` multigrid V cycles over `levels` (finest first; one partition per level, one
    device): `sweeps` iterations per level down and up.  Milliseconds per cycle."""
    lib = hip_lib()
    arr = (C.c_void_p * len(levels))(*[p.h for p in levels])
    ms = C.c_float()
    if lib.cfdp_gpu_vcycle(arr, len(levels), sweeps, cycles, flux_mode, int(use_graph), C.byref(ms)):
        raise GpuError(lib.cfdp_gpu_last_error().decode())
    return ms.value


def group_sync(parts: Sequence[GpuPartition]) -> None:
    for p in parts:
        p.sync()
