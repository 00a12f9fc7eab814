"""ctypes binding of oracle/libcpu_ref.so (cpu_ref.c) + an independent numpy statement.

ORACLE = TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's
`cpu_baseline` leg may import this module; the product (cfd-proxy_amd/) never does.

Parity pin: cpu_ref.c is validated against the COMPILED reference (oracle/_ref/ref_dump_raw,
built by oracle/Makefile from /root/reference/src + our raw-array driver, no product code on
its link line) through tests/golden/*.npz (tests/test_oracle_golden.py).  The reference has
no tests or golden vectors of its own.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def build() -> None:
    r = subprocess.run(["make", "-C", _HERE, "all"], capture_output=True, text=True)
    if r.returncode:
        raise RuntimeError("building the oracle failed:\n" + r.stdout[-2000:] + r.stderr[-2000:])


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "libcpu_ref.so")
        if not os.path.exists(path):
            build()
        _lib = C.CDLL(path)
        vp = C.c_void_p
        _lib.oracle_init_threads.restype = vp
        _lib.oracle_init_threads.argtypes = [C.c_int, vp, vp, C.c_int, C.c_int, vp, C.c_int, C.c_int, vp]
        _lib.oracle_free.argtypes = [vp]
        _lib.oracle_free.restype = None
        _lib.oracle_check_invariants.argtypes = [vp]
        _lib.oracle_total_colours.argtypes = [vp]
        _lib.oracle_total_thread_faces.argtypes = [vp]
        _lib.oracle_total_thread_faces.restype = C.c_long
        _lib.oracle_gradients.argtypes = [vp, vp, vp]
        _lib.oracle_gradients.restype = None
        _lib.oracle_flux.argtypes = [vp, vp, vp, C.c_int]
        _lib.oracle_flux.restype = None
        _lib.oracle_pack.argtypes = [vp, C.c_int, vp, C.c_int, vp]
        _lib.oracle_pack.restype = None
        _lib.oracle_unpack.argtypes = [vp, C.c_int, vp, C.c_int, vp]
        _lib.oracle_unpack.restype = None
        _lib.oracle_timed_iterations.argtypes = [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int]
        _lib.oracle_timed_iterations.restype = C.c_double
    return _lib


def ref_dump_path() -> str:
    return os.path.join(_HERE, "_ref", "ref_dump_raw")


def write_raw_domain(prefix, domain, fpoint, fnormal, pvolume, nown, var=None, ndomains=1, commpartner=None,
                     sendcount=None, recvcount=None, addpoint_owner=None, addpoint_id=None) -> str:
    """One domain as the raw array file oracle/ref_dump_raw.c reads (layout in its header): the
    arrays the reference's loaders would have taken from the NetCDF file, nothing else."""
    fpoint = np.ascontiguousarray(fpoint, np.int32)
    nall = len(pvolume)
    multi = ndomains > 1
    nadd = nall - int(nown) if multi else 0
    ncomm = len(commpartner) if multi else 0
    path = f"{prefix}_{domain}.raw"
    with open(path, "wb") as fp:
        np.array([0x43464450, len(fpoint), int(nown), nall, ndomains, nadd, ncomm, int(var is not None)],
                 np.int32).tofile(fp)
        fpoint.tofile(fp)
        np.ascontiguousarray(fnormal, np.float64).tofile(fp)
        np.ascontiguousarray(pvolume, np.float64).tofile(fp)
        if var is not None:
            v = np.ascontiguousarray(var, np.float64)
            assert v.shape == (nall, 7)
            v.tofile(fp)
        if multi:
            for a, n in ((commpartner, ncomm), (sendcount, ndomains), (recvcount, ndomains),
                         (addpoint_owner, nadd), (addpoint_id, nadd)):
                a = np.ascontiguousarray(a, np.int32)
                assert len(a) == n, (len(a), n)
                a.tofile(fp)
    return path


class CpuRef:
    """The reference's algorithm on the CPU: thread domains, face classes, colours of <=96
    faces, first/last lists (reference src/rangelist.c:320-764, src/points_of_color.c)."""

    def __init__(self, fpoint, fnormal, pvolume, nown, nthreads=1, sendpoints=None):
        self.fpoint = np.ascontiguousarray(fpoint, np.int32)
        self.fnormal = np.ascontiguousarray(fnormal, np.float64)
        self.pvolume = np.ascontiguousarray(pvolume, np.float64)
        self.nown, self.nall = int(nown), len(self.pvolume)
        self.nthreads = int(nthreads)
        sp = np.ascontiguousarray(sendpoints if sendpoints is not None else [], np.int32)
        self.h = lib().oracle_init_threads(len(self.fpoint), self.fpoint.ctypes.data, self.fnormal.ctypes.data,
                                           self.nown, self.nall, self.pvolume.ctypes.data, self.nthreads,
                                           len(sp), sp.ctypes.data if len(sp) else None)

    def check_invariants(self) -> int:
        return lib().oracle_check_invariants(self.h)

    def gradients(self, var, grad_init=None):
        var = np.ascontiguousarray(var, np.float64)
        g = np.ones((self.nall, 7, 3)) if grad_init is None else np.array(grad_init, np.float64, copy=True)
        lib().oracle_gradients(self.h, var.ctypes.data, g.ctypes.data)
        return g

    def flux(self, grad, mode=0, flux_init=None):
        grad = np.ascontiguousarray(grad, np.float64)
        f = np.ones((self.nall, 3)) if flux_init is None else np.array(flux_init, np.float64, copy=True)
        lib().oracle_flux(self.h, grad.ctypes.data, f.ctypes.data, mode)
        return f

    def timed(self, var, niter=25, with_flux=True, flux_mode=0) -> float:
        var = np.ascontiguousarray(var, np.float64)
        g = np.ones((self.nall, 7, 3))
        f = np.ones((self.nall, 3))
        return lib().oracle_timed_iterations(self.h, var.ctypes.data, g.ctypes.data, f.ctypes.data, niter,
                                             int(with_flux), flux_mode)

    def close(self):
        if self.h:
            lib().oracle_free(self.h)
            self.h = None


def pack(sendindex, data):
    """exchange_dbl_copy_in (reference src/threads.c:791-813): rows of `data` -> message."""
    idx = np.ascontiguousarray(sendindex, np.int32)
    data = np.ascontiguousarray(data, np.float64)
    dim2 = int(np.prod(data.shape[1:]))
    out = np.zeros((len(idx), dim2))
    lib().oracle_pack(idx.ctypes.data, len(idx), data.ctypes.data, dim2, out.ctypes.data)
    return out


def unpack(recvindex, data, msg):
    """exchange_dbl_copy_out (reference src/threads.c:816-839): message -> rows of `data`."""
    idx = np.ascontiguousarray(recvindex, np.int32)
    msg = np.ascontiguousarray(msg, np.float64)
    dim2 = int(np.prod(data.shape[1:]))
    assert data.flags.c_contiguous
    lib().oracle_unpack(idx.ctypes.data, len(idx), data.ctypes.data, dim2, msg.ctypes.data)


# --------------------------------------------------------------- independent numpy statement
def np_gradients(fpoint, fnormal, pvolume, var, nown):
    """grad[p] = 1/V_p * (sum_{p0(f)=p} n_f*avg - sum_{p1(f)=p} n_f*avg)  (SURVEY.md section 2.3);
    rows of ghost points are returned as NaN (the kernel never writes them)."""
    fp = np.asarray(fpoint)
    val = 0.5 * (var[fp[:, 0]] + var[fp[:, 1]])
    contrib = val[:, :, None] * np.asarray(fnormal)[:, None, :]
    g = np.zeros((len(pvolume), 7, 3))
    np.add.at(g, fp[:, 0], contrib)
    np.subtract.at(g, fp[:, 1], contrib)
    g /= np.asarray(pvolume)[:, None, None]
    g[nown:] = np.nan
    return g


def np_scale(fpoint, fnormal, pvolume, var):
    """cancellation scale s_p = sum_f |n_f| * 0.5*|var_p0 + var_p1| / V_p  (SURVEY.md section 8c)."""
    fp = np.asarray(fpoint)
    a = 0.5 * np.abs(var[fp[:, 0]] + var[fp[:, 1]])
    contrib = a[:, :, None] * np.abs(np.asarray(fnormal))[:, None, :]
    s = np.zeros((len(pvolume), 7, 3))
    np.add.at(s, fp[:, 0], contrib)
    np.add.at(s, fp[:, 1], contrib)
    return s / np.asarray(pvolume)[:, None, None]


def np_flux(fpoint, fnormal, grad, nown, mode=0):
    """pseudo flux on owned points.  mode 0: consistent; mode 1: the reference's 1-thread
    result (src/flux.c:177-188 with the classes of src/rangelist.c:719-736): the p0 end of a
    face only receives +flux when p1 is a ghost."""
    fp = np.asarray(fpoint)
    n = np.asarray(fnormal)
    d = 0.5 * (grad[fp[:, 0], :3, :] + grad[fp[:, 1], :3, :])  # [F, v, k]
    lam = -2.0 / 3.0
    sxx = lam * (d[:, 1, 1] + d[:, 2, 2] - 2.0 * d[:, 0, 0])
    syy = lam * (d[:, 0, 0] + d[:, 2, 2] - 2.0 * d[:, 1, 1])
    szz = lam * (d[:, 0, 0] + d[:, 1, 1] - 2.0 * d[:, 2, 2])
    sxy = d[:, 0, 1] + d[:, 1, 0]
    sxz = d[:, 0, 2] + d[:, 2, 0]
    syz = d[:, 1, 2] + d[:, 2, 1]
    fl = -np.stack([sxx * n[:, 0] + sxy * n[:, 1] + sxz * n[:, 2],
                    sxy * n[:, 0] + syy * n[:, 1] + syz * n[:, 2],
                    sxz * n[:, 0] + syz * n[:, 1] + szz * n[:, 2]], axis=1)
    out = np.zeros((grad.shape[0], 3))
    own0, own1 = fp[:, 0] < nown, fp[:, 1] < nown
    add0 = own0 & (~own1 if mode == 1 else np.ones_like(own0))
    np.add.at(out, fp[add0, 0], fl[add0])
    np.subtract.at(out, fp[own1, 1], fl[own1])
    out[nown:] = np.nan
    return out
