"""Shared fixtures.  `-m "not gpu"`: oracle vs golden vectors, host logic, ABI symbols (CPU only).
`-m gpu`: the parity tests proper, through the C ABI, on an MI355X."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import build, load_oracle, load_package  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
TOL = 1e-10  # north_star: gradients within 1e-10 rel of the CPU reference


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an MI355X (run with -m gpu on the GPU box)")


def _libs_present():
    lib = os.path.join(ROOT, "cfd-proxy_amd", "lib")
    return all(os.path.exists(os.path.join(lib, n)) for n in ("libcfdproxy_host.so", "libcfdproxy_hip.so")) and \
        os.path.exists(os.path.join(ROOT, "oracle", "libcpu_ref.so"))


@pytest.fixture(scope="session")
def pkg():
    if not _libs_present():
        build()
    return load_package()


@pytest.fixture(scope="session")
def orc():
    if not _libs_present():
        build()
    return load_oracle()


def has_gpu() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu(pkg):
    """GPU tests never fall back: a missing extension is an error, a missing device a skip
    (the CPU container), never a silent pass."""
    pkg.hip_lib()  # raises if the HIP library is not built
    if pkg.hip_lib().cfdp_gpu_device_count() <= 0:
        pytest.skip("no HIP device in this container")
    return pkg


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def golden_domain(pkg, fx, d):
    """rebuild the (solver_data, comm_data) pair of fixture domain d"""
    nd = int(fx["ndomains"])
    kw = {}
    if nd > 1:
        kw = dict(addpoint_owner=fx[f"d{d}_addpoint_owner"], addpoint_idx=fx[f"d{d}_addpoint_idx"],
                  commpartner=fx[f"d{d}_commpartner"], sendcount=fx[f"d{d}_sendcount"],
                  recvcount=fx[f"d{d}_recvcount"])
    return pkg.domain_from_arrays(fx[f"d{d}_fpoint"], fx[f"d{d}_fnormal"], fx[f"d{d}_pvolume"],
                                  int(fx[f"d{d}_nown"]), var=fx[f"d{d}_var"], ndomains=nd, iproc=d, **kw)


def rel_err(orc, got, ref, fpoint, fnormal, pvolume, var, nown):
    """the tolerance of SURVEY.md section 8c: |d| <= tol * max(|g_ref|, s_p) per component, where
    s_p is the cancellation scale (Green-Gauss sums cancel to ~0 on smooth data)"""
    scale = np.maximum(np.abs(ref), orc.np_scale(fpoint, fnormal, pvolume, var))
    scale = np.where(scale > 0, scale, 1.0)
    return float((np.abs(got - ref)[:nown] / scale[:nown]).max())


def whole_mesh_scale(orc, truth, fpoint, fnormal, pvolume, var):
    """per-component scale max(|g_ref|, s_p) of the un-partitioned mesh, to be indexed by global id: the
    criterion of rel_err() for rows of a partition (ghost rows take the scale of the owner point)"""
    scale = np.maximum(np.abs(truth), orc.np_scale(fpoint, fnormal, pvolume, var))
    return np.where(scale > 0, scale, 1.0)


def rel_err_rows(got, truth_rows, scale_rows):
    return float((np.abs(got - truth_rows) / scale_rows).max())
