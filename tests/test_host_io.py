"""NetCDF-classic reader/writer and the dualgrid loader (host C side, CPU only)."""
import os
import subprocess
import sys

import numpy as np
import pytest
from scipy.io import netcdf_file

from conftest import ROOT


def test_writer_is_read_by_scipy_and_by_our_loader(pkg, tmp_path):
    gp = pkg.gen_params(9, 8, 7, ndomains=3)
    prefix = str(tmp_path / "dualgrid")
    pkg.write_mesh(gp, prefix, 2)
    for d in range(3):
        mem = pkg.gen_domain(gp, d)
        path = f"{prefix}_domain_{d}_lvl_2"
        f = netcdf_file(path, "r", mmap=False)
        assert f.version_byte == 1
        assert f.dimensions["nfaces"] == mem.nfaces and f.dimensions["ndomains"] == 3
        assert np.array_equal(f.variables["fpoint"][:], mem.fpoint)
        assert np.array_equal(f.variables["fnormal"][:], mem.fnormal)
        assert np.array_equal(f.variables["pvolume"][:], mem.pvolume)
        assert np.array_equal(f.variables["addpoint_owner"][:], mem.addpoint_owner())
        assert np.array_equal(f.variables["addpoint_idx"][:], mem.addpoint_id())
        f.close()
        dom = pkg.load_domain(prefix, d, 2)
        assert (dom.nown, dom.nall, dom.nfaces) == (mem.nown, mem.nall, mem.nfaces)
        assert np.array_equal(dom.fpoint, mem.fpoint) and np.array_equal(dom.fnormal, mem.fnormal)
        assert np.array_equal(dom.pvolume, mem.pvolume)
        assert np.all(dom.var == 1.0) and np.all(dom.grad == 1.0) and np.all(dom.psd_flux == 1.0)  # solver_data.c:26-63
        assert dom.partners == mem.partners
        for k in range(3):
            assert dom.cd.sendcount[k] == mem.cd.sendcount[k] and dom.cd.recvcount[k] == mem.cd.recvcount[k]
        dom.free()
        mem.free()


@pytest.mark.parametrize("version", [1, 2])
def test_reader_handles_scipy_files_with_attributes_and_other_types(pkg, tmp_path, version):
    """files not written by us: attributes, int16/float32 variables, CDF-2 offsets"""
    import ctypes as C
    path = str(tmp_path / "x.nc")
    f = netcdf_file(path, "w", version=version)
    f.history = "written by scipy"
    f.createDimension("n", 5)
    f.createDimension("m", 3)
    v = f.createVariable("a", "i", ("n",))
    v[:] = np.arange(5) - 2
    v.units = "none"
    w = f.createVariable("b", "d", ("n", "m"))
    w[:] = np.arange(15).reshape(5, 3) * 0.25
    s = f.createVariable("s", "h", ("m",))
    s[:] = np.array([-3, 0, 7], np.int16)
    g = f.createVariable("g", "f", ("m",))
    g[:] = np.array([1.5, -2.25, 1e6], np.float32)
    f.close()
    lib = pkg.host_lib()
    ncid = lib.cfdp_nc_open(path.encode())
    assert lib.get_nc_val(ncid, b"n") == 5 and lib.get_nc_val(ncid, b"m") == 3
    a = np.zeros(5, np.int32)
    lib.get_nc_int(ncid, b"a", a.ctypes.data_as(C.POINTER(C.c_int)))
    assert np.array_equal(a, np.arange(5) - 2)
    b = np.zeros(15)
    lib.get_nc_double(ncid, b"b", b.ctypes.data_as(C.POINTER(C.c_double)))
    assert np.array_equal(b, np.arange(15) * 0.25)
    s2 = np.zeros(3, np.int32)
    lib.get_nc_int(ncid, b"s", s2.ctypes.data_as(C.POINTER(C.c_int)))
    assert list(s2) == [-3, 0, 7]
    g2 = np.zeros(3)
    lib.get_nc_double(ncid, b"g", g2.ctypes.data_as(C.POINTER(C.c_double)))
    assert list(g2) == [1.5, -2.25, 1e6]
    lib.cfdp_nc_close(ncid)


def test_record_variables(pkg, tmp_path):
    import ctypes as C
    path = str(tmp_path / "rec.nc")
    f = netcdf_file(path, "w")
    f.createDimension("t", None)
    f.createDimension("m", 2)
    v = f.createVariable("r", "d", ("t", "m"))
    for i in range(4):
        v[i] = [i, 10 + i]
    q = f.createVariable("q", "i", ("t",))
    for i in range(4):
        q[i] = -i
    f.close()
    lib = pkg.host_lib()
    ncid = lib.cfdp_nc_open(path.encode())
    assert lib.get_nc_val(ncid, b"t") == 4
    r = np.zeros(8)
    lib.get_nc_double(ncid, b"r", r.ctypes.data_as(C.POINTER(C.c_double)))
    assert np.array_equal(r.reshape(4, 2), np.array([[i, 10 + i] for i in range(4)], float))
    q2 = np.zeros(4, np.int32)
    lib.get_nc_int(ncid, b"q", q2.ctypes.data_as(C.POINTER(C.c_int)))
    assert list(q2) == [0, -1, -2, -3]
    lib.cfdp_nc_close(ncid)


def _run(code):
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT, timeout=120)


def test_loader_errors_exit_with_code_2_like_the_reference(tmp_path):
    """reference ERR(): message + exit(2) (src/error_handling.h:4-10)"""
    pre = "import sys; sys.path.insert(0, '.'); from __graft_entry__ import load_package; p = load_package(); l = p.host_lib();"
    r = _run(pre + "l.cfdp_nc_open(b'/nonexistent/file')")
    assert r.returncode == 2 and "Error" in r.stderr
    bad = tmp_path / "bad.nc"
    bad.write_bytes(b"HDF5 not netcdf classic")
    r = _run(pre + f"l.cfdp_nc_open(b'{bad}')")
    assert r.returncode == 2 and "not a NetCDF classic" in r.stderr
    # the containers newer than what the reference's pinned libnetcdf 3.6.3 reads are named, with the fix
    for magic, what in ((b"CDF\x05" + b"\0" * 60, "CDF-5"), (b"\x89HDF\r\n\x1a\n" + b"\0" * 60, "NetCDF-4/HDF5")):
        bad.write_bytes(magic)
        r = _run(pre + f"l.cfdp_nc_open(b'{bad}')")
        assert r.returncode == 2 and what in r.stderr and "nccopy -k classic" in r.stderr, r.stderr
    # libnetcdf's own calling convention (what the reference's main() uses): a code, explained by nc_strerror
    r = _run(pre + "import ctypes as C; i = C.c_int(); l.nc_strerror.restype = C.c_char_p; "
                   "rc = l.nc_open(b'/nonexistent/file', 0, C.byref(i)); print(rc, l.nc_strerror(rc).decode()); "
                   "print(l.nc_close(12345), l.nc_strerror(l.nc_close(12345)).decode())")
    assert r.returncode == 0 and "I/O error" in r.stdout and "not a valid ncid" in r.stdout, r.stdout + r.stderr
    gp_code = pre + f"gp = p.gen_params(4,4,4); p.write_mesh(gp, '{tmp_path}/m', 1); n = l.cfdp_nc_open(b'{tmp_path}/m_domain_0_lvl_1'); l.get_nc_val(n, b'no_such_dim')"
    r = _run(gp_code)
    assert r.returncode == 2 and "no_such_dim" in r.stderr


def test_single_domain_file_has_no_comm_tables(pkg, tmp_path):
    gp = pkg.gen_params(5, 5, 5, ndomains=1, cdf_version=2)
    prefix = str(tmp_path / "m")
    pkg.write_mesh(gp, prefix, 3)
    f = netcdf_file(prefix + "_domain_0_lvl_3", "r", mmap=False)
    assert f.version_byte == 2 and "naddpoints" not in f.dimensions and f.dimensions["ndomains"] == 1
    f.close()
    dom = pkg.load_domain(prefix, 0, 3)
    assert dom.cd.ndomains == 1 and dom.nown == dom.nall == 125 and dom.partners == []
    dom.free()


def test_loader_keeps_its_own_reader_beside_foreign_netcdf_symbols(pkg, tmp_path):
    """an application that defines functions with libnetcdf's names (or links a real libnetcdf) and its own now():
    the library's loader calls bind to the library's own reader (-Bsymbolic-functions), not to the application's"""
    import subprocess
    gp = pkg.gen_params(6, 5, 4, ndomains=1)
    prefix = str(tmp_path / "m")
    pkg.write_mesh(gp, prefix, 2)
    exe = str(tmp_path / "host_foreign_symbols")
    lib = os.path.join(ROOT, "cfd-proxy_amd", "lib")
    r = subprocess.run(["gcc", "-std=gnu99", "-O1", "-Wall", "-Werror", os.path.join(ROOT, "tests", "host_foreign_symbols.c"),
                        "-I" + os.path.join(ROOT, "include"), "-L" + lib, "-lcfdproxy_host", "-Wl,-rpath," + lib, "-o", exe],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe, prefix + "_domain_0_lvl_2"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and r.stdout.split()[1] == "120" and r.stdout.split()[2] == "120", r.stdout + r.stderr
