"""The C-ABI library loads (no GPU needed) and exports every symbol include/*.h declares."""
import ctypes
import os
import re
import subprocess

import pytest

from conftest import ROOT


def declared_functions(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    src = re.sub(r"^\s*#.*?$", "", src, flags=re.M)
    src = re.sub(r"\{[^{}]*\}", "{}", src)  # bodies of structs/enums/inline functions
    src = re.sub(r"\{[^{}]*\}", "{}", src)
    names = []
    for stmt in src.split(";"):
        stmt = " ".join(stmt.split())
        if not stmt or stmt.startswith("typedef") or "static inline" in stmt or stmt.startswith("extern \"C\""):
            stmt = stmt.replace('extern "C" {}', "").replace('extern "C" {', "").strip()
            if not stmt or stmt.startswith("typedef") or "static inline" in stmt:
                continue
        m = re.match(r"^(?:[\w]+[\s\*]+)+(\w+)\s*\(", stmt)
        if m:
            names.append(m.group(1))
    return names


@pytest.mark.parametrize("header,lib", [("cfdproxy_hip.h", "libcfdproxy_hip.so"), ("cfdproxy_host.h", "libcfdproxy_hip.so"),
                                        ("cfdproxy_dropin.h", "libcfdproxy_hip.so"), ("cfdproxy_host.h", "libcfdproxy_host.so")])
def test_every_declared_symbol_is_exported(pkg, header, lib):
    names = declared_functions(header)
    assert len(names) >= 10, names
    handle = ctypes.CDLL(os.path.join(ROOT, "cfd-proxy_amd", "lib", lib))
    missing = [n for n in names if not hasattr(handle, n)]
    assert not missing, f"{lib} does not export {missing}"


def test_expected_entry_points_are_declared():
    hip = declared_functions("cfdproxy_hip.h")
    for n in ("cfdp_gpu_create", "cfdp_gpu_upload_plan", "cfdp_gpu_gradients", "cfdp_gpu_flux", "cfdp_gpu_pack",
              "cfdp_gpu_unpack", "cfdp_gpu_rank_gradients", "cfdp_gpu_rank_flux", "cfdp_gpu_time_kernels",
              "cfdp_gpu_step_pre", "cfdp_gpu_step_post", "cfdp_gpu_set_fusion", "cfdp_gpu_bind_grad_alt",
              "cfdp_gpu_time_fused", "cfdp_gpu_vcycle"):
        assert n in hip
    drop = declared_functions("cfdproxy_dropin.h")
    for n in ("init_communication", "read_communication_data", "compute_communication_tables",
              "free_communication_ressources", "read_solver_data", "init_solver_data", "init_threads", "test_solver",
              "compute_psd_flux", "get_nc_int", "get_nc_double", "get_nc_val", "cfdp_test_vcycle") + tuple(
            "compute_gradients_gg_" + v for v in ("comm_free", "mpi_bulk_sync", "mpi_early_recv", "mpi_async",
                                                  "gaspi_bulk_sync", "gaspi_async", "mpifence_bulk_sync",
                                                  "mpifence_async", "mpipscw_bulk_sync", "mpipscw_async")):
        assert n in drop, n


def test_product_never_touches_the_oracle():
    """the product path must not import, link or execute anything under oracle/"""
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "cfd-proxy_amd")):
        if "/build" in base or base.endswith("/lib") or base.endswith("/bin") or "__pycache__" in base:
            continue
        for f in files:
            if f.endswith((".c", ".h", ".hip", ".py", ".cpp")) or f == "Makefile":
                txt = open(os.path.join(base, f), errors="ignore").read()
                if re.search(r"cpu_ref|libcpu_ref|load_oracle|oracle/", txt) and "never" not in txt.split("oracle/")[0][-200:]:
                    for line in txt.splitlines():
                        if re.search(r"cpu_ref|libcpu_ref|load_oracle", line) or (
                                "oracle/" in line and not line.lstrip().startswith(("*", "//", "#", '"""'))):
                            bad.append((f, line.strip()))
    assert not bad, bad


def test_gpu_entry_points_fail_loudly_without_a_device(pkg):
    """no CPU fallback: without a GPU the ABI reports an error instead of computing"""
    hip = pkg.hip_lib()
    if hip.cfdp_gpu_device_count() > 0:
        pytest.skip("a GPU is present")
    h = ctypes.c_void_p()
    assert hip.cfdp_gpu_create(0, ctypes.byref(h)) != 0
    assert b"HIP" in hip.cfdp_gpu_last_error() or b"hip" in hip.cfdp_gpu_last_error()
    gp = pkg.gen_params(4, 4, 4)
    dom = pkg.gen_domain(gp, 0)
    with pytest.raises(pkg.GpuError):
        pkg.GpuPartition(dom)
    dom.free()


def test_integration_md_host_program_compiles_and_links(tmp_path):
    """the reference's main() as INTEGRATION.md section 2 shows it against the drop-in header: compiled
    and linked with gcc against libcfdproxy_hip.so (no GPU call is made here)"""
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    block = re.search(r"```c\n(#include \"cfdproxy_dropin.h\".*?)```", text, re.S).group(1)
    src = tmp_path / "main.c"
    first, rest = block.split("\n", 1)  # (NTHREADS is a variable of the reference's main(): src/hybrid.f6.c:29-52)
    src.write_text("#include <stdio.h>\n#include <stdlib.h>\n" + first + "\nstatic const int NTHREADS = 1;\n" + rest)
    exe = tmp_path / "host"
    r = subprocess.run(["gcc", "-std=gnu99", "-Wall", "-Werror", str(src), "-I" + os.path.join(root, "include"),
                        "-L" + os.path.join(root, "cfd-proxy_amd", "lib"), "-lcfdproxy_hip", "-fopenmp",
                        "-Wl,-rpath," + os.path.join(root, "cfd-proxy_amd", "lib"), "-Wl,--allow-shlib-undefined",
                        "-o", str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_omp_test_host_compiles_and_links(tmp_path):
    """tests/host_omp_driver.c (every OpenMP thread calls the entry points, src/solver.c:45-55) builds against
    the drop-in header and library with -Wall -Werror; it runs in the GPU suite"""
    import subprocess
    lib = os.path.join(ROOT, "cfd-proxy_amd", "lib")
    r = subprocess.run(["gcc", "-std=gnu99", "-Wall", "-Werror", "-fopenmp", os.path.join(ROOT, "tests", "host_omp_driver.c"),
                        "-I" + os.path.join(ROOT, "include"), "-L" + lib, "-lcfdproxy_hip", "-Wl,-rpath," + lib,
                        "-Wl,--allow-shlib-undefined", "-o", str(tmp_path / "host")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


@pytest.mark.skipif(not os.path.exists("/root/reference/src/hybrid.f6.c"), reason="needs the reference checkout")
def test_reference_main_links_unchanged_against_the_dropin(tmp_path):
    """src/hybrid.f6.c:54-91 -- nc_open(fname, NC_NOWRITE, &ncid), ERR() -> nc_strerror, f_exist, nc_close -- is
    compiled where it lies (stdin, so its #include "..." lines find include/compat/) and linked against the product
    library; without a GPU it gets as far as init_threads and stops with the library's message"""
    import subprocess
    exe = os.path.join(ROOT, "oracle", "_ref", "hybrid.f6.dropin")
    assert os.path.exists(exe), "make -C oracle builds it"
    nm = subprocess.run(["nm", "-u", exe], capture_output=True, text=True).stdout
    for sym in ("nc_open", "nc_close", "nc_strerror", "f_exist", "read_solver_data", "init_threads", "test_solver"):
        assert sym in nm, sym
    from __graft_entry__ import load_package
    pkg = load_package()
    if pkg.hip_lib().cfdp_gpu_device_count() > 0:
        pytest.skip("a GPU is present: the GPU suite runs the binary to the end")
    gp = pkg.gen_params(6, 6, 6, ndomains=1)
    pkg.write_mesh(gp, str(tmp_path / "dualgrid"), 2)
    # (relative prefix: the reference's main() builds the file name in a char[80], src/hybrid.f6.c:57-62)
    r = subprocess.run([exe, "-lvl", "2", "dualgrid"], env=dict(os.environ, OMP_NUM_THREADS="2"), cwd=str(tmp_path),
                       capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "no HIP device" in r.stderr, r.stdout + r.stderr


def _build_call_election_host(tmp_path, lib_name="cfdproxy_host"):
    exe = str(tmp_path / "host_call_election")
    lib = os.path.join(ROOT, "cfd-proxy_amd", "lib")
    r = subprocess.run(["gcc", "-std=gnu99", "-O1", "-Wall", "-Werror", "-fopenmp", os.path.join(ROOT, "tests", "host_call_election.c"),
                        "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "cfd-proxy_amd", "host"),
                        "-L" + lib, "-l" + lib_name, "-Wl,-rpath," + lib, "-lpthread", "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


@pytest.mark.parametrize("lib_name", ["cfdproxy_host", "cfdproxy_hip"])
@pytest.mark.parametrize("scenario", ["team", "serial_threads", "mixed", "master", "pthread_team", "team_then_master"])
def test_entry_point_calls_are_performed_exactly_once(pkg, tmp_path, scenario, lib_name):
    """host/call_election.c: which caller of compute_gradients_gg_* / compute_psd_flux enqueues the work.  The
    reference's harness (every thread of a team of 4 makes every call, team mates running ahead, src/solver.c:45-55),
    serial callers on ever new threads, serial calls mixed with teams of 4 and 2, one thread of a team making all
    calls, a team of pthreads, and every-thread regions followed by master sections (the master pattern is judged
    per call, not once per solver): every call is performed exactly once, in the order
    it was issued.  Against
    libcfdproxy_hip.so the process holds TWO OpenMP runtimes (the gcc host's libgomp and the libomp hipcc links): the
    team is the one the HOST's runtime knows, and the other runtime is never called (never initialised)"""
    exe = _build_call_election_host(tmp_path, lib_name)
    env = {k: v for k, v in os.environ.items() if k != "CFDP_CALL_MODE"}
    r = subprocess.run([exe, scenario], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 0 and "order broken" not in r.stdout, r.stdout + r.stderr


def test_calls_from_omp_single_sections_are_never_lost_silently(pkg, tmp_path):
    """a team that does not follow the reference's convention (calls from omp single sections): either nothing is
    lost, or the run stops with the message that names the fix -- and with that fix every call is performed"""
    exe = _build_call_election_host(tmp_path)
    env = {k: v for k, v in os.environ.items() if k != "CFDP_CALL_MODE"}
    r = subprocess.run([exe, "single"], capture_output=True, text=True, timeout=120, env=env)
    assert (r.returncode == 0 and "performed 80 of 80" in r.stdout) or \
        (r.returncode == 1 and "not by every thread of the team" in r.stderr and "CFDP_CALL_MODE=every" in r.stderr), r.stdout + r.stderr
    r = subprocess.run([exe, "single"], capture_output=True, text=True, timeout=120, env=dict(env, CFDP_CALL_MODE="every"))
    assert r.returncode == 0 and "performed 80 of 80" in r.stdout, r.stdout + r.stderr


def test_product_library_holds_no_diagnostic_kernel(pkg):
    """the diagnostic instantiations of the fused pass (DIAG = 1 phase stamps, 2 data movement only, 3 the skip-pre timing
    experiment: csrc/gg_diag.hip) live in lib/libcfdproxy_diag.so, which the product library loads only when a diagnostic is
    asked for: libcfdproxy_hip.so holds the kernels a timed run can execute and nothing else"""
    import re
    libdir = os.path.join(ROOT, "cfd-proxy_amd", "lib")

    def fused_forms(lib):
        out = subprocess.run(["nm", "-C", "--defined-only", os.path.join(libdir, lib)], capture_output=True, text=True, check=True).stdout
        return set(re.findall(r"gg_fused_split_kernel<([^>]*)>", out))

    product, diag = fused_forms("libcfdproxy_hip.so"), fused_forms("libcfdproxy_diag.so")
    assert product and all(f.split(", ")[6] == "0" for f in product), sorted(product)
    assert diag and {f.split(", ")[6] for f in diag} == {"1", "2", "3"}, sorted(diag)
    # both capacities of the real pass, with and without the exchange riding in it, with and without row lists
    assert {tuple(f.split(", ")[2:6]) for f in product} == {("5", "3", "3", "3"), ("6", "4", "3", "4")}
    d = ctypes.CDLL(os.path.join(libdir, "libcfdproxy_diag.so"))
    assert d.gg_diag_launch_fused and d.gg_diag_set_stamp_buffer
