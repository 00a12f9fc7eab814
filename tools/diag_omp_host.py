"""diagnostics: the C host of tests/test_gpu_parity.py::test_entry_points_called_by_every_thread_of_an_omp_region, run
directly under `timeout` with the call trace on, in several configurations; everything goes to gpurun_out/diag_omp_*"""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
pkg = load_package()
tmp = tempfile.mkdtemp()
gp = pkg.gen_params(14, 12, 10, ndomains=1)
prefix = os.path.join(tmp, "dualgrid")
pkg.write_mesh(gp, prefix, 2)
exe = os.path.join(tmp, "host_omp_driver")
lib = os.path.join(ROOT, "cfd-proxy_amd", "lib")
r = subprocess.run(["gcc", "-std=gnu99", "-O1", "-g", "-fopenmp", os.path.join(ROOT, "tests", "host_omp_driver.c"), "-I" + os.path.join(ROOT, "include"),
                    "-L" + lib, "-lcfdproxy_hip", "-Wl,-rpath," + lib, "-Wl,--allow-shlib-undefined", "-o", exe], capture_output=True, text=True)
print("build", r.returncode, r.stderr[-500:], flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
for label, env in (("t4_auto", dict(OMP_NUM_THREADS="4")), ("t4_auto_notrace", dict(OMP_NUM_THREADS="4", CFDP_CALL_TRACE="0"))):
    for fusion in ("1", "0"):
        out = os.path.join(ROOT, "gpurun_out", f"diag_omp_{label}_f{fusion}")
        with open(out + ".out", "w") as so, open(out + ".err", "w") as se:
            rc = subprocess.run(["timeout", "-k", "5", "40", exe, prefix, "2", "3", os.path.join(tmp, "o")],
                                env=dict(dict(os.environ, CFDP_FUSION=fusion, CFDP_CALL_TRACE="1"), **env), stdout=so, stderr=se).returncode
        print(label, "fusion", fusion, "rc", rc, flush=True)
