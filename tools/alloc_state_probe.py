"""round 5: the 64^3 fused pass shows two states on one box (35.2-36 us / 37.9-38.5 us).  Is it where the arrays sit?
Several partitions of the same mesh in one process, kept alive or closed in between; the device addresses of grad and var
(mod 2 MiB) beside the time of each.   python tools/alloc_state_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
m = load_package()
import ctypes as C
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dom = m.gen_domain(m.gen_params(n, ndomains=1), 0); m.fill_var(dom, None, m.VAR_HASH)
lib = m.hip_lib()
lib.cfdp_gpu_grad_ptr.restype = C.c_void_p; lib.cfdp_gpu_grad_ptr.argtypes = [C.c_void_p]
lib.cfdp_gpu_var_ptr.restype = C.c_void_p; lib.cfdp_gpu_var_ptr.argtypes = [C.c_void_p]
keep = []
for mode in ("closed in between", "kept alive"):
    for rep in range(4):
        part = m.GpuPartition(dom); part.set_fusion(True)
        part.time_fused(800)
        ts = sorted(part.time_fused(200) for _ in range(7))
        g, v = lib.cfdp_gpu_grad_ptr(part.h), lib.cfdp_gpu_var_ptr(part.h)
        print(f"{mode:18s} #{rep}: fused pass {ts[0]*1e3:.2f} / {ts[3]*1e3:.2f} / {ts[6]*1e3:.2f} us   grad {g:#x} (mod 2 MiB {g % (2<<20):#x})  var {v:#x} (mod 2 MiB {v % (2<<20):#x})", flush=True)
        if mode == "kept alive": keep.append(part)
        else: part.close()
for p in keep: p.close()
