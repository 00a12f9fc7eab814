"""host <-> device cost of the drop-in boundary's field transfers on the 64^3 mesh (file numbering on the
host side, so each includes the renumbering pass): not part of any benchmark figure."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
m = load_package()
dom = m.gen_domain(m.gen_params(64, ndomains=1), 0); m.fill_var(dom, None, m.VAR_HASH)
part = m.GpuPartition(dom)
part.run_iterations(2, True, 0, use_graph=False); part.sync()
def t(f, n=10):
    f(); part.sync(); t0 = time.perf_counter()
    for _ in range(n): f()
    part.sync(); return (time.perf_counter() - t0) / n * 1e3
print("set_var  (64 B/pt up)   %.3f ms" % t(lambda: part._ck(part.lib.cfdp_gpu_set_var(part.h, dom.sd.var))))
print("get_grad (168 B/pt down) %.3f ms" % t(lambda: part._ck(part.lib.cfdp_gpu_get_grad(part.h, dom.sd.grad))))
print("get_flux (24 B/pt down)  %.3f ms" % t(lambda: part._ck(part.lib.cfdp_gpu_get_flux(part.h, dom.sd.psd_flux))))
part.close()
