import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from __graft_entry__ import load_package
import numpy as np
m = load_package()
import torch
dom = m.gen_domain(m.gen_params(32, ndomains=1), 0); m.fill_var(dom, None, m.VAR_HASH)
free0 = torch.cuda.mem_get_info()[0]
for i in range(150):
    part = m.GpuPartition(dom); part.set_fusion(True)
    part.run_iterations(60, True, 0, use_graph=True)
    part.pull_fields()
    if i == 0: g0 = dom.grad.copy(); f0 = dom.psd_flux.copy()
    assert np.array_equal(dom.grad, g0) and np.array_equal(dom.psd_flux, f0)
    part.close()
    if i == 5: free5 = torch.cuda.mem_get_info()[0]
free1 = torch.cuda.mem_get_info()[0]
print("create/destroy x150: device memory drift after warm-up %.1f MB" % ((free5 - free1) / 1e6), flush=True)
dom.free()
dom = m.gen_domain(m.gen_params(64, ndomains=1), 0); m.fill_var(dom, None, m.VAR_HASH)
part = m.GpuPartition(dom); part.set_fusion(True)
part.run_iterations(1000, True, 0, use_graph=True); part.pull_fields(); g0 = dom.grad.copy(); f0 = dom.psd_flux.copy()
t = time.time(); part.run_iterations(300001, True, 0, use_graph=True); part.sync(); dt = time.time() - t
part.pull_fields()
assert np.array_equal(dom.grad, g0) and np.array_equal(dom.psd_flux, f0)
print("300001 iterations in %.2f s = %.2f us each, results bitwise unchanged" % (dt, dt / 300001 * 1e6), flush=True)
