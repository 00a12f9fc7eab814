"""development helper: the self-sendrecv plumbing test of the C-side RCCL exchange, with progress
lines (a hang shows where) -- run under `timeout`"""
import faulthandler, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
faulthandler.dump_traceback_later(60, exit=True)
import numpy as np
def log(*a):
    print(*a, flush=True)
import torch
from __graft_entry__ import load_package
pkg = load_package()
from cfd_proxy_amd import multigpu as mg
gp = pkg.gen_params(12, 10, 8, ndomains=2)
parts = [mg.build_rank_partition(gp, 2, 2, r, via_files=False)[0] for r in range(2)]
reqs = [{int(k): (v[0], v[1]) for k, v in pkg.merge_requests(p).items()} for p in parts]
for r, p in enumerate(parts):
    mg.exchange_requests(p, r, 2, None, all_requests=reqs)
part = parts[0]
g = pkg.GpuPartition(part, tile_points=int(os.environ.get("TP", "64")))
g.set_fusion(os.environ.get("FUSION", "0") == "1")
lib = mg.RankSolver.torch_rccl_path()
log("lib", lib)
uid = pkg.GpuPartition.rccl_unique_id(lib)
log("unique id ok")
g.rccl_init(uid, 1, 0, rank_of_partner=[0], libpath=lib, self_exchange=True)
log("comm init ok")
def check(tag):
    g.pull_fields()
    sent = part.grad[part.sendindex(1)]; got = part.grad[part.recvindex(1)]
    log(tag, "equal", bool(np.array_equal(got, sent)), "max", float(np.abs(sent).max()))
part.grad[:] = -1.0; g.push_fields()
g.step_rccl(True, True, True); log("step enqueued")
g.sync(); log("step synced")
check("1 step")
for _ in range(3): g.step_rccl(True, False, True)
check("3 bulk steps")
g.run_steps_rccl(8, True, True, True); log("eager run")
check("eager 8")
import time
def host_cost(tag, **kw):
    g.sync(); t = time.perf_counter(); g.run_steps_rccl(500, **kw); t1 = time.perf_counter(); g.sync(); t2 = time.perf_counter()
    log("%-34s host enqueue %.1f us/step, total %.1f us/step" % (tag, (t1 - t) / 500 * 1e6, (t2 - t) / 500 * 1e6))
for _ in range(2):
    host_cost("no exchange", with_exchange=False, overlap=False, with_flux=True)
    host_cost("exchange, bulk", with_exchange=True, overlap=False, with_flux=True)
    host_cost("exchange, overlapped", with_exchange=True, overlap=True, with_flux=True)
g.close(); log("done")
