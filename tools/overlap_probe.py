"""development helper: step time of rank 0 of the 2-rank bench decomposition on ONE GPU, the rank
exchanging with itself through RCCL (real send/recv kernels on the comm stream): what do the
boundary/interior schedules cost?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from __graft_entry__ import load_package
pkg = load_package()
from cfd_proxy_amd import multigpu as mg
world = 2
dims, nd = mg.bench_mesh(world)
gp = pkg.gen_params(*dims, ndomains=nd)
parts = [mg.build_rank_partition(gp, nd, world, r, via_files=False)[0] for r in range(world)]
reqs = [{int(k): (v[0], v[1]) for k, v in pkg.merge_requests(p).items()} for p in parts]
for r, p in enumerate(parts):
    mg.exchange_requests(p, r, world, None, all_requests=reqs)
part = parts[0]
print("send", len(part.sendindex(1)), "recv", len(part.recvindex(1)), flush=True)
g = pkg.GpuPartition(part)
g.set_fusion(True)
lib = mg.RankSolver.torch_rccl_path()
g.rccl_init(pkg.GpuPartition.rccl_unique_id(lib), 1, 0, rank_of_partner=[0], libpath=lib, self_exchange=True)
def cost(tag, **kw):
    g.run_steps_rccl(50, **kw); g.sync(); t = time.perf_counter(); g.run_steps_rccl(500, **kw); t1 = time.perf_counter(); g.sync(); t2 = time.perf_counter()
    print("%-28s host %.1f us/step, total %.1f us/step" % (tag, (t1 - t) / 500 * 1e6, (t2 - t) / 500 * 1e6), flush=True)
for _ in range(2):
    cost("no exchange", with_exchange=False, overlap=False, with_flux=True)
    cost("exchange, bulk", with_exchange=True, overlap=False, with_flux=True)
    cost("exchange, overlapped", with_exchange=True, overlap=True, with_flux=True)
g.pull_fields()
print("ghost == sent:", bool(np.array_equal(part.grad[part.recvindex(1)], part.grad[part.sendindex(1)])), flush=True)
g.close()
